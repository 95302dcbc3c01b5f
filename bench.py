#!/usr/bin/env python3
"""bench.py -- spectrogram frames/s (fwd + invert), n_fft=1024 hop=256 (BASELINE.json metric).

One "step" = one pass of the hot path over one batch of synthetic audio that is
already resident in HBM:
    X, feat = STFT.forward(x) + Magnitude(mel, n_mels=128).forward(X)   one fused kernel:
              X (B, 690, 513) complex64 and feat (B, 690, 128) float32 (log1p + unipolar normalise)
    y       = STFT.invert(X)                     (B, 176384)   float32
(--unfused runs the two forward stages as separate kernels: STFT, then the MFMA projection)
on BASELINE config[1]: batch = 1024 clips x 4 s @ 44.1 kHz mono per GPU, fp32.
N > 1: one process per GPU (launched by torch.distributed.run), clips sharded,
weak scaling, no data-path collective; `value` is the whole-job frames/s.
Extra figures (all-gather of the features over RCCL, PGHI round trip, per-kernel
roofline, CPU baseline) ride along in the same JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

N_FFT, HOP, SR = 1024, 256, 44100
CLIP_LEN = 4 * SR                     # 176400 samples
T_FRAMES = 1 + CLIP_LEN // HOP        # 690
F_BINS = N_FFT // 2 + 1               # 513
N_MELS = 128
HBM_PEAK_GBS = 8000.0                 # MI355X HBM3E spec (MI355X_MICROARCH.md: 8 TB/s, ~6.3 achievable)
MFMA_F32_PEAK_TFLOPS = 157.3          # fp32-input MFMA spec
# algorithmic bytes per frame
BYTES_STFT_FWD = HOP * 4 + F_BINS * 8            # 5128
BYTES_ISTFT = F_BINS * 8 + HOP * 4               # 5128
BYTES_MEL = F_BINS * 8 + N_MELS * 4              # 4616 (unfused: reads the complex spectrum)
FLOPS_MEL = 2 * F_BINS * N_MELS                  # 131328 dense


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=1024, help="clips per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--unfused", action="store_true", help="STFT and Magnitude as two kernels (MFMA projection)")
    ap.add_argument("--no-extras", action="store_true", help="skip the all-gather / PGHI side measurements")
    ap.add_argument("--pghi-clips", type=int, default=1024, help="clips for the DGT+PGHI round-trip side measurement")
    ap.add_argument("--streams", type=int, default=256, help="concurrent streams for the RealtimeDGT side measurement")
    return ap.parse_args()


def cpu_baseline(sample_clips=96, reps=3):
    """The oracle (CPU restatement, torch CPU ops) timed on this box's host cores on a bounded
    sample of the same workload: STFT fwd + Magnitude(mel128) + ISTFT."""
    from oracle import oracle as O
    threads = torch.get_num_threads()
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(sample_clips, CLIP_LEN, generator=g) * 0.1
    w = O.hann_window(N_FFT)
    fwd, _ = O.magnitude_banks(O.melscale_fbanks(F_BINS, 0.0, SR / 2, N_MELS, SR))
    X = O.stft_forward(x[:4], w, N_FFT, HOP)
    off, sc = O.magnitude_scale_stats(X, "log1p", "unipolar")

    def step():
        X = O.stft_forward(x, w, N_FFT, HOP)
        O.magnitude_forward(X, fwd, "log1p", off, sc)
        O.istft(X, w, N_FFT, HOP)

    step()
    t0 = time.perf_counter()
    n = 0
    while n < reps or (time.perf_counter() - t0 < 10.0 and n < 50):
        step()
        n += 1
    dt = (time.perf_counter() - t0) / n
    return {"value": sample_clips * T_FRAMES / dt, "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": "%d clips x 4 s (%d frames) per pass, %d passes, oracle = torch CPU stft+matmul+istft, %d threads"
                      % (sample_clips, sample_clips * T_FRAMES, n, threads)}


def cpu_baseline_pghi(clips=2, frames=173):
    """oracle/pghi_ref.c (exact-order C PGHI) on one host core, dense-noise magnitudes."""
    from oracle import oracle as O
    import numpy as np
    rng = np.random.RandomState(7)
    mag = np.abs(rng.randn(clips, frames, F_BINS) + 1j * rng.randn(clips, frames, F_BINS)).astype(np.float32)
    t0 = time.perf_counter()
    O.pghi_offline_batch(mag, N_FFT, HOP)
    dt = time.perf_counter() - t0
    return {"value": clips * frames / dt, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": "%d clips x %d frames dense noise, exact-heap C PGHI, 1 thread" % (clips, frames)}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        print(json.dumps({"error": "no ROCm device: bench.py measures the HIP path only"}))
        sys.exit(2)
    rehearsal = os.environ.get("ACIDS_BENCH_REHEARSAL") == "1"   # dev only: N ranks share cuda:0 over gloo
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    red_dev = torch.device("cpu") if rehearsal else dev
    import acids_transforms_amd as A

    B = args.batch
    frames_per_step = B * T_FRAMES
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    x = torch.randn(B, CLIP_LEN, device=dev, generator=gen) * 0.1      # synthetic audio, resident in HBM
    stft = A.STFT(sr=SR, n_fft=N_FFT, hop_length=HOP).to(dev)
    mag = A.Magnitude(sr=SR, n_fft=N_FFT, n_mels=N_MELS, mode="unipolar", contrast="log1p").to(dev)
    X = stft(x[:8])
    mag.scale_data(X)                                                  # one-off calibration, outside the timed region
    del X

    ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731
    fused = (not args.unfused) and mag.can_fuse_with(stft, x)
    ktimes = {"stft_fwd": [], "mel": [], "istft": []}

    def step(record=False):
        if record:
            e = [ev() for _ in range(4)]
            e[0].record()
        if fused:
            X, feat = mag.forward_fused(stft, x, return_spectrum=True)
            if record:
                e[1].record()
                e[2].record()
        else:
            X = stft(x)
            if record:
                e[1].record()
            feat = mag(X)
            if record:
                e[2].record()
        y = stft.invert(X)
        if record:
            e[3].record()
            return X, feat, y, e
        return X, feat, y, None

    def barrier():
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    evs = []
    for _ in range(args.steps):
        out = step(record=True)
        evs.append(out[3])
    del out
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    for e in evs:
        ktimes["stft_fwd"].append(e[0].elapsed_time(e[1]))
        ktimes["mel"].append(e[1].elapsed_time(e[2]))
        ktimes["istft"].append(e[2].elapsed_time(e[3]))
    if world > 1:
        tt = torch.tensor([elapsed], device=red_dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    ms_per_step = elapsed / args.steps * 1e3
    value = world * frames_per_step * args.steps / elapsed

    avg = {k: sum(v) / len(v) for k, v in ktimes.items()}

    def hbm_entry(name, bytes_per_frame):
        a = frames_per_step * bytes_per_frame / (avg[name] * 1e-3) / 1e9
        return {"kernel": name, "bound": "hbm", "achieved": round(a, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(a / HBM_PEAK_GBS, 4), "ms": round(avg[name], 4),
                "algorithmic_bytes_per_frame": bytes_per_frame}

    if fused:
        # one kernel reads the audio and writes spectrum + features: 1024 + 4104 + 512 bytes per frame
        kernels = [hbm_entry("stft_fwd", BYTES_STFT_FWD + 4 * N_MELS), hbm_entry("istft", BYTES_ISTFT)]
        kernels[0]["kernel"] = "stft_fwd+mel (fused)"
        # the stand-alone projection (what Magnitude.forward runs on its own), timed outside the step
        Xs = stft(x)
        for _ in range(2):
            mag(Xs)
        e0, e1 = ev(), ev()
        e0.record()
        for _ in range(5):
            mag(Xs)
        e1.record()
        torch.cuda.synchronize()
        avg["mel"] = e0.elapsed_time(e1) / 5
        # the framing pass on its own (STFT.forward without the fused epilogue), also outside the step
        e0, e1 = ev(), ev()
        e0.record()
        for _ in range(5):
            Xs = stft(x)
        e1.record()
        torch.cuda.synchronize()
        avg["stft_fwd_plain"] = e0.elapsed_time(e1) / 5
        del Xs
        kernels.append(hbm_entry("stft_fwd_plain", BYTES_STFT_FWD))
        kernels[-1]["kernel"] = "stft_fwd (framing + FFT only, outside the step)"
    else:
        kernels = [hbm_entry("stft_fwd", BYTES_STFT_FWD), hbm_entry("istft", BYTES_ISTFT)]
    kernels.append(hbm_entry("mel", BYTES_MEL))
    kernels[-1]["kernel"] = "mel (stand-alone banded projection%s)" % (", outside the step" if fused else "")
    kernels[-1]["note"] = ("HBM-bound banded walk (mel_banded.hip); the dense exact-fp32 MFMA contraction of mel.hip "
                           "remains for banks that are not banded")
    dominant = max(("stft_fwd", "istft"), key=lambda k: avg[k])
    roof = dict(kernels[0] if dominant == "stft_fwd" else kernels[1])
    roof.pop("algorithmic_bytes_per_frame")
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    if os.path.exists(pmc):
        try:
            key = dominant if (fused or dominant != "stft_fwd") else "stft_fwd_unfused"
            traffic = json.load(open(pmc)).get(key)
        except Exception:
            traffic = None
    roof["traffic"] = traffic
    roof["kernel"] = {"stft_fwd": "stft1024_h256_fwd_kernel<false,%d>" % (1 if fused else 0),
                      "istft": "istft1024_ola_kernel<0>"}[dominant]

    extras = {}

    def guarded(name, fn):
        """Side measurements must never cost the headline line."""
        try:
            out = fn()
            if out is not None:
                extras[name] = out
        except Exception as exc:
            extras[name + "_error"] = repr(exc)[:300]

    def extra_allgather(wire_dtype=None):
        # features reassembled on every rank with one RCCL all-gather on a side stream, overlapped with the next step
        # (wire_dtype=torch.bfloat16: the shard is cast on the side stream first -- half the bytes over xGMI; the
        # features computed and kept on the owning rank stay fp32)
        from acids_transforms_amd.dist import all_gather_features
        comm = torch.cuda.Stream(device=dev)
        torch.cuda.synchronize()
        barrier()
        t1 = time.perf_counter()
        pending = None
        for _ in range(args.steps):
            _, feat, _, _ = step()
            done = torch.cuda.Event()
            done.record()
            if pending is not None:
                pending.wait()
            with torch.cuda.stream(comm):
                comm.wait_event(done)
                feat.record_stream(comm)
                wire = feat if wire_dtype is None else feat.to(wire_dtype)
                _, pending = all_gather_features(wire, world * B, async_op=True)
        if pending is not None:
            pending.wait()
        torch.cuda.synchronize()
        barrier()
        tt = torch.tensor([time.perf_counter() - t1], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return world * frames_per_step * args.steps / float(tt.item())

    def extra_pghi():
        # BASELINE config 3: DGT + PGHI invert round trip, dense noise (worst case: every bin above tolerance).
        # One wave per clip, latency-bound: throughput grows with the number of clips in flight.
        dgt = A.DGT(sr=SR, n_fft=N_FFT, hop_length=HOP).to(dev)
        pg = {}
        for nb in sorted({min(args.pghi_clips, B), min(4 * args.pghi_clips, 4096)}):
            xs = x if nb <= B else torch.randn(nb, CLIP_LEN, device=dev, generator=gen) * 0.1
            m = dgt(xs[:nb]).abs()
            yp = dgt.invert(m, inversion_mode="pghi")      # warm-up: first-touch of the (7 MB/clip) workspace
            del yp
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            yp = dgt.invert(m, inversion_mode="pghi")
            torch.cuda.synchronize()
            dt = time.perf_counter() - t1
            pg["clips_%d" % nb] = {"frames_per_s": nb * T_FRAMES / dt, "seconds": dt,
                                   "heap_pops_per_s": float(nb) * T_FRAMES * F_BINS / dt}
            del m, yp, xs
        pg["input"] = "|DGT(randn*0.1)|: ~100% of bins above tolerance; PGHI + polar ISTFT"
        return pg

    def extra_stream():
        # BASELINE config 5 (streaming): chunk -> OverlapAdd frames -> RealtimeDGT -> |X| -> RTPGHI -> irfft ->
        # overlap-add, 1024-sample chunks (4 hops; the reference's streaming state needs chunks >= 768 samples),
        # eager launches vs one hipGraph replay per chunk
        from acids_transforms_amd.streaming import StreamingDGTSession
        S, C = args.streams, 1024
        chunk = torch.randn(S, C, device=dev, generator=gen) * 0.1
        rtres = {"streams": S, "chunk_samples": C, "realtime_budget_ms": C / SR * 1e3,
                 "mel": "%d log1p mel features per analysed frame inside the step (fp32 banded projection; a bf16 "
                        "bank would break the 1e-5 parity bar)" % N_MELS}
        for tag, use_graph in (("eager", False), ("hipgraph", True)):
            sess = StreamingDGTSession(S, C, N_FFT, HOP, SR, device=dev, use_graph=use_graph, mel_bands=N_MELS)
            for _ in range(3):
                sess.step(chunk)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            nst = 50
            for _ in range(nst):
                sess.step(chunk)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t1) / nst
            rtres["ms_per_chunk_" + tag] = dt * 1e3
            rtres["frames_per_s_" + tag] = S * (C // HOP) / dt
            del sess
        return rtres

    def extra_phase_repr():
        # the stft+polar chain's phase side at the same size; HBM-bound scans along time
        from acids_transforms_amd import ops as _ops
        Xs = stft(x)
        res = {}

        def timed(fn, n=5):
            fn()
            torch.cuda.synchronize()
            e0, e1 = ev(), ev()
            e0.record()
            for _ in range(n):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / n

        def entry(ms, bytes_per_bin):
            a = frames_per_step * F_BINS * bytes_per_bin / (ms * 1e-3) / 1e9
            return {"ms": round(ms, 4), "achieved_GBps": round(a, 1), "frac_of_8TBps": round(a / HBM_PEAK_GBS, 4),
                    "algorithmic_bytes_per_bin": bytes_per_bin}

        res["phase_angle"] = entry(timed(lambda: _ops.phase_scan(Xs, "angle")), 12)
        res["if_forward"] = entry(timed(lambda: _ops.phase_scan(Xs, "forward")), 12)
        inst = _ops.phase_scan(Xs, "forward")
        res["if_invert_forward"] = entry(timed(lambda: _ops.phase_integrate(inst, "forward")), 8)
        res["if_invert_central"] = entry(timed(lambda: _ops.phase_integrate(inst, "central")), 8)
        mg_ = Xs.abs()
        res["polar_to_complex"] = entry(timed(lambda: _ops.polar_to_complex(mg_, inst)), 16)
        res["note"] = "angle+unwrap+finite difference(+Normalize) fused, one thread per (clip, bin) column"
        return res

    def extra_mfcc40():
        # BASELINE config 4's per-GPU work: audio -> log-mel (fused STFT kernel) -> DCT-II, 40 coefficients
        mf = A.MFCC(sr=SR, n_fft=N_FFT, hop_length=HOP, n_mels=N_MELS, n_mfcc=40).to(dev)
        for _ in range(2):
            mf(x)
        torch.cuda.synchronize()
        e0, e1 = ev(), ev()
        e0.record()
        for _ in range(5):
            mf(x)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        return {"ms": round(ms, 4), "frames_per_s": frames_per_step / (ms * 1e-3),
                "note": "MFCC(n_mfcc=40) forward, 1024 clips: fused STFT+log-mel kernel + DCT projection (extension: the "
                        "reference's MFCC class has no DCT)"}

    def extra_hbm_probe():
        # what this box's HBM gives a flat streaming kernel today (a measured ceiling next to the spec)
        n = 1 << 29                                   # 2 GiB of fp32
        a = torch.empty(n, device=dev)
        b = torch.empty(n, device=dev)

        def timed(fn, reps=6):
            fn()
            fn()
            torch.cuda.synchronize()
            e0, e1 = ev(), ev()
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps * 1e-3

        t_fill = timed(lambda: a.fill_(1.0))
        t_copy = timed(lambda: b.copy_(a))
        t_read = timed(lambda: a.sum())
        return {"copy_GBps": round(2 * 4 * n / t_copy / 1e9, 1), "fill_GBps": round(4 * n / t_fill / 1e9, 1),
                "read_sum_GBps": round(4 * n / t_read / 1e9, 1),
                "note": "torch fill_/copy_/sum on 2 GiB fp32 buffers: the practical ceiling for the fractions above "
                        "(roofline.peak stays the 8 TB/s spec)"}

    def extra_other_hops():
        # n_fft = 1024 at the other hops the sliding kernels cover (hop 128: 1379 frames per clip, 5.8 GB of spectrum)
        res = {}
        for hop in (128, 512):
            st = A.STFT(sr=SR, n_fft=N_FFT, hop_length=hop).to(dev)
            Xh = st(x)
            st.invert(Xh)
            torch.cuda.synchronize()
            e = [ev() for _ in range(3)]
            e[0].record()
            for _ in range(3):
                Xh = st(x)
            e[1].record()
            for _ in range(3):
                st.invert(Xh)
            e[2].record()
            torch.cuda.synchronize()
            f_ms, i_ms = e[0].elapsed_time(e[1]) / 3, e[1].elapsed_time(e[2]) / 3
            frames = B * Xh.shape[-2]
            res["hop_%d" % hop] = {"frames_per_clip": int(Xh.shape[-2]), "forward_ms": round(f_ms, 4), "inverse_ms": round(i_ms, 4),
                                   "frames_per_s_fwd_plus_inv": frames / ((f_ms + i_ms) * 1e-3)}
            del Xh
        return res

    def extra_griffin_lim():
        # STFT's default inversion mode (stft.py:37, 174-178): 30 iterations of {forward, phase update + inverse}
        m = stft(x).abs()
        stft.invert(m)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        stft.invert(m)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        return {"seconds": dt, "frames_per_s": frames_per_step / dt, "iterations": 30,
                "note": "phase update fused into the inverse kernel's load stage; ~11.6 GB of HBM traffic per iteration"}

    if not args.no_extras:
        if rank == 0:
            guarded("hbm_probe", extra_hbm_probe)
            guarded("other_hops", extra_other_hops)
            guarded("griffin_lim_invert", extra_griffin_lim)
            guarded("phase_representations", extra_phase_repr)
            guarded("mfcc40_forward", extra_mfcc40)
        if world > 1 and not rehearsal:
            guarded("with_feature_allgather_frames_per_s", extra_allgather)
            guarded("with_feature_allgather_bf16_wire_frames_per_s", lambda: extra_allgather(torch.bfloat16))
        if rank == 0 and args.pghi_clips > 0:
            guarded("pghi_invert", extra_pghi)
        if rank == 0 and args.streams > 0:
            guarded("realtime_dgt_stream", extra_stream)
        barrier()

    result = {
        "metric": "spectrogram frames/sec (fwd+invert), n_fft=1024 hop=256",
        "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "configs[1]: batch=%d clips/GPU x 4 s mono 44.1 kHz, STFT fwd + Magnitude(mel=128, log1p, "
                               "unipolar)%s + ISTFT invert, fp32" % (B, " [one fused kernel]" if fused else ""),
                   "n_fft": N_FFT, "hop": HOP, "frames_per_clip": T_FRAMES, "clips_per_gpu": B,
                   "sharding": "clips, no data-path collective"},
        "roofline": roof,
        "kernels": kernels,
    }
    result.update(extras)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline()
        result["cpu_baseline_pghi"] = cpu_baseline_pghi()
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
