#!/usr/bin/env python3
"""bench.py -- spectrogram frames/s (fwd + invert), n_fft=1024 hop=256 (BASELINE.json metric).

One "step" = one pass of the hot path over one batch of synthetic audio that is
already resident in HBM:
    X, feat = STFT.forward(x) + Magnitude(mel, n_mels=128).forward(X)   one fused kernel:
              X (B, 690, 513) complex64 and feat (B, 690, 128) float32 (log1p + unipolar normalise)
    y       = STFT.invert(X)                     (B, 176384)   float32
(--unfused runs the two forward stages as separate kernels)
on BASELINE configs[1]: batch = 1024 clips x 4 s @ 44.1 kHz mono per GPU, fp32.

`--gpus N` with N > 1: when not already running under a launcher (no WORLD_SIZE in the
environment) this process starts N ranks itself -- one child process per GPU, before anything
touches the GPU -- relays rank 0's JSON line and exits with the children's status.  Under
`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` the ranks are the
launcher's; a WORLD_SIZE that disagrees with --gpus is an error, never a silent N=1.
Clips are sharded over ranks, weak scaling, no data-path collective; `value` is the whole-job
frames/s.  `--pipeline config4` times BASELINE configs[3] instead (per GPU: 1024 clips -> fused
STFT+mel128 -> MFCC(40); features reassembled with an RCCL all-gather) and reports its three
figures: compute only, + all-gather fp32, + all-gather with a bf16 wire format.

Side figures (per-kernel roofline, CPU baselines, PGHI round trip, streaming matrix, parity spot
check of what the timed loop produced) ride along in the same JSON line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=1024, help="clips per GPU")
    ap.add_argument("--pipeline", choices=("step", "config4"), default="step",
                    help="step: configs[1] fwd+mel+invert (the headline); config4: STFT+mel128+MFCC(40) + all-gather")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--unfused", action="store_true", help="STFT and Magnitude as two kernels")
    ap.add_argument("--no-extras", action="store_true", help="skip the side measurements")
    ap.add_argument("--settle-steps", type=int, default=-1,
                    help="untimed steps run BEFORE the --warmup steps so that the timed region sees the chip's sustained "
                         "clocks (a fresh process runs its first two steps at boost clock, is then clamped ~35 %% below by "
                         "the power controller and recovers over 25 to 60 steps: tools/ramp_probe.py); -1 = until three "
                         "consecutive 10-step averages agree within 1.5 %% (at least 40, at most 600 steps); 0 = none")
    ap.add_argument("--pghi-clips", type=int, default=1024, help="clips for the DGT+PGHI round-trip side measurement")
    ap.add_argument("--streams", type=int, default=256, help="concurrent streams for the RealtimeDGT side measurement")
    ap.add_argument("--stream-steps", type=int, default=1000, help="steps per cell of the streaming matrix")
    ap.add_argument("--dry-plan", action="store_true",
                    help="print, WITHOUT touching a GPU or importing torch, what `--gpus N` will do: rank -> device map, clip "
                         "ranges per rank, wire bytes per rank of the three configs[3] figures, timeouts, the keys of the line")
    return ap.parse_args()


# ----------------------------------------------------------------------------------------------
# rank launcher: the parent never touches the GPU and never re-execs
# ----------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def visible_gpu_census():
    """(count, definite): GPUs this process could open, counted WITHOUT any HIP / torch call: KFD topology nodes that have
    SIMDs and whose render node is accessible, cut down by ROCR_ / HIP_ / CUDA_VISIBLE_DEVICES the way the runtime
    applies them (a list of indices or UUIDs; the list ends at the first negative entry).  An over-count is harmless: a
    rank whose device does not exist fails on its own (`no ROCm device` / `LOCAL_RANK ... not visible`, exit 3) and the
    parent relays it.  An UNDER-count would refuse a valid job, so the count is only `definite` when it rests on
    something this function could actually read: a visibility mask, or a KFD topology with at least one readable GPU
    node (a missing / masked /sys/class/kfd, or zero nodes while /dev/kfd exists, is "unknown", not "none")."""
    import glob
    n = 0
    readable = 0
    for prop in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        try:
            kv = dict(line.split(None, 1) for line in open(prop).read().splitlines() if " " in line)
        except OSError:
            continue
        readable += 1
        if int(kv.get("simd_count", "0")) <= 0:
            continue                                           # a CPU node
        minor = int(kv.get("drm_render_minor", "-1"))
        if minor >= 0 and not os.access("/dev/dri/renderD%d" % minor, os.R_OK | os.W_OK):
            continue                                           # not handed to this container
        n += 1
    definite = readable > 0 and (n > 0 or not os.path.exists("/dev/kfd"))
    masked = None
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        val = os.environ.get(var)
        if val is None:
            continue
        listed = 0
        for item in val.split(","):
            item = item.strip()
            if not item or item.startswith("-"):
                break
            listed += 1
        masked = listed if masked is None else min(masked, listed)
    if masked is not None:
        # a mask is an upper bound whatever the topology says (HIP indices are relative to ROCR's list: min() of the
        # list lengths can only over-count)
        return (min(n, masked) if definite else masked), True
    return n, definite


def visible_gpu_count() -> int:
    return visible_gpu_census()[0]


HOOK_VARS = ("ACIDS_BENCH_CORRUPT", "ACIDS_BENCH_INJECT_FAILURE", "ACIDS_BENCH_FORCE_DIST", "ACIDS_BENCH_REHEARSAL")


def hooks_armed():
    """Test-only switches of this program that are set in the environment: printed into the JSON line, so that a line
    measured with one of them armed cannot pass for a clean one."""
    return sorted(v for v in HOOK_VARS if os.environ.get(v))


def dist_timeout_s():
    return int(os.environ.get("ACIDS_BENCH_DIST_TIMEOUT", "180"))


def dry_plan(args):
    """What `python bench.py --gpus N [--pipeline config4]` will do, computed from the arguments alone (no torch, no
    HIP): the first-run-proof sheet for a node nobody has been able to rehearse on."""
    n, B = args.gpus, args.batch
    T, F, n_mels, n_mfcc = 1 + (4 * 44100) // 256, 513, 128, 40
    ndev, definite = visible_gpu_census()
    feat_elems = B * T * (n_mels + n_mfcc)
    plan = {
        "dry_plan": True, "n_gpus": n, "visible_devices": ndev, "visible_devices_definite": definite,
        "launcher": "this process starts %d children (RANK = LOCAL_RANK = 0..%d, MASTER_ADDR 127.0.0.1, a free MASTER_PORT) "
                    "unless WORLD_SIZE is already set (torch.distributed.run), in which case the launcher's ranks are used" % (n, n - 1),
        "backend": "nccl (RCCL over xGMI), init_process_group(device_id=cuda:LOCAL_RANK, timeout=%d s)" % dist_timeout_s(),
        "timeout_s": dist_timeout_s(),
        "environment": {"HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0 (default set by the ranks)")},
        "rank_to_device": {str(r): "cuda:%d" % r for r in range(n)},
        "scaling": "weak: %d clips x 4 s per GPU, %d clips in all" % (B, n * B),
        "clip_ranges": {str(r): [r * B, (r + 1) * B] for r in range(n)},
        "frames_per_rank_per_step": B * T,
        "headline_step": {"per_rank": "fused STFT + Magnitude(mel=128) -> ISTFT on the rank's own clips; no data-path collective",
                          "collectives_in_timed_region": ["all_reduce(MIN) of the ranks' verdicts: the barrier on both sides",
                                                          "all_reduce(MAX) of the elapsed time"],
                          "value": "n_gpus x clips x 690 x steps / max-over-ranks elapsed"},
        "config4": {"workload": "configs[3]: %d clips in all (%d per GPU): fused STFT + log-mel128 (spectrum never stored) -> "
                                "DCT-II 40, features reassembled with all_gather_into_tensor" % (n * B, B),
                    "figures": ["compute_only", "with_allgather_fp32", "with_allgather_bf16_wire", "with_allgather_mfcc40_only_fp32"],
                    "wire_bytes_sent_per_rank_per_step": {
                        "with_allgather_fp32": 4 * feat_elems, "with_allgather_bf16_wire": 2 * feat_elems,
                        "with_allgather_mfcc40_only_fp32": 4 * B * T * n_mfcc},
                    "gathered_bytes_per_rank_per_step": {
                        "with_allgather_fp32": 4 * feat_elems * n, "with_allgather_bf16_wire": 2 * feat_elems * n,
                        "with_allgather_mfcc40_only_fp32": 4 * B * T * n_mfcc * n},
                    "where": "the headline line carries them under `config4` (run on all ranks before any rank-0-only leg); "
                             "`--pipeline config4` makes with_allgather_fp32 the line's `value`"},
        "commands": {"driver": "python -m torch.distributed.run --nnodes=1 --nproc-per-node %d --master-addr 127.0.0.1 --master-port P "
                               "bench.py --gpus %d --steps K --warmup W" % (n, n),
                     "self_launch": "python bench.py --gpus %d --steps K --warmup W" % n},
        "line_keys": ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                      "vs_baseline", "dtype", "data", "config", "world_size_observed", "backend", "roofline", "kernels",
                      "config4", "hooks_armed", "verification"],
        "exit_status": {"0": "measured and verified", "2": "WORLD_SIZE != --gpus / no ROCm device", "3": "fewer devices than ranks",
                        "4": "a side leg raised", "5": "a spot check of timed output failed", "6": "a rank failed inside a timed region"},
        "hooks_armed": hooks_armed(),
    }
    if definite and ndev < n:
        plan["would_refuse"] = "--gpus %d asked for, %d ROCm device(s) visible" % (n, ndev)
    return plan


def launch_ranks(args) -> int:
    n = args.gpus
    rehearsal = os.environ.get("ACIDS_BENCH_REHEARSAL") == "1"
    ndev, definite = visible_gpu_census()     # sysfs only: the parent neither imports torch nor touches HIP
    if not rehearsal and definite and ndev < n:
        print(json.dumps({"error": "--gpus %d asked for, %d ROCm device(s) visible: refusing to report a smaller "
                                   "job under that name" % (n, ndev), "n_gpus_visible": ndev}))
        return 3
    port = int(os.environ.get("MASTER_PORT", "0")) or _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    alive = list(procs)
    while alive:
        time.sleep(0.2)
        for p in list(alive):
            code = p.poll()
            if code is None:
                continue
            alive.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in alive:               # one rank failed: the job failed; stop exactly the ranks we started
                    q.terminate()
    for p in procs:
        try:
            p.wait(timeout=30)
        except subprocess.TimeoutExpired:
            p.kill()
    return rc if rc >= 0 else 1




if __name__ == "__main__" and "WORLD_SIZE" not in os.environ:
    # The rank launcher runs HERE, before torch is imported: a parent that starts N children must be a process that
    # has provably never initialised the GPU (torch.cuda.device_count() falls back to hipGetDeviceCount() when amdsmi
    # is unavailable, and a process that has made a HIP call must not start further GPU processes on this pool).
    _args = parse()
    if _args.dry_plan:
        print(json.dumps(dry_plan(_args)))
        sys.exit(0)
    if _args.gpus > 1:
        sys.exit(launch_ranks(_args))

if int(os.environ.get("WORLD_SIZE", "1")) > 1:
    # Multi-process GPU work on this pool: the host driver supports dmabuf IPC only, and RCCL's peer buffers (and any
    # cross-process sharing of device memory) fail with `hipIpcGetMemHandle: invalid argument` unless the HSA runtime is
    # told so BEFORE it initialises.  The image exports this already; a default only (never overrides the caller), set
    # here so that a rank started by any launcher has it before torch loads the runtime.
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

N_FFT, HOP, SR = 1024, 256, 44100
CLIP_LEN = 4 * SR                     # 176400 samples
T_FRAMES = 1 + CLIP_LEN // HOP        # 690
F_BINS = N_FFT // 2 + 1               # 513
N_MELS = 128
N_MFCC = 40
HBM_PEAK_GBS = 8000.0                 # MI355X HBM3E spec (MI355X_MICROARCH.md: 8 TB/s, ~6.3 achievable)
MFMA_F32_PEAK_TFLOPS = 157.3          # fp32-input MFMA spec
MFMA_BF16_PEAK_TFLOPS = 2500.0        # dense bf16 MFMA spec
# algorithmic bytes per frame
BYTES_STFT_FWD = HOP * 4 + F_BINS * 8            # 5128
BYTES_ISTFT = F_BINS * 8 + HOP * 4               # 5128
BYTES_MEL = F_BINS * 8 + N_MELS * 4              # 4616 (unfused: reads the complex spectrum)
BYTES_FUSED_FEATURES_ONLY = HOP * 4 + N_MELS * 4  # 1536 (spectrum never stored)
FLOPS_MEL = 2 * F_BINS * N_MELS                  # 131328 dense


# ----------------------------------------------------------------------------------------------
# socket power / shader clock next to the timed kernels (sysfs hwmon of the device; plain file reads from a side thread)
# ----------------------------------------------------------------------------------------------
def device_hwmon(index):
    """hwmon directory of HIP device `index` (PCI bus id from the runtime -> /sys/bus/pci/devices/<bdf>/hwmon/hwmon*), or None."""
    import ctypes
    import glob
    try:
        hip = ctypes.CDLL("libamdhip64.so")
        buf = ctypes.create_string_buffer(64)
        if hip.hipDeviceGetPCIBusId(buf, 64, index) != 0:
            return None
        hits = glob.glob("/sys/bus/pci/devices/%s/hwmon/hwmon*" % buf.value.decode().lower())
        return hits[0] if hits and os.path.exists(os.path.join(hits[0], "power1_input")) else None
    except Exception:
        return None


class PowerSampler:
    """(time, watts, sclk MHz) every 10 ms while running.  The step is power-managed: at 1024 clips both kernels sit on the
    socket's power cap and the shader clock is what the controller leaves them (profiles/r05_power_clock.md)."""

    def __init__(self, hwmon, period=0.01):
        import threading
        self.hwmon, self.period, self.rows, self.stop = hwmon, period, [], False
        self.thread = threading.Thread(target=self._run, daemon=True)

    def _run(self):
        try:
            fp = open(os.path.join(self.hwmon, "power1_input"))
            ff = open(os.path.join(self.hwmon, "freq1_input"))
        except OSError:
            return
        while not self.stop:
            try:
                fp.seek(0)
                ff.seek(0)
                self.rows.append((time.perf_counter(), int(fp.read()) * 1e-6, int(ff.read()) * 1e-6))
            except (OSError, ValueError):
                pass
            time.sleep(self.period)

    def __enter__(self):
        if self.hwmon:
            self.thread.start()
        return self

    def __exit__(self, *exc):
        self.stop = True
        return False

    def window(self, t0, t1):
        rows = [r for r in self.rows if t0 <= r[0] <= t1]
        if not rows:
            return None
        return {"samples": len(rows), "watts": sum(r[1] for r in rows) / len(rows), "sclk_mhz": sum(r[2] for r in rows) / len(rows),
                "sclk_mhz_min": min(r[2] for r in rows)}

    def cap_watts(self):
        try:
            return int(open(os.path.join(self.hwmon, "power1_cap")).read()) * 1e-6
        except Exception:
            return None


# ----------------------------------------------------------------------------------------------
# CPU baselines (the oracle timed on this box's host cores; rank 0, N = 1 only)
# ----------------------------------------------------------------------------------------------
def host_cpus():
    """Host threads this process may actually use: the affinity mask, capped by the cgroup CPU quota (a GPU box hands
    one GPU's job a share of the host, not all of os.cpu_count())."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = float(q) / float(period)
    except Exception:
        pass
    usable = n if quota is None else max(1, min(n, int(quota + 0.5)))
    return {"os_cpu_count": os.cpu_count(), "affinity": n, "cgroup_quota": quota, "usable": usable}


def cpu_baseline(threads, sample_clips, budget_s=10.0):
    """The oracle (CPU restatement: torch CPU stft + matmul + istft) on `threads` host threads, on a bounded
    sample of the same workload: STFT fwd + Magnitude(mel128) + ISTFT."""
    from oracle import oracle as O
    prev = torch.get_num_threads()
    torch.set_num_threads(threads)
    try:
        g = torch.Generator().manual_seed(1234)
        x = torch.randn(sample_clips, CLIP_LEN, generator=g) * 0.1
        w = O.hann_window(N_FFT)
        fwd, _ = O.magnitude_banks(O.melscale_fbanks(F_BINS, 0.0, SR / 2, N_MELS, SR))
        X = O.stft_forward(x[:2], w, N_FFT, HOP)
        off, sc = O.magnitude_scale_stats(X, "log1p", "unipolar")

        def step():
            X = O.stft_forward(x, w, N_FFT, HOP)
            O.magnitude_forward(X, fwd, "log1p", off, sc)
            O.istft(X, w, N_FFT, HOP)

        step()
        t0 = time.perf_counter()
        n = 0
        while n < 2 or (time.perf_counter() - t0 < budget_s and n < 50):
            step()
            n += 1
        dt = (time.perf_counter() - t0) / n
    finally:
        torch.set_num_threads(prev)
    return {"value": sample_clips * T_FRAMES / dt, "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": "%d clips x 4 s (%d frames) per pass, %d passes, oracle = torch CPU stft+matmul+istft, %d threads "
                      "(os.cpu_count() = %d)" % (sample_clips, sample_clips * T_FRAMES, n, threads, os.cpu_count())}


def synth_tonal(n_clips, length, seed=7, device="cpu"):
    """SURVEY 8d C3's tonal set: 8 random-frequency exponentially decaying sinusoids per clip."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    f = 50.0 + 8000.0 * torch.rand(n_clips, 8, 1, generator=g)
    a = 0.02 + 0.1 * torch.rand(n_clips, 8, 1, generator=g)
    d = 0.3 + 3.0 * torch.rand(n_clips, 8, 1, generator=g)
    ph = 6.2831853 * torch.rand(n_clips, 8, 1, generator=g)
    f, a, d, ph = (t.to(device) for t in (f, a, d, ph))
    t = torch.arange(length, device=device, dtype=torch.float32) / SR
    out = torch.empty(n_clips, length, device=device)
    for lo in range(0, n_clips, 64):
        sl = slice(lo, lo + 64)
        out[sl] = (a[sl] * torch.exp(-d[sl] * t) * torch.sin(6.2831853 * f[sl] * t + ph[sl])).sum(1)
    return out


def cpu_baseline_pghi(mags, thread_counts):
    """oracle/pghi_ref.c (exact-order C PGHI), one clip per host thread, on magnitudes computed by the device
    DGT (the very inputs of the GPU figure)."""
    from oracle import oracle as O
    out = {}
    for tag, mag in mags.items():
        if mag is None:
            continue
        m = mag.numpy()
        clips, frames = m.shape[0], m.shape[1]
        res = {}
        for th in [1] + sorted(set(thread_counts)):
            nclips = min(clips, 4 if tag == "noise" else 32) if th == 1 else clips
            t0 = time.perf_counter()
            _, pops = O.pghi_offline_batch(m[:nclips], N_FFT, HOP, threads=th, want_pops=True)
            dt = time.perf_counter() - t0
            res["threads_%d" % th] = {
                "value": nclips * frames / dt, "unit": "frames/s", "cores": th, "kind": "port", "heap_pops_per_s": pops / dt,
                "seconds": dt, "sample": "%d clips x %d frames, %s magnitudes, exact-heap C PGHI, %d thread(s), one clip per "
                                         "thread at a time" % (nclips, frames, tag, th)}
        best = max((k for k in res if k != "threads_1"), key=lambda k: res[k]["value"])
        res["all_cores"] = res[best]
        out[tag] = res
    return out


def cpu_baseline_northstar(samples, threads):
    """The north-star chain on the host: Mono -> DGT -> |.| -> mel bank -> log1p -> unipolar normalise, and back
    (de-normalise, expm1, inverse bank, PGHI, polar ISTFT with the dual window), as the oracle states it: torch CPU ops
    with `threads` threads + the exact-order C PGHI, one clip per thread.  `samples`: {tag: (clips, 2, L) stereo audio}.
    A bounded sample of the GPU leg's own inputs."""
    from oracle import oracle as O
    prev = torch.get_num_threads()
    torch.set_num_threads(threads)
    out = {}
    try:
        win = O.gauss_window(N_FFT)
        dual = O.dual_window(win, N_FFT, HOP)
        fwd, inv = O.magnitude_banks(O.melscale_fbanks(F_BINS, 0.0, SR / 2, F_BINS, SR))   # the reference's 513-filter bank
        for tag, xs in samples.items():
            n = xs.shape[0]
            t0 = time.perf_counter()
            mono = xs.sum(-2) / 2
            X = O.stft_forward(mono, win, N_FFT, HOP)
            off, sc = O.magnitude_scale_stats(X, "log1p", "unipolar")
            y = O.magnitude_forward(X, fwd, "log1p", off, sc)
            t1 = time.perf_counter()
            mag = O.magnitude_invert(y, inv, "log1p", off, sc)
            phase = O.pghi_offline_batch(mag.numpy(), N_FFT, HOP, threads=threads)
            audio = O.polar_istft(mag, torch.from_numpy(phase), dual, N_FFT, HOP)
            t2 = time.perf_counter()
            del audio
            out[tag] = {"value": n * T_FRAMES / (t2 - t0), "unit": "frames/s", "cores": threads, "kind": "port",
                        "forward_s": t1 - t0, "invert_s": t2 - t1,
                        "sample": "%d stereo clips x 4 s, %s; oracle chain (torch CPU + exact-heap C PGHI), %d threads"
                                  % (n, tag, threads)}
    finally:
        torch.set_num_threads(prev)
    return out


# genuine-reference composite, extrapolated from BASELINE.md section 2 (reference imported in the build container, 8
# vCPU): forward STFT+Magnitude 124 k frames/s; Magnitude.invert taken as its forward (294 k); torch.istft 311 k;
# DGT.invert(pghi) 12.7 frames/s on dense noise, 753 frames/s on a tonal clip (pure-Python heap, one core)
REFERENCE_COMPOSITE = {
    "noise": 1.0 / (1 / 124e3 + 1 / 294e3 + 1 / 12.7 + 1 / 311e3),
    "tonal": 1.0 / (1 / 124e3 + 1 / 294e3 + 1 / 753.0 + 1 / 311e3),
}


# ----------------------------------------------------------------------------------------------
# what the exit status says (pure functions: tests/test_bench_cpu.py feeds them result dictionaries)
# ----------------------------------------------------------------------------------------------
def verification_failures(result):
    """Paths of every `*spot_check` entry of the result whose `ok` is not true: the bench checks what its timed calls
    produced against the oracle (parity_spot_check, pghi_invert.*_spot_check), and a number on unverified output must
    not leave with status 0."""
    bad = []

    def walk(node, path):
        if isinstance(node, dict):
            for k, v in node.items():
                here = path + [str(k)]
                if str(k).endswith("spot_check") and isinstance(v, dict) and v.get("ok") is not True:
                    bad.append(".".join(here))
                walk(v, here)

    walk(result, [])
    return sorted(bad)


def exit_status(result):
    """0: every figure measured and verified.  5: a spot check of a timed result failed (takes precedence: the headline
    itself is in doubt).  4: a side measurement raised (`*_error` key); the line is complete either way."""
    if verification_failures(result):
        return 5
    if any(str(k).endswith("_error") for k in result):
        return 4
    return 0


class LegFailed(RuntimeError):
    """A measurement leg that holds collectives failed on some rank; raised on EVERY rank (see all_ok)."""


# ----------------------------------------------------------------------------------------------
def main():
    args = parse()
    if args.dry_plan:
        print(json.dumps(dry_plan(args)))
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))          # (imported and called as a function; the script path launched above)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(json.dumps({"error": "--gpus %d but WORLD_SIZE=%d: start bench.py as `python bench.py --gpus N` or under "
                                       "torch.distributed.run with --nproc-per-node equal to --gpus" % (args.gpus, world)}))
        sys.exit(2)
    if not torch.cuda.is_available():
        print(json.dumps({"error": "no ROCm device: bench.py measures the HIP path only"}))
        sys.exit(2)
    rehearsal = os.environ.get("ACIDS_BENCH_REHEARSAL") == "1"   # dev only: N ranks share cuda:0 over gloo
    if rehearsal:
        local_rank = 0
    if local_rank >= torch.cuda.device_count():
        print(json.dumps({"error": "rank %d: LOCAL_RANK %d but only %d ROCm device(s) visible: refusing to run a smaller "
                                   "job under the name --gpus %d" % (rank, local_rank, torch.cuda.device_count(), args.gpus)}))
        sys.exit(3)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    backend = None
    # dev only: ACIDS_BENCH_FORCE_DIST=1 runs the RCCL code paths (process group, barriers, max-over-ranks, the config-4
    # all-gathers) with a world of ONE rank, which a one-GPU box can do -- an API rehearsal, not a measurement
    use_dist = world > 1 or os.environ.get("ACIDS_BENCH_FORCE_DIST") == "1"
    if use_dist and world == 1:
        os.environ.setdefault("MASTER_PORT", str(_free_port()))
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL prints a version banner on STDOUT when its communicator comes up; stdout is for the one JSON line, so the
        # group is created (and its communicator forced up by a first barrier) with fd 1 pointing at stderr
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            # an explicit timeout: a rank that dies (or never issues its half of a collective) turns into an error on
            # the others after this long instead of a hang -- the parent / torchrun then ends the job non-zero
            import datetime
            limit = datetime.timedelta(seconds=dist_timeout_s())
            if rehearsal:
                dist.init_process_group("gloo", timeout=limit)
            else:
                dist.init_process_group("nccl", device_id=dev, timeout=limit)
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)
        backend = dist.get_backend()
        assert dist.get_world_size() == world
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
    mfcc = A.MFCC(sr=SR, n_fft=N_FFT, hop_length=HOP, n_mels=N_MELS, n_mfcc=N_MFCC).to(dev)

    ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731
    fused = (not args.unfused) and mag.can_fuse_with(stft, x)
    config4 = args.pipeline == "config4"

    # dev only: ACIDS_BENCH_INJECT_FAILURE="<rank>:<leg>" makes that rank raise inside that leg ("step" | "config4"):
    # the rehearsal test of the N > 1 control flow (tests/test_bench_gpu.py)
    inject = os.environ.get("ACIDS_BENCH_INJECT_FAILURE", "")
    inject_rank, inject_leg = (int(inject.split(":")[0]), inject.split(":")[1]) if ":" in inject else (-1, "")

    def maybe_fail(leg):
        if rank == inject_rank and leg == inject_leg:
            raise RuntimeError("injected failure on rank %d in leg %s" % (rank, leg))

    def all_ok(ok=True):
        """The barrier of every leg that holds collectives, carrying each rank's verdict: an all-reduce(MIN) of `ok`.
        A rank whose share of a leg raised still arrives here, so the others are never left inside a collective; if any
        rank failed, every rank learns it at the same point and leaves the leg together (LegFailed on all of them)."""
        if not use_dist:
            return ok
        tt = torch.tensor([1 if ok else 0], device=red_dev, dtype=torch.int32)
        dist.all_reduce(tt, op=dist.ReduceOp.MIN)
        return bool(int(tt.item()))

    def max_over_ranks(seconds):
        if use_dist:
            tt = torch.tensor([seconds], device=red_dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            return float(tt.item())
        return seconds

    def timed_region(fn, steps, warmup, leg="leg"):
        """W untimed steps, then exactly K steps between barrier + synchronize on both sides; max over ranks.  The two
        barriers are all_ok(): a failure on one rank ends the leg on all of them."""
        err = None
        try:
            for _ in range(warmup):
                fn()
            torch.cuda.synchronize()
        except Exception as exc:      # noqa: BLE001 -- reported, and every rank must still reach the collective below
            err = exc
        if not all_ok(err is None):
            raise LegFailed("%s: %s" % (leg, repr(err) if err is not None else "another rank failed during warm-up"))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        try:
            for _ in range(steps):
                fn()
            torch.cuda.synchronize()
        except Exception as exc:      # noqa: BLE001
            err = exc
        ok = all_ok(err is None)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if not ok:
            raise LegFailed("%s: %s" % (leg, repr(err) if err is not None else "another rank failed in the timed region"))
        return max_over_ranks(dt)

    # -- the headline step ------------------------------------------------------------------------
    ktimes = {"stft_fwd": [], "mel": [], "istft": []}
    last = {}

    def step(record=False):
        # The previous step's outputs are released BEFORE this step allocates its own, so every step writes the same three
        # blocks of the caching allocator.  Holding them across the call (as this function used to) makes the allocator
        # alternate between two sets of blocks, and the sets are not equally fast: same kernels, same box, 0.81-0.84 ms
        # into one spectrum buffer and 0.87-0.89 ms into the other (tools/placement_probe.py; a process's earlier
        # allocations are the faster ones, tools/placement_probe2.py: 0.677 / 0.685 / 0.685 / 0.703 ms for four buffers
        # in allocation order) -- the timed steps then alternated 1.49 / 1.60 ms.
        last.clear()
        maybe_fail("step")
        e = None
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
        last["X"], last["feat"], last["y"] = X, feat, y
        return e

    # -- BASELINE configs[3]: per GPU 1024 clips -> fused STFT + mel128 (features only) -> MFCC(40) -----------
    def config4_compute():
        """log-mel (B, T, 128) from the fused kernel (the spectrum never goes to HBM) and its DCT-II, 40
        coefficients, channel-major (B, 40, T) like the reference's MFCC layout."""
        from acids_transforms_amd import ops as _ops
        maybe_fail("config4")
        _, _, logmel = _ops.stft_mel_forward(x, mfcc.window, mfcc._band, "log", None, None, eps=1e-10, power=2,
                                             want_spectrum=False)
        coef = _ops.mel_forward_real(logmel, mfcc.dct, None, None, channel_major_T=logmel.shape[-2])
        return logmel, coef

    def config4_figures(steps, warmup):
        from acids_transforms_amd.dist import all_gather_features
        res = {}
        t_compute = timed_region(config4_compute, steps, warmup, "config4 compute")
        res["compute_only"] = {"frames_per_s": world * frames_per_step * steps / t_compute,
                               "ms_per_step": t_compute / steps * 1e3}
        if use_dist:
            # rehearsal (gloo, every rank on cuda:0): gloo gathers host tensors only, so the wire tensor is staged through the
            # host there -- the control flow (side stream, one gather in flight, wire dtype, handles) is the real one
            comm = torch.cuda.Stream(device=dev)

            def with_gather(wire_dtype, what):
                pending = []

                def fn():
                    logmel, coef = config4_compute()
                    done = torch.cuda.Event()
                    done.record()
                    for h in pending:                 # at most one gather in flight behind the next step's compute
                        h.wait()
                    pending.clear()
                    with torch.cuda.stream(comm):
                        comm.wait_event(done)
                        for t in ((logmel, coef) if what == "mel128+mfcc40" else (coef,)):
                            t.record_stream(comm)
                            wire = t if wire_dtype is None else t.to(wire_dtype)
                            if rehearsal:
                                wire = wire.cpu()
                            _, h = all_gather_features(wire, world * B, async_op=True)
                            if h is not None:
                                pending.append(h)

                try:
                    t = timed_region(fn, steps, warmup, "config4 " + what)
                finally:
                    for h in pending:
                        h.wait()
                    pending.clear()
                torch.cuda.synchronize()
                return {"frames_per_s": world * frames_per_step * steps / t, "ms_per_step": t / steps * 1e3}

            res["with_allgather_fp32"] = with_gather(None, "mel128+mfcc40")
            res["with_allgather_bf16_wire"] = with_gather(torch.bfloat16, "mel128+mfcc40")
            res["with_allgather_mfcc40_only_fp32"] = with_gather(None, "mfcc40")
            res["wire_bytes_per_rank_fp32"] = B * T_FRAMES * (N_MELS + N_MFCC) * 4
        res["note"] = ("per GPU: %d clips -> fused STFT+log-mel128 kernel (spectrum not stored) -> DCT-II 40; the gathers "
                       "run on a side stream, overlapped with the next step's compute; timing includes them "
                       "(barrier + synchronize on both sides)" % B)
        return res

    def fail_all_ranks(exc):
        """A collective leg of the headline failed somewhere: every rank is here (LegFailed is raised on all of them),
        rank 0 says so on stdout, everybody exits non-zero."""
        if rank == 0:
            print(json.dumps({"error": "rank failure inside a timed region: %s" % exc, "n_gpus": world, "hooks_armed": hooks_armed()}))
            sys.stdout.flush()
        if use_dist:
            dist.destroy_process_group()
        sys.exit(6)

    if config4:
        try:
            c4 = config4_figures(args.steps, args.warmup)
        except LegFailed as exc:
            fail_all_ranks(exc)
        key = "with_allgather_fp32" if "with_allgather_fp32" in c4 else "compute_only"
        value = c4[key]["frames_per_s"]
        ms_per_step = c4[key]["ms_per_step"]
        result = {
            "metric": "spectrogram frames/sec (STFT+mel128+MFCC40 fwd, features all-gathered), n_fft=1024 hop=256",
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[3]: batch=%d clips total (%d per GPU) x 4 s mono 44.1 kHz, STFT + mel128 + "
                                   "MFCC(40), RCCL all-gather of the features; value = %s" % (world * B, B, key),
                       "n_fft": N_FFT, "hop": HOP, "frames_per_clip": T_FRAMES, "clips_per_gpu": B,
                       "sharding": "clips; all-gather only to reassemble outputs"},
            "world_size_observed": dist.get_world_size() if use_dist else 1, "backend": backend,
            "config4": c4, "hooks_armed": hooks_armed(),
        }
        if rank == 0:
            print(json.dumps(result))
        if use_dist:
            dist.destroy_process_group()
        return

    # settle: the power controller's transient of a fresh process is over before the W warm-up steps start
    # (reported as `settle_steps`; the W warm-up steps and the K timed steps follow as the contract says).  The first
    # W + K steps of the process are what `--settle-steps 0` would have timed: they are clocked on the way
    # (`fresh_process_ms_per_step`, HIP events, no host synchronisation) so that both figures come from one run.
    settled = 0
    fresh_ms = None
    err = None
    try:
        if args.settle_steps != 0:
            fa, fb = ev(), ev()
            for _ in range(args.warmup):
                step()
            fa.record()
            for _ in range(args.steps):
                step()
            fb.record()
            settled = args.warmup + args.steps
        if args.settle_steps > 0:
            for _ in range(max(0, args.settle_steps - settled)):
                step()
            settled = max(settled, args.settle_steps)
        elif args.settle_steps < 0:
            # adaptive: 10-step chunks timed with events; the host waits for chunk k while chunk k + 1 is already queued,
            # so the GPU never idles (an idle gap is itself a transient: the first step after one runs ~12 % slow)
            marks = [ev()]
            marks[0].record()
            means = []
            while settled < 600:
                for _ in range(10):
                    step()
                settled += 10
                marks.append(ev())
                marks[-1].record()
                if len(marks) >= 3:
                    marks[-2].synchronize()
                    means.append(marks[-3].elapsed_time(marks[-2]) / 10.0)
                if settled >= 40 and len(means) >= 3 and max(means[-3:]) <= 1.015 * min(means[-3:]):
                    break
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        if args.settle_steps != 0:
            fresh_ms = fa.elapsed_time(fb) / max(1, args.steps)
    except Exception as exc:          # noqa: BLE001 -- every rank must reach the collective below
        err = exc
    if not all_ok(err is None):
        fail_all_ranks("settle / warm-up: %s" % (repr(err) if err is not None else "another rank failed"))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    evs = []
    try:
        for _ in range(args.steps):
            evs.append(step(record=True))
        torch.cuda.synchronize()
    except Exception as exc:          # noqa: BLE001
        err = exc
    ok = all_ok(err is None)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if not ok:
        fail_all_ranks("timed region: %s" % (repr(err) if err is not None else "another rank failed"))
    elapsed = max_over_ranks(elapsed)
    for e in evs:
        ktimes["stft_fwd"].append(e[0].elapsed_time(e[1]))
        ktimes["mel"].append(e[1].elapsed_time(e[2]))
        ktimes["istft"].append(e[2].elapsed_time(e[3]))
    del evs
    ms_per_step = elapsed / args.steps * 1e3
    value = world * frames_per_step * args.steps / elapsed
    avg = {k: sum(v) / len(v) for k, v in ktimes.items()}
    step_ms_events = [round(a + b + c, 4) for a, b, c in zip(ktimes["stft_fwd"], ktimes["mel"], ktimes["istft"])]

    extras = {}

    def guarded(name, fn):
        """Side measurements must never cost the headline line."""
        try:
            out = fn()
            if out is not None:
                extras[name] = out
        except Exception as exc:
            extras[name + "_error"] = repr(exc)[:300]

    # -- what did the timed loop produce?  clips {0, B/2-1, B-1} of the LAST timed step against the oracle --------
    def parity_spot_check():
        from oracle import oracle as O
        ids = sorted({0, B // 2 - 1 if B > 1 else 0, B - 1})
        xs = x[ids].cpu()
        corrupt = os.environ.get("ACIDS_BENCH_CORRUPT")       # dev only ("X" | "feat" | "y"): the test of the exit status
        if corrupt in last:
            last[corrupt].view(-1)[: last[corrupt].numel() // 2].zero_()
        Xg, fg, yg = (last[k][ids].cpu() for k in ("X", "feat", "y"))
        w = O.hann_window(N_FFT)
        Xr = O.stft_forward(xs, w, N_FFT, HOP)
        fwd, _ = O.magnitude_banks(O.melscale_fbanks(F_BINS, 0.0, SR / 2, N_MELS, SR))
        off, sc = float(mag.norm.offset), float(mag.norm.scale)
        fr = O.magnitude_forward(Xr, fwd, "log1p", off, sc)
        yr = O.istft(Xr, w, N_FFT, HOP)

        def rel(a, b):
            return float((a - b).abs().max() / b.abs().max())

        r = {"X": rel(torch.view_as_real(Xg), torch.view_as_real(Xr)), "feat": rel(fg, fr), "y": rel(yg, yr)}
        return {"clips": ids, "max_rel": max(r.values()), "per_output": r, "tolerance": 1e-5,
                "ok": bool(max(r.values()) < 1e-5),
                "note": "outputs of the last timed step (rank 0) vs the oracle on CPU, max|d|/max|ref|"}

    def hbm_entry(name, bytes_per_frame, ms=None):
        ms = avg[name] if ms is None else ms
        a = frames_per_step * bytes_per_frame / (ms * 1e-3) / 1e9
        return {"kernel": name, "key": name, "bound": "hbm", "achieved": round(a, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(a / HBM_PEAK_GBS, 4), "ms": round(ms, 4), "algorithmic_bytes_per_frame": bytes_per_frame}

    def timed_ms(fn, n=10, warm=3):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        e0, e1 = ev(), ev()
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n

    if fused:
        # one kernel reads the audio and writes spectrum + features: 1024 + 4104 + 512 bytes per frame
        kernels = [hbm_entry("stft_fwd", BYTES_STFT_FWD + 4 * N_MELS), hbm_entry("istft", BYTES_ISTFT)]
        kernels[0]["kernel"] = "stft_fwd+mel (fused)"
        if rank == 0 and world == 1:                             # cheap (~300 launches): kept under --no-extras too
            # 25 launches before the 30 timed ones: a change of kernel is a change of power draw, and the chip's
            # power controller takes ~25 launches to settle on the new sustained clock (tools/ramp_probe.py)
            Xs = stft(x)
            avg["mel"] = timed_ms(lambda: mag(Xs), 30, 25)               # the stand-alone projection, outside the step
            del Xs
            avg["stft_fwd_plain"] = timed_ms(lambda: stft(x), 30, 25)    # framing + FFT only, outside the step
            kernels.append(hbm_entry("stft_fwd_plain", BYTES_STFT_FWD))
            kernels[-1]["kernel"] = "stft_fwd (framing + FFT only, outside the step)"
            from acids_transforms_amd import ops as _ops
            off_, sc_ = mag._affine()
            avg["fwd_features_only"] = timed_ms(lambda: _ops.stft_mel_forward(
                x, stft.window[:N_FFT], mag._banded(), "log1p", off_, sc_, mag._eps, want_spectrum=False), 30, 25)
            kernels.append(hbm_entry("fwd_features_only", BYTES_FUSED_FEATURES_ONLY))
            kernels[-1]["kernel"] = "stft_fwd+mel, features only (spectrum not stored; outside the step)"
            kernels[-1]["note"] = ("literal configs[1] 'fwd': 1536 algorithmic B/frame; VALU/LDS-bound (FFT + band walk per "
                                   "frame), not HBM-bound -- reported against its own HBM roofline for honesty")
            # the reference's DEFAULT bank: Magnitude() builds 513 mel filters (spectral_repr.py:170-189); SURVEY 8d C2
            # asks for both.  Fused spectrum + features: 1024 + 4104 + 2052 B/frame; features only (the README chain's
            # forward, README.md:48-50): 1024 + 2052.
            mag513 = A.Magnitude(sr=SR, n_fft=N_FFT, mode="unipolar", contrast="log1p").to(dev)
            Xs = stft(x[:8])
            mag513.scale_data(Xs)
            del Xs
            if mag513.can_fuse_with(stft, x):
                avg["fused_mel513"] = timed_ms(lambda: mag513.forward_fused(stft, x, return_spectrum=True), 30, 25)
                kernels.append(hbm_entry("fused_mel513", BYTES_STFT_FWD + 4 * F_BINS))
                kernels[-1]["kernel"] = "stft_fwd+mel513 (fused, reference-default 513-filter bank; outside the step)"
                off5, sc5 = mag513._affine()
                avg["fwd_features_only_mel513"] = timed_ms(lambda: _ops.stft_mel_forward(
                    x, stft.window[:N_FFT], mag513._banded(), "log1p", off5, sc5, mag513._eps, want_spectrum=False), 30, 25)
                kernels.append(hbm_entry("fwd_features_only_mel513", HOP * 4 + 4 * F_BINS))
                kernels[-1]["kernel"] = "stft_fwd+mel513, features only (README chain forward; outside the step)"
            del mag513
    else:
        kernels = [hbm_entry("stft_fwd", BYTES_STFT_FWD), hbm_entry("istft", BYTES_ISTFT)]
    if "mel" in avg and avg["mel"] > 0:
        kernels.append(hbm_entry("mel", BYTES_MEL))
        kernels[-1]["kernel"] = "mel (stand-alone banded projection%s)" % (", outside the step" if fused else "")
    dominant = max(("stft_fwd", "istft"), key=lambda k: avg[k])
    roof = dict(kernels[0] if dominant == "stft_fwd" else kernels[1])
    roof.pop("algorithmic_bytes_per_frame")
    roof["traffic"], roof["traffic_source"] = pmc_traffic(dominant if (fused or dominant != "stft_fwd") else "stft_fwd_unfused")
    roof["kernel"] = {"stft_fwd": "stft1024_h256_fwd_kernel (fused mel epilogue)" if fused else "stft1024_h256_fwd_kernel",
                      "istft": "istft1024_ola_kernel"}[dominant]
    # everything else the step and its neighbours run, where the driver's record can see it (it keeps `roofline` whole
    # and only the NAMES of extra keys): the step's other kernel, the whole step, and the forms outside the step
    step_bytes = (BYTES_STFT_FWD + (4 * N_MELS if fused else 0) + BYTES_ISTFT) if fused else (BYTES_STFT_FWD + BYTES_MEL + BYTES_ISTFT)
    step_gbs = frames_per_step * step_bytes / (ms_per_step * 1e-3) / 1e9 if world == 1 else None
    roof["others"] = [{k: v for k, v in kk.items() if k in ("kernel", "achieved", "frac", "ms", "algorithmic_bytes_per_frame")}
                      for kk in kernels if kk["kernel"] != (kernels[0] if dominant == "stft_fwd" else kernels[1])["kernel"]]
    roof["whole_step"] = {"algorithmic_bytes_per_frame": step_bytes, "ms": round(ms_per_step, 4),
                          "achieved": None if step_gbs is None else round(step_gbs, 1),
                          "frac": None if step_gbs is None else round(step_gbs / HBM_PEAK_GBS, 4),
                          "note": "both kernels of the step over the step's wall time (launch gaps included), per GPU"}

    # The same as SCALAR members of `roofline` (the driver's record keeps the scalar members and only the names of
    # everything nested): the step's other kernel, the forms outside the step, the whole step, even / odd steps
    by_key = {kk["key"]: kk for kk in kernels}
    roof.pop("key", None)
    for key, short in (("stft_fwd", "fused_fwd" if fused else "fwd"), ("istft", "istft"), ("stft_fwd_plain", "plain_fwd"),
                       ("fwd_features_only", "features_only"), ("fused_mel513", "fused_mel513"),
                       ("fwd_features_only_mel513", "features_only_mel513"), ("mel", "mel_standalone")):
        if key in by_key:
            roof[short + "_ms"] = by_key[key]["ms"]
            roof[short + "_frac"] = by_key[key]["frac"]
    roof["whole_step_ms"] = round(ms_per_step, 4)
    roof["whole_step_frac"] = None if step_gbs is None else round(step_gbs / HBM_PEAK_GBS, 4)
    ev_, od_ = step_ms_events[0::2], step_ms_events[1::2]
    roof["step_even_ms"] = round(sum(ev_) / len(ev_), 4)
    roof["step_odd_ms"] = round(sum(od_) / len(od_), 4) if od_ else None

    # -- side measurements -----------------------------------------------------------------------------------
    def extra_pattern_copy():
        """What the memory system gives the step's two ACCESS PATTERNS on this box, in this process: copy kernels with the
        forward's (audio in, 512-byte aligned blocks of the spectrum stream + 128 features out) and the inverse's (spectrum
        rows in, hops out) streams and no arithmetic, waves taking 8-frame runs in dispatch order (tools/ubench/pattern_lib.hip,
        the kernels of stream_pattern3.hip).  The ceiling the product kernels are compared with -- not torch's copy_."""
        import ctypes
        so = os.path.join(ROOT, "tools", "ubench", "libpattern.so")
        if not os.path.exists(so):       # normally built by __graft_entry__.build(); three seconds of hipcc otherwise
            subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so,
                                   os.path.join(ROOT, "tools", "ubench", "pattern_lib.hip")], stdout=subprocess.DEVNULL)
        h = ctypes.CDLL(so)
        V, I64, I = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
        h.pat_fwd.argtypes = [V, V, V, I64, I, I, V]
        h.pat_inv.argtypes = [V, V, I64, I, I, V]
        from acids_transforms_amd._lib import stream_ptr
        Xb = torch.empty((B, T_FRAMES, F_BINS), dtype=torch.complex64, device=dev)
        fb = torch.empty((B, T_FRAMES, N_MELS), dtype=torch.float32, device=dev)
        yb = torch.empty((B * T_FRAMES * HOP,), dtype=torch.float32, device=dev)
        xb = torch.zeros((B * T_FRAMES * HOP + 1024,), dtype=torch.float32, device=dev)
        res = {}
        for G in (8, 173):
            f_ms = timed_ms(lambda: h.pat_fwd(xb.data_ptr(), Xb.data_ptr(), fb.data_ptr(), frames_per_step, G, 4, stream_ptr()), 30, 25)
            i_ms = timed_ms(lambda: h.pat_inv(Xb.data_ptr(), yb.data_ptr(), frames_per_step, G, 4, stream_ptr()), 30, 25)
            res["G%d" % G] = {"fwd_ms": round(f_ms, 4), "inv_ms": round(i_ms, 4)}
        res["note"] = ("pattern-only copy kernels, dispatch order, 4 waves per workgroup, G frames per wave; G = 8 is the "
                       "compact write front (best case), G = 173 the product kernels' run length at this batch")
        return res

    def extra_power():
        """Socket power and shader clock while the step runs (1 s of steps after the timed region; the sampler is a side
        thread reading two sysfs files every 10 ms)."""
        hw = device_hwmon(local_rank)
        if hw is None:
            return {"available": False}
        with PowerSampler(hw) as ps:
            out = {"available": True, "cap_watts": ps.cap_watts()}
            X0 = stft(x)
            legs = [("step", step)]
            if fused:
                legs += [("fused_fwd", lambda: mag.forward_fused(stft, x, return_spectrum=True)), ("istft", lambda: stft.invert(X0))]
            for name, fn in legs:
                for _ in range(30):
                    fn()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                n = 0
                while time.perf_counter() - t1 < 1.0:
                    for _ in range(20):
                        fn()
                    n += 20
                    torch.cuda.synchronize()
                t2 = time.perf_counter()
                w = ps.window(t1 + 0.25, t2)
                if w:
                    ms = (t2 - t1) / n * 1e3
                    w.update({"ms": round(ms, 4), "microjoules_per_frame": round(w["watts"] * ms * 1e-3 / frames_per_step * 1e6, 4)})
                    out[name] = {k: (round(v, 1) if isinstance(v, float) and k != "microjoules_per_frame" and k != "ms" else v)
                                 for k, v in w.items()}
            last.clear()
            del X0
        return out

    def extra_h2d_inclusive():
        # the boundary handed host buffers: H2D of the audio + the step + D2H of features and audio (pinned memory)
        xh = torch.empty(B, CLIP_LEN, pin_memory=True)
        xh.copy_(x)
        fh = torch.empty(B, T_FRAMES, N_MELS, pin_memory=True)
        yh = torch.empty(B, HOP * (T_FRAMES - 1), pin_memory=True)

        def one():
            x.copy_(xh, non_blocking=True)
            step()
            fh.copy_(last["feat"], non_blocking=True)
            yh.copy_(last["y"], non_blocking=True)
            torch.cuda.synchronize()

        one()
        t1 = time.perf_counter()
        n = 3
        for _ in range(n):
            one()
        dt = (time.perf_counter() - t1) / n
        return {"frames_per_s": frames_per_step / dt, "ms_per_step": dt * 1e3,
                "bytes_h2d": xh.numel() * 4, "bytes_d2h": (fh.numel() + yh.numel()) * 4,
                "note": "PCIe-inclusive: pinned host audio -> device, step, features + audio -> pinned host (the spectrum "
                        "stays on the device); never `value`"}

    def extra_pghi(collect):
        # BASELINE configs[2]: DGT + PGHI invert round trip.  Dense noise is the worst case (every bin above the
        # tolerance); the tonal set (8 decaying sinusoids per clip) is SURVEY 8d's second input.
        dgt = A.DGT(sr=SR, n_fft=N_FFT, hop_length=HOP).to(dev)
        pg = {}
        sizes = sorted({min(args.pghi_clips, B), min(4 * args.pghi_clips, 4096)})
        for tag in ("noise", "tonal", "decaying"):
            for nb in (sizes if tag == "noise" else sizes[:1]):
                if tag == "noise":
                    xs = x if nb <= B else torch.randn(nb, CLIP_LEN, device=dev, generator=gen) * 0.1
                elif tag == "tonal":
                    xs = synth_tonal(nb, CLIP_LEN, device=dev)
                else:   # SURVEY Appendix B's third workload: noise x exp(-8 t) -- hundreds of reseeds per clip
                    xs = x[:nb] * torch.exp(-8.0 * torch.arange(CLIP_LEN, device=dev, dtype=torch.float32) / SR)
                m = dgt(xs[:nb]).abs()
                if nb == sizes[0]:
                    collect[tag] = m[:256].cpu()              # the CPU baseline runs on the same magnitudes
                yp = dgt.invert(m, inversion_mode="pghi")      # warm-up: first touch of the workspace
                del yp
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                yp = dgt.invert(m, inversion_mode="pghi")
                torch.cuda.synchronize()
                dt = time.perf_counter() - t1
                thr = m.amax(dim=(1, 2), keepdim=True) * float(dgt.tolerance)
                pops = int((m >= thr).sum())
                pg["%s_clips_%d" % (tag, nb)] = {"frames_per_s": nb * T_FRAMES / dt, "seconds": dt,
                                                 "heap_pops_per_s": pops / dt, "bins_above_tolerance": pops / m.numel()}
                if nb == sizes[0]:
                    # outside the timing: what the timed call produced, clips {0, nb/2-1, nb-1}, against the exact-order
                    # C oracle -- pop order bit for bit (results do not depend on the batch a clip rides in, so the
                    # order is re-recorded on those three clips alone) and the audio by SNR
                    pg["%s_spot_check" % tag] = pghi_spot_check(dgt, m, yp, sorted({0, nb // 2 - 1 if nb > 1 else 0, nb - 1}))
                del m, yp, xs
        pg["input"] = ("noise: |DGT(randn*0.1)|, ~100% of bins above tolerance; tonal: 8 decaying sinusoids per clip; "
                       "decaying: noise x exp(-8 t), ~10% of bins, ~300 reseeds per clip; "
                       "DGT.invert(mag, 'pghi') = gradients + heap integration + polar ISTFT")
        return pg

    def pghi_spot_check(dgt, m, yp, ids):
        from oracle import oracle as O
        from acids_transforms_amd import ops as _ops
        sub = m[ids].contiguous()
        _, npops, order = _ops.pghi_offline(sub, float(dgt.gamma), N_FFT, HOP, float(dgt.tolerance), float(dgt.eps), debug=True)
        dual = dgt.inv_window[:N_FFT].cpu()
        worst_snr, order_ok, pops = float("inf"), True, 0
        for j, b in enumerate(ids):
            mb = sub[j].cpu()
            r = O.pghi_offline(mb, N_FFT, HOP, tol=float(dgt.tolerance), want_order=True)
            k = len(r["order"])
            pops += k
            got = order[j][:k].cpu().numpy()
            order_ok = order_ok and int(npops[j]) == k and bool((got == r["order"][:, 0] * F_BINS + r["order"][:, 1]).all())
            ref = O.polar_istft(mb.unsqueeze(0), torch.from_numpy(r["phase"]).unsqueeze(0), dual, N_FFT, HOP)[0].double()
            err = yp[b].cpu().double() - ref
            worst_snr = min(worst_snr, 10.0 * float(torch.log10(ref.pow(2).sum() / err.pow(2).sum().clamp_min(1e-300))))
        return {"clips": ids, "pops_checked": pops, "pop_order_identical": order_ok, "min_audio_snr_db": round(worst_snr, 1),
                "ok": bool(order_ok and worst_snr > 40.0),
                "note": "vs oracle/pghi_ref.c: heap pop order bit for bit; audio of the TIMED call vs istft(mag e^{i phase_oracle})"}

    def extra_northstar(collect):
        # north_star's composite: the reference README's chain (README.md:48-61) Mono() + DGT(pghi) + Magnitude(mel,
        # unipolar, log1p), scale_data once, then forward AND invert on B clips x 4 s of stereo audio, one timed call each
        chain = (A.Mono() + A.DGT(sr=SR, n_fft=N_FFT, hop_length=HOP, inversion_mode="pghi")
                 + A.Magnitude(sr=SR, mel=True, mode="unipolar", contrast="log1p")).to(dev)
        res = {"chain": repr(chain), "clips": B}
        for tag in ("noise", "tonal"):
            if tag == "noise":
                xs = torch.stack([x, x.flip(0)], 1)                                  # (B, 2, L) stereo
            else:
                tn = synth_tonal(B, CLIP_LEN, device=dev)
                xs = torch.stack([tn, 0.5 * tn.roll(1, 0)], 1)
            chain.scale_data(xs[:64])
            y = chain(xs)
            back = chain.invert(y)                                                   # warm-up: workspaces, caches
            del back
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            y = chain(xs)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            back = chain.invert(y)
            torch.cuda.synchronize()
            t3 = time.perf_counter()
            res[tag] = {"frames_per_s": B * T_FRAMES / (t3 - t1), "forward_s": t2 - t1, "invert_s": t3 - t2,
                        "out_shape": list(back.shape), "finite": bool(torch.isfinite(back).all())}
            collect[tag] = xs[:64].cpu()
            del xs, y, back
        res["reference_extrapolated_frames_per_s"] = dict(REFERENCE_COMPOSITE)
        res["vs_reference_extrapolated"] = {t: res[t]["frames_per_s"] / REFERENCE_COMPOSITE[t] for t in ("noise", "tonal")}
        res["target"] = ">= 100x the reference CPU STFT+mel+PGHI-invert throughput (north_star)"
        res["note"] = ("forward = mix-down + fused DGT/mel kernel; invert = banded inverse bank + PGHI (gradients, heap "
                       "integration) + polar ISTFT + channel axis; reference figure: BASELINE.md section 2, extrapolated")
        return res

    def extra_stream():
        # BASELINE configs[4]: 256 streams, RealtimeDGT fwd + bf16-MFMA mel + RTPGHI + inverse + overlap-add, one
        # hipGraph replay per step; chunk sizes 256 (hop-sized step), 1024, 4096 samples
        from acids_transforms_amd.streaming import StreamingDGTSession
        S = args.streams
        rtres = {"streams": S, "steps_per_cell": args.stream_steps, "cells": {}}
        for C in (256, 1024, 4096):
            # 16 different chunks in turn: the SAME hop-sized chunk fed again and again makes every analysis frame of a
            # stream equal to the one before it -- a degenerate input (all magnitudes of the flood tie)
            chunks = [torch.randn(S, C, device=dev, generator=gen) * 0.1 for _ in range(16)]
            cell = {"realtime_budget_ms": C / SR * 1e3}
            for tag, use_graph in (("eager", False), ("hipgraph", True)):
                nst = args.stream_steps if use_graph else max(50, args.stream_steps // 10)
                try:
                    sess = StreamingDGTSession(S, C, N_FFT, HOP, SR, device=dev, use_graph=use_graph, mel_bands=N_MELS,
                                               mel_dtype="bf16")
                except Exception as exc:
                    cell["error_" + tag] = repr(exc)[:200]
                    continue
                for i in range(5):
                    sess.step(chunks[i % 16])
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for i in range(nst):
                    sess.step(chunks[i % 16])
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t1) / nst
                cell["ms_per_step_" + tag] = dt * 1e3
                cell["steps_per_s_" + tag] = 1.0 / dt
                cell["frames_per_s_" + tag] = S * (C // HOP) / dt
                del sess
            if C == 256:
                # the degenerate input kept visible: the SAME hop-sized chunk every step makes row f-1 == row f --
                # every source ties with its own bin (held frames, hop-periodic tones do this).  Round 4's rank bitmap fell
                # back to the heap there (1.03 ms); the scan path takes it: a source only ever visits its own bin, and with
                # no source larger than the frame maximum the unmarked seed is provably the flood's first pop
                try:
                    sess = StreamingDGTSession(S, C, N_FFT, HOP, SR, device=dev, use_graph=True, mel_bands=N_MELS, mel_dtype="bf16")
                    for i in range(5):
                        sess.step(chunks[0])
                    torch.cuda.synchronize()
                    nst = max(50, args.stream_steps // 4)
                    t1 = time.perf_counter()
                    for i in range(nst):
                        sess.step(chunks[0])
                    torch.cuda.synchronize()
                    cell["ms_per_step_hipgraph_repeated_chunk"] = (time.perf_counter() - t1) / nst * 1e3
                    del sess
                except Exception as exc:
                    cell["error_repeated_chunk"] = repr(exc)[:200]
            rtres["cells"]["chunk_%d" % C] = cell
        rtres["mel"] = "bf16 MFMA projection (v_mfma_f32_32x32x16_bf16, fp32 accumulate), %d log1p mel features per frame" % N_MELS
        return rtres

    def extra_phase_repr():
        # the stft+polar chain's phase side at the same size; HBM-bound scans along time
        from acids_transforms_amd import ops as _ops
        Xs = stft(x)
        res = {}

        def entry(ms, bytes_per_bin):
            a = frames_per_step * F_BINS * bytes_per_bin / (ms * 1e-3) / 1e9
            return {"ms": round(ms, 4), "achieved_GBps": round(a, 1), "frac_of_8TBps": round(a / HBM_PEAK_GBS, 4),
                    "algorithmic_bytes_per_bin": bytes_per_bin}

        res["phase_angle"] = entry(timed_ms(lambda: _ops.phase_scan(Xs, "angle"), 5, 1), 12)
        res["if_forward"] = entry(timed_ms(lambda: _ops.phase_scan(Xs, "forward"), 5, 1), 12)
        inst = _ops.phase_scan(Xs, "forward")
        res["if_invert_forward"] = entry(timed_ms(lambda: _ops.phase_integrate(inst, "forward"), 5, 1), 8)
        res["if_invert_central"] = entry(timed_ms(lambda: _ops.phase_integrate(inst, "central"), 5, 1), 8)
        mg_ = Xs.abs()
        res["polar_to_complex"] = entry(timed_ms(lambda: _ops.polar_to_complex(mg_, inst), 5, 1), 16)
        del inst, mg_
        # the stacked representations (one pass over the spectrum each way; 8 B in, 8 B out per bin)
        for name, tr in (("polar", A.Polar()), ("polar_if", A.PolarIF()), ("cartesian", A.Cartesian())):
            tr = tr.to(dev)
            tr.scale_data(Xs)
            yy = tr(Xs)
            res[name + "_forward"] = entry(timed_ms(lambda: tr(Xs), 3, 1), 16)
            res[name + "_invert"] = entry(timed_ms(lambda: tr.invert(yy), 3, 1), 16)
            del yy
        return res

    def extra_hbm_probe():
        # what this box's HBM gives a flat streaming kernel today (a measured ceiling next to the spec)
        n = 1 << 29                                   # 2 GiB of fp32
        a = torch.empty(n, device=dev)
        b = torch.empty(n, device=dev)
        t_fill = timed_ms(lambda: a.fill_(1.0), 6, 2) * 1e-3
        t_copy = timed_ms(lambda: b.copy_(a), 6, 2) * 1e-3
        t_read = timed_ms(lambda: a.sum(), 6, 2) * 1e-3
        return {"copy_GBps": round(2 * 4 * n / t_copy / 1e9, 1), "fill_GBps": round(4 * n / t_fill / 1e9, 1),
                "read_sum_GBps": round(4 * n / t_read / 1e9, 1),
                "note": "torch fill_/copy_/sum on 2 GiB fp32 buffers: the practical ceiling for the fractions above "
                        "(roofline.peak stays the 8 TB/s spec)"}

    def extra_other_hops():
        res = {}
        for hop in (128, 512):
            st = A.STFT(sr=SR, n_fft=N_FFT, hop_length=hop).to(dev)
            Xh = st(x)
            f_ms = timed_ms(lambda: st(x), 12, 12)
            i_ms = timed_ms(lambda: st.invert(Xh), 12, 12)
            frames = B * Xh.shape[-2]
            res["hop_%d" % hop] = {"frames_per_clip": int(Xh.shape[-2]), "forward_ms": round(f_ms, 4),
                                   "inverse_ms": round(i_ms, 4), "frames_per_s_fwd_plus_inv": frames / ((f_ms + i_ms) * 1e-3)}
            del Xh
        return res

    def extra_other_sizes():
        # n_fft 512 / 2048 / 4096 (hop n_fft / 4) on the register FFT core: two frames per wave FFT / two / four FFTs per
        # frame; n_fft 400 (hop 160) on the mixed-radix kernels
        res = {}
        for n in (512, 2048, 4096):
            st = A.STFT(sr=SR, n_fft=n, hop_length=n // 4).to(dev)
            Xh = st(x)
            f_ms = timed_ms(lambda: st(x), 12, 12)
            i_ms = timed_ms(lambda: st.invert(Xh), 12, 12)
            frames = B * Xh.shape[-2]
            bytes_per_frame = (n // 4) * 4 + (n // 2 + 1) * 8
            res["n_fft_%d" % n] = {"frames_per_clip": int(Xh.shape[-2]), "forward_ms": round(f_ms, 4), "inverse_ms": round(i_ms, 4),
                                   "forward_frac_of_8TBps": round(frames * bytes_per_frame / (f_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                   "inverse_frac_of_8TBps": round(frames * bytes_per_frame / (i_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
            del Xh
        # MelSpectrogram at torchaudio's / librosa's usual 2048 / 512 / 128 mels: one kernel, the spectrum is never written
        ms_ = A.MFCC(sr=SR, n_fft=2048, hop_length=512, n_mels=128).to(dev)
        res["melspectrogram_2048_512_128"] = {"ms": round(timed_ms(lambda: ms_(x), 12, 12), 4)}
        st = A.STFT(sr=SR, n_fft=400, hop_length=160).to(dev)
        Xh = st(x)
        res["n_fft_400_hop_160"] = {"frames_per_clip": int(Xh.shape[-2]), "forward_ms": round(timed_ms(lambda: st(x), 12, 12), 4),
                                    "inverse_ms": round(timed_ms(lambda: st.invert(Xh), 12, 12), 4)}
        del Xh
        return res

    def extra_griffin_lim():
        m = stft(x).abs()
        stft.invert(m)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        stft.invert(m)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        return {"seconds": dt, "frames_per_s": frames_per_step / dt, "iterations": 30}

    # Order: every leg that holds collectives runs on ALL ranks first (config4 is the only one;
    # `guarded` records its LegFailed identically on every rank); the rank-0-only legs follow and hold none, so no rank
    # ever waits in a collective for one that is busy elsewhere.  They are single-GPU figures and run at N = 1 only:
    # at N > 1 the other ranks would sit idle for a minute, and the N = 1 line of the same sweep carries them.
    pghi_inputs = {}
    northstar_inputs = {}
    if not args.no_extras:
        guarded("config4", lambda: config4_figures(max(10, args.steps // 5), 3))
    if rank == 0:
        guarded("parity_spot_check", parity_spot_check)          # CPU oracle on three clips of the last timed step
    if rank == 0 and world == 1:
        guarded("pattern_copy", extra_pattern_copy)               # ~1 s: kept under --no-extras too
        guarded("power", extra_power)
    if not args.no_extras:
        if rank == 0 and world == 1:
            guarded("hbm_probe", extra_hbm_probe)
            guarded("h2d_inclusive", extra_h2d_inclusive)
            guarded("other_hops", extra_other_hops)
            guarded("other_sizes", extra_other_sizes)
            guarded("griffin_lim_invert", extra_griffin_lim)
            guarded("phase_representations", extra_phase_repr)
            if args.pghi_clips > 0:
                guarded("pghi_invert", lambda: extra_pghi(pghi_inputs))
                guarded("northstar_pipeline", lambda: extra_northstar(northstar_inputs))
            if args.streams > 0:
                guarded("realtime_dgt_stream", extra_stream)

    result = {
        "metric": "spectrogram frames/sec (fwd+invert), n_fft=1024 hop=256",
        "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "configs[1]: batch=%d clips/GPU x 4 s mono 44.1 kHz, STFT fwd + Magnitude(mel=128, log1p, "
                               "unipolar)%s + ISTFT invert, fp32" % (B, " [one fused kernel]" if fused else ""),
                   "n_fft": N_FFT, "hop": HOP, "frames_per_clip": T_FRAMES, "clips_per_gpu": B,
                   "sharding": "clips, no data-path collective"},
        "world_size_observed": dist.get_world_size() if use_dist else 1, "backend": backend,
        "hooks_armed": hooks_armed(),
        "roofline": roof,
        "kernels": kernels,
        "settle_steps": settled,
        "fresh_process_ms_per_step": round(fresh_ms if fresh_ms is not None else ms_per_step, 4),
        "fresh_process_note": ("steps W+1 .. W+K of this process's life, before any settling: what `--settle-steps 0` "
                               "times (HIP events around the K steps); `ms_per_step` is the settled figure"),
        "timed_step_ms": {"first": step_ms_events[0], "last": step_ms_events[-1], "min": min(step_ms_events),
                          "max": max(step_ms_events), "all": step_ms_events,
                          "note": "per-step kernel time (HIP events) inside the timed region: flat = steady state"},
    }
    result.update(extras)
    if "pattern_copy" in extras:
        pc = extras["pattern_copy"]
        roof["pattern_copy_fwd_ms"], roof["pattern_copy_inv_ms"] = pc["G8"]["fwd_ms"], pc["G8"]["inv_ms"]
        roof["pattern_copy_fwd_long_runs_ms"], roof["pattern_copy_inv_long_runs_ms"] = pc["G173"]["fwd_ms"], pc["G173"]["inv_ms"]
        roof["achieved_vs_pattern_copy"] = round((pc["G8"]["fwd_ms"] if dominant == "stft_fwd" else pc["G8"]["inv_ms"]) / roof["ms"], 4)
        if fused:
            roof["step_vs_pattern_copy"] = round((pc["G8"]["fwd_ms"] + pc["G8"]["inv_ms"]) / ms_per_step, 4)
    if extras.get("power", {}).get("available"):
        pw = extras["power"]
        roof["power_cap_watts"] = pw.get("cap_watts")
        for leg in ("step", "fused_fwd", "istft"):
            if leg in pw:
                roof[leg + "_socket_watts"] = pw[leg]["watts"]
                roof[leg + "_sclk_mhz"] = pw[leg]["sclk_mhz"]
                roof[leg + "_microjoules_per_frame"] = pw[leg]["microjoules_per_frame"]
    if "hbm_probe" in extras:
        # the pool's boxes differ by up to 9 % on the step kernels and that spread follows what each box gives a MIXED
        # read / write stream (torch copy_: 4.77 ... 5.07 TB/s), not its write-only or read-only rate (fill_ / sum: equal
        # on all of them): the achieved rate next to the same box's copy_, for comparing runs across boxes
        roof["achieved_vs_box_copy"] = round(roof["achieved"] / extras["hbm_probe"]["copy_GBps"], 4)
        roof["box_copy_GBps"] = extras["hbm_probe"]["copy_GBps"]
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        hc = host_cpus()
        result["host_cpus"] = hc
        # torch's CPU ops do not scale to every hardware thread of the box (oversubscription beyond the job's CPU
        # share): time a few thread counts briefly, report the best as THE baseline and keep the sweep
        sweep = {}
        for th in sorted({1, 8, 16, 32, 64, hc["usable"], hc["affinity"]}):
            if th <= hc["affinity"]:
                sweep[th] = cpu_baseline(th, 8 if th == 1 else 96, budget_s=3.0)
        best = max(sweep, key=lambda t: sweep[t]["value"])
        for v in sweep.values():
            # `cores` (the contract's field) = threads used; next to it what those threads ran on
            v.update({"threads": v["cores"], "cpu_share": hc["cgroup_quota"] if hc["cgroup_quota"] is not None else hc["affinity"],
                      "host_threads": hc["os_cpu_count"]})
        result["cpu_baseline"] = sweep[best]
        result["cpu_baseline_1thread"] = sweep[1]
        result["cpu_baseline_sweep"] = {str(t): round(v["value"], 1) for t, v in sweep.items()}
        if northstar_inputs and "northstar_pipeline" in result:
            th = min(hc["affinity"], 2 * hc["usable"])
            cpu_ns = cpu_baseline_northstar(northstar_inputs, th)
            result["cpu_baseline_northstar"] = cpu_ns
            result["northstar_pipeline"]["vs_host_port_all_cores"] = {
                t: result["northstar_pipeline"][t]["frames_per_s"] / cpu_ns[t]["value"] for t in cpu_ns}
        if pghi_inputs:
            # one clip per thread: oversubscribed threads only queue, so every hardware thread we may run on is used
            result["cpu_baseline_pghi"] = cpu_baseline_pghi(pghi_inputs, [hc["usable"], 2 * hc["usable"], hc["affinity"]])
            try:
                gpu = result["pghi_invert"]
                cpu = result["cpu_baseline_pghi"]
                result["pghi_gpu_vs_all_cores"] = {
                    t: gpu["%s_clips_%d" % (t, min(args.pghi_clips, B))]["frames_per_s"] / cpu[t]["all_cores"]["value"]
                    for t in cpu}
            except Exception:
                pass
    failed = sorted(k for k in result if k.endswith("_error"))
    unverified = verification_failures(result)
    result["verification"] = {"spot_checks_failed": unverified, "ok": not unverified}
    status = exit_status(result)
    if rank == 0:
        print(json.dumps(result))
        sys.stdout.flush()
    if use_dist:
        # collective legs fail on all ranks together; the spot checks run on rank 0 only, so only rank 0 can leave with
        # status 5 -- the launcher (launch_ranks / torchrun) relays any non-zero rank
        dist.destroy_process_group()
    if unverified:
        sys.stderr.write("bench.py: spot check(s) of timed results FAILED: %s\n" % ", ".join(unverified))
    if failed:
        # the line above is complete and valid, but a side measurement raised: say so with the exit status
        # (the driver's record lists key names only; a silent `*_error` string would pass unnoticed)
        sys.stderr.write("bench.py: side measurement(s) failed: %s\n" % ", ".join(failed))
    if status:
        sys.exit(status)


def pmc_traffic(key):
    """HBM bytes per launch of the dominant kernel from this round's own rocprofv3 --pmc passes
    (tools/profile_gpu.sh writes profiles/rNN_pmc_traffic.json; the newest one is read)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_traffic.json")))
    for f in reversed(files):
        try:
            v = json.load(open(f)).get(key)
            if v is not None:
                return v, os.path.relpath(f, ROOT)
        except Exception:
            continue
    return None, None


if __name__ == "__main__":
    main()
