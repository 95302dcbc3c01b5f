"""Dev timing of the individual kernels at BASELINE config-2 size (not the bench contract)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import acids_transforms_amd as A

which = sys.argv[1].split(",") if len(sys.argv) > 1 else ["fwd", "inv", "mel128", "mel513"]
# PERF_VARIANTS="epilogue=1,frame_kernels=1": kernel variants through the C ABI (at_set_variant); the library reads no environment
for kv in filter(None, os.environ.get("PERF_VARIANTS", "").split(",")):
    from acids_transforms_amd import _lib as _L
    _L.check(_L.lib().at_set_variant(_L.VARIANTS[kv.split("=")[0]], int(kv.split("=")[1])), "at_set_variant")
B, L = 1024, 176400
dev = torch.device("cuda:0")
x = torch.randn(B, L, device=dev) * 0.1
m = A.STFT().to(dev)
T = 1 + L // 256
frames = B * T


N_ITER = int(os.environ.get("PERF_N", "10"))
N_WARM = int(os.environ.get("PERF_WARM", "3"))


def timeit(fn, n=None, warm=None):
    n = N_ITER if n is None else n          # a fresh process needs ~15 launches to reach steady clocks: PERF_WARM=20 PERF_N=40
    warm = N_WARM if warm is None else warm
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


def report(name, t, bpf, flops=0):
    extra = "  %6.1f TFLOP/s" % (frames * flops / t / 1e9) if flops else ""
    print("%-16s %8.3f ms  %8.1f Mframes/s  %6.2f TB/s algorithmic (%.0f%% of 8 TB/s)%s" % (
        name, t, frames / t / 1e3, frames * bpf / t / 1e9, frames * bpf / t / 1e9 / 8 * 100, extra), flush=True)


X = m(x)
if "fwd" in which:
    report("stft_fwd", timeit(lambda: m(x)), 5128)
if "inv" in which:
    report("istft", timeit(lambda: m.invert(X)), 5128)
if "mel128" in which:
    mg = A.Magnitude(n_mels=128, mode=None).to(dev)
    report("mel128", timeit(lambda: mg(X)), 4104 + 512, 2 * 513 * 128)
if "melbf16" in which:
    mgb = A.Magnitude(n_mels=128, mode=None, bank_dtype="bf16").to(dev)      # dense bf16 MFMA contraction (config 5)
    report("mel128 bf16 MFMA", timeit(lambda: mgb(X)), 4104 + 512, 2 * 513 * 128)
    mgb5 = A.Magnitude(mode=None, bank_dtype="bf16").to(dev)
    report("mel513 bf16 MFMA", timeit(lambda: mgb5(X), n=5), 4104 + 2052, 2 * 513 * 513)
if "mel513" in which:
    mg = A.Magnitude(mode=None).to(dev)
    report("mel513", timeit(lambda: mg(X), n=5), 4104 + 2052, 2 * 513 * 513)
if "fused" in which or "fused2" in which or "fusedfeat" in which:
    mgf = A.Magnitude(n_mels=128, mode="unipolar", contrast="log1p").to(dev)
    mgf.scale_data(X[:8])
    if "fused" in which:
        report("fwd+mel fused", timeit(lambda: mgf.forward_fused(m, x, return_spectrum=True)), 5640)
    if "fusedfeat" in which:     # the literal configs[1] forward: features only, row-major, spectrum never stored
        from acids_transforms_amd import ops
        off, sc = mgf._affine()
        report("fwd+mel feat-only", timeit(lambda: ops.stft_mel_forward(x, m.window[:1024], mgf._banded(), "log1p", off, sc,
                                                                         mgf._eps, want_spectrum=False)), 1536)
    if "fused2" in which:
        mf = A.MFCC().to(dev)   # features only: the spectrum never reaches HBM
        report("fwd+mel (no X)", timeit(lambda: mf(x)), 1024 + 512)
if "fused513" in which or "fused513feat" in which:
    mg5 = A.Magnitude().to(dev)            # reference default: 513-filter bank
    mg5.scale_data(X[:8])
    if "fused513" in which:
        report("fwd+mel513 fused", timeit(lambda: mg5.forward_fused(m, x, return_spectrum=True)), 1024 + 4104 + 2052)
    if "fused513feat" in which:
        from acids_transforms_amd import ops
        off5, sc5 = mg5._affine()
        report("fwd+mel513 feat-only", timeit(lambda: ops.stft_mel_forward(x, m.window[:1024], mg5._banded(), "log1p", off5, sc5,
                                                                            mg5._eps, want_spectrum=False)), 1024 + 2052)
if "fusedraw" in which:
    mgr = A.Magnitude(n_mels=128, mode=None, contrast=None).to(dev)
    report("fwd+mel raw", timeit(lambda: mgr.forward_fused(m, x, return_spectrum=True)), 5640)
if "phase" in which:
    from acids_transforms_amd import ops
    F = 513
    report("angle elementwise", timeit(lambda: ops.angle(X)), 12 * F)
    report("angle (Phase)", timeit(lambda: ops.phase_scan(X, "angle")), 12 * F)
    report("unwrap", timeit(lambda: ops.phase_scan(X, "unwrap")), 12 * F)
    for meth in ("forward", "backward", "central"):
        report("IF " + meth, timeit(lambda: ops.phase_scan(X, meth)), 12 * F)
    y = ops.phase_scan(X, "forward")
    for meth in ("forward", "backward", "central"):
        report("IF.invert " + meth, timeit(lambda: ops.phase_integrate(y, meth)), 8 * F)
    mg_ = X.abs()
    report("polar->complex", timeit(lambda: ops.polar_to_complex(mg_, y)), 16 * F)
    del y, mg_
if "mfcc40" in which:
    mf40 = A.MFCC(n_mfcc=40).to(dev)
    report("MFCC(n_mfcc=40)", timeit(lambda: mf40(x)), 1024 + 4 * 40)
if "dct40" in which:      # the DCT behind MFCC(n_mfcc=40): 128 log-mel values -> 40 coefficients, channel-major
    from acids_transforms_amd import ops
    lm = torch.randn(B, T, 128, device=dev)
    dct = torch.randn(128, 40, device=dev)
    report("DCT 128->40", timeit(lambda: ops.mel_forward_real(lm, dct, None, None, channel_major_T=T)), 4 * 128 + 4 * 40)
    got = ops.mel_forward_real(lm[:4], dct, None, None, channel_major_T=T)
    want = (lm[:4].double() @ dct.double()).transpose(-1, -2)
    print("    max |err| vs float64: %.3g (|out| max %.3g)" % ((got.double() - want).abs().max().item(), want.abs().max().item()))
    del lm
if "polarfwd" in which:
    from acids_transforms_amd import ops
    pol = A.Polar().to(dev)
    pol.scale_data(X[:8])
    report("Polar.forward (one pass)", timeit(lambda: pol(X), n=5), 8 * 513 + 8 * 513)
    Yp = pol(X)
    report("Polar.invert (one pass)", timeit(lambda: pol.invert(Yp), n=5), 8 * 513 + 8 * 513)
    report("Polar.invert (parts)", timeit(lambda: ops.polar_to_complex(pol.magnitude.invert(Yp[..., 0, :]), pol.phase.invert(Yp[..., 1, :])), n=5), 8 * 513 + 8 * 513)
    del Yp
    pol.stack = -3                      # generic path: Magnitude, Phase, torch.stack
    try:
        report("Polar.forward (parts+stack)", timeit(lambda: torch.stack([pol.magnitude(X), pol.phase(X)], -2), n=5), 8 * 513 + 8 * 513)
    finally:
        pol.stack = -2
if "stftpolar" in which:
    pol2 = A.Polar().to(dev)
    pol2.scale_data(X[:8])
    comp = m + pol2
    report("STFT+Polar (one kernel)", timeit(lambda: comp(x), n=5), 1024 + 8 * 513)
    report("STFT then Polar", timeit(lambda: pol2(m(x)), n=5), 1024 + 8 * 513)
if "sinebank" in which:
    mgs = X.abs()
    ph0 = 2 * torch.pi * torch.rand(513, 1)
    nb = int(os.environ.get("SINE_B", "1024"))
    t_ = timeit(lambda: m.get_sinebank_inversion(mgs[:nb], random_phase=ph0), n=3, warm=1)
    Lout = 256 * T + 1024
    print("sinebank offline B=%d  %.2f ms  %.1f Mframes/s  %.1f TFLOP/s (3 passes x 2 B F L)  %.1f Gsine-terms/s" % (
        nb, t_, nb * T / t_ / 1e3, 3 * 2 * nb * 513 * Lout / t_ / 1e9, nb * 513 * Lout / t_ / 1e6), flush=True)
    rt = A.RealtimeSTFT().to(dev)
    ch = torch.rand(256, 4, 513, device=dev)
    t_ = timeit(lambda: rt.get_sinebank_inversion(ch), n=10, warm=2)
    print("sinebank realtime 256 streams x 4 frames  %.3f ms per chunk  (%.1f Gsines/s)" % (t_, 256 * 4 * 1024 * 513 / t_ / 1e6))
    del mgs
if "polar" in which:
    mag, ph = X.abs(), X.angle()
    report("istft_polar", timeit(lambda: m._istft(mag=mag, phase=ph)), 5128)
if "pghi" in which:
    d = A.DGT().to(dev)
    nb = int(os.environ.get("PGHI_B", "256"))
    mg = d(x[:nb]).abs()
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    d.pghi(mg, d.tolerance)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("pghi B=%d  %.3f s  %.1f kframes/s  %.2f Mpops/s" % (nb, dt, nb * T / dt / 1e3, nb * T * 513 / dt / 1e6))
if "rt" in which:
    import time
    S, nfr = int(os.environ.get("RT_STREAMS", "256")), int(os.environ.get("RT_FRAMES", "16"))
    rt = A.RealtimeDGT(batch_size=[S]).to(dev)
    fr = torch.randn(S, nfr, 1024, device=dev) * 0.1
    Xr = rt(fr)
    mg = Xr.abs()
    rt.invert(mg, inversion_mode="pghi")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        rt.invert(mg, inversion_mode="pghi")
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print("rtpghi S=%d n=%d  %.2f ms per step  %.1f kframes/s  (real-time budget %.1f ms)" % (
        S, nfr, dt * 1e3, S * nfr / dt / 1e3, nfr * 256 / 44.1))
if "sizes" in which:
    # round-2 additions away from n_fft 1024: register core at 4096 / 2048 / 512, mixed radix, long-row banded walk,
    # one-kernel MelSpectrogram at 2048 / 512, stacked representations in place
    def plain(name, t):
        print("%-44s %8.3f ms" % (name, t), flush=True)
    del X
    for n_fft, hop in ((4096, 1024), (2048, 512), (512, 128), (400, 160)):
        st = A.STFT(n_fft=n_fft, hop_length=hop).to(dev)
        Xn = st(x)
        plain("STFT(%d, hop %d) forward" % (n_fft, hop), timeit(lambda: st(x), n=5, warm=2))
        plain("STFT(%d, hop %d) invert" % (n_fft, hop), timeit(lambda: st.invert(Xn), n=5, warm=2))
        if n_fft >= 2048:
            for nm in (128, None):
                mgn = A.Magnitude(n_fft=n_fft, n_mels=nm, mode=None).to(dev)
                yn = mgn(Xn)
                plain("Magnitude(n_fft %d, %s mels) forward" % (n_fft, nm or "n_fft/2+1"), timeit(lambda: mgn(Xn), n=5, warm=2))
                plain("Magnitude(n_fft %d, %s mels) invert" % (n_fft, nm or "n_fft/2+1"), timeit(lambda: mgn.invert(yn), n=5, warm=2))
                del yn
        if n_fft in (2048, 512):
            mf = A.MFCC(n_fft=n_fft, hop_length=hop, n_mels=128 if n_fft == 2048 else 64).to(dev)
            plain("MFCC(%d, hop %d) mel spectrogram, one kernel" % (n_fft, hop), timeit(lambda: mf(x), n=5, warm=2))
        if n_fft == 2048:
            for name, tr in (("Cartesian", A.Cartesian()), ("PolarIF", A.PolarIF(magnitude_args={"mode": "bipolar", "n_fft": 2048}))):
                tr = tr.to(dev)
                tr.scale_data(Xn)
                yy = tr(Xn)
                plain("%s @ n_fft 2048 forward" % name, timeit(lambda: tr(Xn), n=5, warm=2))
                plain("%s @ n_fft 2048 invert" % name, timeit(lambda: tr.invert(yy), n=5, warm=2))
                del yy
        del Xn
