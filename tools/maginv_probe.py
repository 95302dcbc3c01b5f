"""Magnitude.invert at n_fft 1024 (128 and 513 mel filters -> 513 bins), 1024 clips x 690 frames (ACIDS_BANDED_NO_DEFER=1: stores per pass)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import acids_transforms_amd as A
dev = torch.device("cuda:0")
B, T, F = 1024, 690, 513
X = torch.view_as_complex(torch.randn(B, T, F, 2, device=dev))


def timeit(fn, n=20, warm=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n


for nm in (128, None):
    for mode in (None, "unipolar"):
        mg = A.Magnitude(n_mels=nm, mode=mode).to(dev)
        if mode: mg.scale_data(X[:4])
        y = mg(X)
        t = timeit(lambda: mg.invert(y))
        n_in = y.shape[-1]
        print("Magnitude(n_mels=%s, mode=%s).invert  %.3f ms  (%.2f TB/s on %d B/frame)" % (nm, mode, t, B * T * (4 * n_in + 4 * F) / t / 1e9, 4 * n_in + 4 * F), flush=True)
        del y
