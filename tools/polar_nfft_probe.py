"""Polar forward / invert at other FFT sizes (rows of 257 / 513 / 1025 / 2049 bins), 1024 clips x 4 s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import acids_transforms_amd as A

dev = torch.device("cuda:0")


def timeit(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


for n_fft in (512, 1024, 2048, 4096):
    F, T = n_fft // 2 + 1, 176400 // (n_fft // 4) + 1
    X = torch.randn(1024, T, F, dtype=torch.complex64, device=dev)
    pol = A.Polar(magnitude_args={"n_fft": n_fft}).to(dev)
    pol.scale_data(X[:4])
    y = pol(X)
    gb = 16.0 * X.numel() / 1e9
    tf, ti = timeit(lambda: pol(X)), timeit(lambda: pol.invert(y))
    print("n_fft %4d  (1024 x %d x %d)  forward %.3f ms (%.2f TB/s)   invert %.3f ms (%.2f TB/s)"
          % (n_fft, T, F, tf, gb / tf, ti, gb / ti), flush=True)
    del X, y
