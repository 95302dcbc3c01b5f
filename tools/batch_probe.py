"""Dev timing: the headline chain at small batches (latency side): fused STFT + mel128 forward, ISTFT, offline PGHI."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import acids_transforms_amd as A

dev = torch.device("cuda:0")
stft = A.STFT().to(dev)
dgt = A.DGT().to(dev)
mag = A.Magnitude(n_mels=128).to(dev)
xs = torch.randn(1024, 176400, device=dev) * 0.1
mag.scale_data(stft(xs[:8]))


def timeit(fn, n=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for B in (1, 2, 8, 32, 128, 512, 1024):
    x = xs[:B]
    X = stft(x)
    f = timeit(lambda: mag.forward_fused(stft, x, return_spectrum=True))
    i = timeit(lambda: stft.invert(X))
    m = dgt(x).abs()
    p = timeit(lambda: dgt.pghi(m, dgt.tolerance), n=2, warm=1) if B <= 128 else float("nan")
    print("B %5d: fused forward %8.1f us  (%6.2f us/clip)   inverse %8.1f us   pghi %10.1f us" % (B, f, f / B, i, p), flush=True)
