"""Dev timing: host-side cost of one call (what a single-clip caller pays per op), with a cProfile of the hot spots."""
import cProfile
import os
import pstats
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import acids_transforms_amd as A

dev = torch.device("cuda:0")
x = torch.randn(1, 44100, device=dev) * 0.1
stft = A.STFT().to(dev)
mag = A.Magnitude(n_mels=128).to(dev)
X = stft(x)
mag.scale_data(X)


def bench(name, fn, n=2000):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%-34s host %.1f us per call (%.1f us with the final sync)" % (name, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6), flush=True)


bench("STFT.forward", lambda: stft(x))
bench("STFT.invert", lambda: stft.invert(X))
bench("Magnitude.forward", lambda: mag(X))
bench("STFT+Magnitude fused", lambda: mag.forward_fused(stft, x))
bench("torch.empty(1, 173, 513) alone", lambda: torch.empty((1, 173, 513), dtype=torch.complex64, device=dev))
pr = cProfile.Profile()
pr.enable()
for _ in range(2000):
    stft(x)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
