"""Dev experiment: does the 256 MB infinity cache serve the inverse's reads when forward and inverse run chunk by chunk?
fused forward (spectrum + mel) then ISTFT over `c` clips at a time; the spectrum of a chunk is c * 690 * 513 * 8 bytes."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import acids_transforms_amd as A

dev = torch.device("cuda:0")
B = 1024
x = torch.randn(B, 176400, device=dev) * 0.1
stft = A.STFT().to(dev)
mag = A.Magnitude(n_mels=128).to(dev)
mag.scale_data(stft(x[:8]))


def run(c):
    outs = []
    for i in range(0, B, c):
        X, feat = mag.forward_fused(stft, x[i:i + c], return_spectrum=True)
        y = stft.invert(X)
        outs.append((feat, y))
    return outs


def timeit(fn, n=20, warm=10):
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


for c in (1024, 512, 256, 128, 64, 32, 16):
    eager = timeit(lambda: run(c))
    # the same sequence replayed as one hipGraph: no host time between the launches
    g = torch.cuda.CUDAGraph()
    run(c)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        keep = run(c)
    graphed = timeit(g.replay)
    print("chunk %4d clips (%6.1f MB of spectrum): eager %.3f ms, graph %.3f ms per 1024 clips"
          % (c, c * 690 * 513 * 8 / 1e6, eager, graphed), flush=True)
    del g, keep
