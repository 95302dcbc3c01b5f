"""How much does one more kernel node cost in a hipGraph replay (and in eager launches) on this stack?  k tiny kernels
(at_affine on 256 x 513 floats, the size of the streaming step's per-frame tensors) per graph, k = 1 .. 12."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from acids_transforms_amd import ops

dev = torch.device("cuda:0")
x = torch.randn(256, 513, device=dev)
off, sc = torch.zeros((), device=dev), torch.ones((), device=dev)


def body(k):
    src = x
    for i in range(k):
        src = ops.affine(src, off, sc)


for k in (1, 2, 4, 6, 8, 10, 12):
    for _ in range(3):
        body(k)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body(k)
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(500):
        g.replay()
    torch.cuda.synchronize()
    tg = (time.perf_counter() - t0) / 500 * 1e6
    t0 = time.perf_counter()
    for _ in range(200):
        body(k)
    torch.cuda.synchronize()
    te = (time.perf_counter() - t0) / 200 * 1e6
    print("k = %2d kernels: graph replay %.1f us, eager %.1f us" % (k, tg, te), flush=True)
