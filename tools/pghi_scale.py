"""Dev measurement: offline PGHI (exact heap order) throughput against the number of clips resident on one GPU.
The integration is serial per clip (one wavefront each), so clips are the only parallelism: the time of a launch
stays nearly flat until every SIMD holds several waves."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import acids_transforms_amd as A

dev = torch.device("cuda:0")
L, T, F = 176400, 690, 513
d = A.DGT().to(dev)
sizes = [int(s) for s in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["1024", "4096", "8192"])]
gen = torch.Generator(device=dev).manual_seed(3)


def mags(B):
    out = torch.empty(B, T, F, device=dev)
    for i in range(0, B, 1024):
        n = min(1024, B - i)
        x = torch.randn(n, L, device=dev, generator=gen) * 0.1
        out[i:i + n] = d(x).abs()
        del x
    return out


m = mags(256)
d.pghi(m, d.tolerance)
torch.cuda.synchronize()
del m
for B in sizes:
    m = mags(B)
    d.pghi(m, d.tolerance)                      # first touch of this size's workspace (tens of GB) is not part of the figure
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ph = d.pghi(m, d.tolerance)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("clips %6d  %7.3f s  %8.1f kframes/s  %7.1f Mpops/s  (%.1f GB allocated)" % (
        B, dt, B * T / dt / 1e3, B * T * F / dt / 1e6, torch.cuda.max_memory_allocated() / 1e9), flush=True)
    del m, ph
    torch.cuda.empty_cache()
