"""Dev probe: offline PGHI on the bench's tonal set (8 decaying sinusoids per clip): time, pops and reseeds per clip."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import acids_transforms_amd as A
from bench import synth_tonal, CLIP_LEN

dev = torch.device("cuda:0")
B = int(os.environ.get("PGHI_B", "1024"))
d = A.DGT().to(dev)
for tag, x in (("tonal", synth_tonal(B, CLIP_LEN, device=dev)),
               ("decaying noise", torch.randn(B, CLIP_LEN, device=dev) * torch.exp(-8.0 * torch.arange(CLIP_LEN, device=dev) / 44100.0))):
    m = d(x).abs()
    d.pghi(m, d.tolerance)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ph = d.pghi(m, d.tolerance)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    thr = m.amax(dim=(1, 2), keepdim=True) * float(d.tolerance)
    live = (m >= thr)
    pops = float(live.sum()) / B
    # seeds = cells whose phase stayed exactly 0 while above the threshold (every flood starts from one)
    seeds = float(((ph == 0) & live).sum()) / B
    print("%-16s %4d clips  %.4f s  %.3e frames/s  pops/clip %.0f (%.1f %% of bins)  seeds/clip %.1f"
          % (tag, B, dt, B * 690 / dt, pops, 100 * pops / (690 * 513), seeds), flush=True)
