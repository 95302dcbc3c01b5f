"""Dev: check the rank records of the realtime pre-pass per (stream, frame): sort order, inverse table, the number of
"same magnitude as the rank before" bits."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes
import numpy as np
import torch
import acids_transforms_amd as A
from acids_transforms_amd import ops
from acids_transforms_amd._lib import lib, ptr, check, stream_ptr
dev = torch.device("cuda:0")
S, F, n = 4, 513, 4
rt = A.RealtimeDGT(n_fft=1024, hop_length=256, batch_size=[S]).to(dev)
g = torch.Generator().manual_seed(1)
m = (torch.randn(S, n + 2, F, generator=g) ** 2 + torch.randn(S, n + 2, F, generator=g) ** 2).sqrt()
hist, mag = m[:, :2].contiguous().to(dev), m[:, 2:].contiguous().to(dev)
prev = (torch.rand(S, F, generator=g) * 6.28).to(dev)
noise = torch.randn(S, n, F, generator=g).to(dev)
wsb = lib().at_pghi_rt_workspace_bytes(S, n, F)
ws = torch.zeros((wsb + 7) // 8, dtype=torch.int64, device=dev)
phase = torch.empty_like(mag)
check(lib().at_pghi_realtime(ptr(hist), ptr(mag), ptr(prev), ptr(noise), S, n, F, float(rt.gamma), 1024, 256, float(rt.tolerance),
                             float(rt.eps), ptr(phase), None, None, ptr(ws), wsb, stream_ptr()), "rt")
torch.cuda.synchronize()
raw = ws.cpu().numpy().view(np.uint8)
per = (n + 2) * F
base = ws.data_ptr()
heap_off = ((base + 6 * S * per * 4 + 15) & ~15) - base
rank_off = ((base + heap_off + S * (4 * F + 8) * 8 + 15) & ~15) - base
stride_b = ((8 * F + 4 * ((2 * F + 31) // 32) + 15) & ~15)
smax = float(m.max())
print("abstol", 1e-2 * smax, "tolerance", float(rt.tolerance))
for s in range(S):
    for fr in range(n):
        rec = raw[rank_off + (s * n + fr) * stride_b:][:stride_b]
        eor = rec[:4 * F].view(np.uint16)[:2 * F]
        roe = rec[4 * F:8 * F].view(np.uint16)
        sp = rec[8 * F:8 * F + 4 * ((2 * F + 31) // 32)].view(np.uint32)
        tie = int(sum(bin(int(w)).count("1") for w in sp))
        rows = m[s, fr + 1:fr + 3].reshape(-1).numpy()
        order_ok = np.array_equal(np.argsort(-rows, kind="stable"), eor.astype(np.int64))
        print(s, fr, "tied ranks", tie, "sorted ok", order_ok, "roe ok", np.array_equal(roe[eor], np.arange(2 * F)))
