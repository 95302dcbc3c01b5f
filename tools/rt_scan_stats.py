"""Dev build only (ACIDS_HIP_LIB=tools/ab/libacids_dev.so): how often the realtime flood's scan path takes a frame, how
many resolution rounds it needs, why it declines; and ms per call of at_pghi_realtime by variant (0 scan -> rank -> heap,
4 rank -> heap, 3 heap) for 256 streams x n frames."""
import ctypes
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import acids_transforms_amd as A
from acids_transforms_amd import ops, _lib
from acids_transforms_amd._lib import variant

dev = torch.device("cuda:0")
L = _lib.lib()
has_stats = hasattr(L, "at_dev_rt_stats")
S, F = 256, 513
rt = A.RealtimeDGT(batch_size=[S]).to(dev)
g = torch.Generator(device=dev).manual_seed(3)
for kind in ("noise", "tonal"):
    for n in (1, 4, 16):
        if kind == "noise":
            m = (torch.randn(S, n + 2, F, device=dev, generator=g) ** 2 + torch.randn(S, n + 2, F, device=dev, generator=g) ** 2).sqrt()
        else:
            k = torch.arange(F, device=dev).float()
            c = torch.rand(S, 1, 6, device=dev, generator=g) * 400 + 20
            m = (torch.exp(-((k.view(1, 1, F, 1) - c.unsqueeze(2)) / 3.0) ** 2)).sum(-1).expand(S, n + 2, F).contiguous() + 1e-9
            m = m * (1 + 0.01 * torch.rand(S, n + 2, F, device=dev, generator=g))
        hist, mag = m[:, :2].contiguous(), m[:, 2:].contiguous()
        prev = torch.rand(S, F, device=dev, generator=g) * 6.28
        noise = torch.randn(S, n, F, device=dev, generator=g)
        args = (float(rt.gamma), 1024, 256, float(rt.tolerance), float(rt.eps))
        line = "%-6s n=%2d " % (kind, n)
        ref = None
        for kern in (0, 4, 3):
            with variant("pghi_kernel", kern):
                for _ in range(3):
                    out = ops.pghi_realtime(hist, mag, prev, noise, *args)
                torch.cuda.synchronize()
                if has_stats and kern == 0:
                    buf = (ctypes.c_uint * 8)()
                    L.at_dev_rt_stats(buf, 1)
                t0 = time.perf_counter()
                for _ in range(20):
                    out = ops.pghi_realtime(hist, mag, prev, noise, *args)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / 20 * 1e3
                if has_stats and kern == 0:
                    L.at_dev_rt_stats(buf, 1)
                    fr = max(1, buf[0])
                    line += "[frames %d scan ok %.2f %% rounds/frame %.2f declined: reseed %d tie %d; us per frame: scan %.1f, all %.1f] " % (
                        buf[0], 100.0 * buf[1] / fr, buf[2] / max(1, buf[1]), buf[3], buf[4], buf[5] / fr / 100.0, buf[6] / fr / 100.0)
            ref = out if ref is None else ref
            assert torch.equal(out, ref), (kind, n, kern)
            line += "v%d %.3f ms  " % (kern, dt)
        print(line, flush=True)
