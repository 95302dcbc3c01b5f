import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import acids_transforms_amd as A
from acids_transforms_amd import ops
from acids_transforms_amd._lib import variant
dev = torch.device("cuda:0")
S, F = int(os.environ.get("STREAMS", "256")), 513
rt = A.RealtimeDGT(n_fft=1024, hop_length=256, batch_size=[S]).to(dev)
g = torch.Generator().manual_seed(1)
for n in (1, 4, 16):
    m = (torch.randn(S, n + 2, F, generator=g) ** 2 + torch.randn(S, n + 2, F, generator=g) ** 2).sqrt()
    hist, mag = m[:, :2].contiguous().to(dev), m[:, 2:].contiguous().to(dev)
    prev = (torch.rand(S, F, generator=g) * 6.28).to(dev)
    noise = torch.randn(S, n, F, generator=g).to(dev)
    args = (float(rt.gamma), 1024, 256, float(rt.tolerance), float(rt.eps))
    for kern in (0, 3):
        with variant("pghi_kernel", kern):
            for _ in range(3):
                ops.pghi_realtime(hist, mag, prev, noise, *args)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                ops.pghi_realtime(hist, mag, prev, noise, *args)
            torch.cuda.synchronize()
            print("n=%d kernel %d: %.3f ms" % (n, kern, (time.perf_counter() - t0) / 20 * 1e3), flush=True)
