import sys, os
sys.path.insert(0, os.getcwd())
import torch
import acids_transforms_amd as A
from acids_transforms_amd import ops
from acids_transforms_amd._lib import variant
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(9)
for n_fft, S, n in ((1024, 1100, 3), (512, 2500, 2), (2048, 300, 2), (256, 4100, 2), (128, 700, 4)):
    F = n_fft // 2 + 1
    rt = A.RealtimeDGT(n_fft=n_fft, hop_length=n_fft // 4, batch_size=[S]).to(dev)
    m = (torch.randn(S, n + 2, F, device=dev, generator=g) ** 2 + torch.randn(S, n + 2, F, device=dev, generator=g) ** 2).sqrt()
    m = m * (torch.rand(S, n + 2, F, device=dev, generator=g) < 0.97)        # a few dead bins everywhere
    hist, mag = m[:, :2].contiguous(), m[:, 2:].contiguous()
    prev = torch.rand(S, F, device=dev, generator=g) * 6.28
    noise = torch.randn(S, n, F, device=dev, generator=g)
    args = (float(rt.gamma), n_fft, n_fft // 4, float(rt.tolerance), float(rt.eps))
    got = ops.pghi_realtime(hist, mag, prev, noise, *args)
    with variant("pghi_kernel", 3):
        ref = ops.pghi_realtime(hist, mag, prev, noise, *args)
    with variant("pghi_kernel", 2):
        ser = ops.pghi_realtime(hist, mag, prev, noise, *args)
    print(n_fft, S, n, "scan==heap", bool(torch.equal(got, ref)), "heap==serial", bool(torch.equal(ref, ser)), flush=True)
    assert torch.equal(got, ref) and torch.equal(ref, ser)
print("ok")
