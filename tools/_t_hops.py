import sys, torch, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import acids_transforms_amd as A
from oracle import oracle as O
dev = torch.device("cuda")
for cls in (A.STFT, A.DGT):
  for hop in (128, 256, 512):
    for L in (4096, 40000, 5000, 1024, 1536, 2047):
        torch.manual_seed(hop + L)
        x = torch.randn(3, L) * 0.1
        t = cls(n_fft=1024, hop_length=hop).to(dev)
        X = t(x.to(dev))
        y = t.invert(X).cpu()
        w = t.window[:1024].cpu(); wi = t.inv_window[:1024].cpu()
        Xr = O.stft_forward(x, w, 1024, hop)
        yr = O.istft(Xr, wi, 1024, hop)
        e1 = (X.cpu() - Xr).abs().max() / Xr.abs().max()
        e2 = (y - yr).abs().max() / yr.abs().max() if yr.numel() else 0.0
        assert y.shape == yr.shape, (y.shape, yr.shape)
        assert e1 < 1e-5 and e2 < 1e-5, (cls.__name__, hop, L, float(e1), float(e2))
    print(cls.__name__, hop, "ok", flush=True)
# polar input + batch independence at hop 128
t = A.STFT(n_fft=1024, hop_length=128).to(dev)
x0 = torch.randn(1, 30000, device=dev) * 0.1
xb = torch.cat([x0, torch.randn(499, 30000, device=dev) * 0.1])
assert torch.equal(t.invert(t(x0))[0], t.invert(t(xb))[0])
print("batch independent")
