"""Offline PGHI: a clip's result must not depend on the batch it rides in (the LDS share of its heap does)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import acids_transforms_amd as A
dev = torch.device("cuda")
d = A.DGT().to(dev)
ref = None
x0 = torch.randn(1, 40000, device=dev) * 0.1
for B in (1, 3, 300, 512, 700, 1100):
    x = torch.cat([x0, torch.randn(B - 1, 40000, device=dev) * 0.1]) if B > 1 else x0
    m = d(x).abs()
    y = d.invert(m, inversion_mode="pghi")
    torch.cuda.synchronize()
    assert torch.isfinite(y).all()
    # the first clip's result must not depend on how many clips share the launch (different LDS shares of the heap)
    if ref is None:
        ref = y[0].clone()
    else:
        assert torch.equal(y[0], ref), B
    print("B=%d ok" % B, flush=True)
