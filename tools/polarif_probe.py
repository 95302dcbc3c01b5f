"""Polar / PolarIF forward and invert at 1024 x 690 x 513 (PROBE_SCAN_FLAT=1: flattened-column scans through at_set_variant)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import acids_transforms_amd as A
dev = torch.device("cuda:0")
if os.environ.get("PROBE_SCAN_FLAT"):
    from acids_transforms_amd import _lib as _L
    _L.check(_L.lib().at_set_variant(_L.VARIANTS["scan_layout"], 1), "at_set_variant")
B, T, F = 1024, 690, 513
X = torch.view_as_complex(torch.randn(B, T, F, 2, device=dev))


def timeit(fn, n=10, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n


for name, tr in (("polar", A.Polar()), ("polar_if", A.PolarIF()), ("cartesian", A.Cartesian())):
    tr = tr.to(dev)
    tr.scale_data(X[:4])
    y = tr(X)
    print("%-9s forward %.3f ms   invert %.3f ms   [%s]" % (name, timeit(lambda: tr(X)), timeit(lambda: tr.invert(y)),
                                                           "flat" if os.environ.get("PROBE_SCAN_FLAT") else "default"), flush=True)
    del y
