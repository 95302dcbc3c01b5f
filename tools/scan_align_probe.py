"""Column-walk scans at F = 513 against F = 512 (aligned rows): is the row misalignment what holds them at 3.4 TB/s?"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from acids_transforms_amd import ops

dev = torch.device("cuda:0")
B, T = 1024, 690


def timeit(fn, n=20, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


for F in (513, 512, 576):
    y = torch.randn(B, T, F, device=dev)
    X = torch.randn(B, T, F, 2, device=dev)
    X = torch.view_as_complex(X)
    t1 = timeit(lambda: ops.phase_integrate(y, "forward"))
    t2 = timeit(lambda: ops.phase_scan(X, "forward"))
    t3 = timeit(lambda: ops.phase_scan(X, "angle"))
    t4 = timeit(lambda: ops.phase_integrate(y, "central"))
    t5 = timeit(lambda: ops.phase_integrate(y[:, :T - 1].contiguous() if False else y_odd, "central")) if (y_odd := y[:, :T - 1].contiguous()) is not None else 0
    print("F=%d  IF.invert forward %.3f ms (%.2f TB/s)   IF forward %.3f ms (%.2f TB/s)   angle %.3f ms (%.2f TB/s)" % (
        F, t1, B * T * F * 8 / t1 / 1e9, t2, B * T * F * 12 / t2 / 1e9, t3, B * T * F * 12 / t3 / 1e9), flush=True)
    print("F=%d  IF.invert central even T %.3f ms (%.2f TB/s)   odd T %.3f ms (%.2f TB/s)" % (
        F, t4, B * T * F * 8 / t4 / 1e9, t5, B * (T - 1) * F * 8 / t5 / 1e9), flush=True)
    del y, X, y_odd

