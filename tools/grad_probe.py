import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import acids_transforms_amd as A
from acids_transforms_amd import ops
dev = torch.device("cuda:0")
d = A.DGT().to(dev)
mag = torch.rand(1024, 690, 513, device=dev) + 0.01
def timeit(fn, n=10, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
t = timeit(lambda: ops.pghi_gradients(mag, d.gamma, 1024, 256))
print("pghi gradients 1024 clips: %.3f ms (%.2f TB/s on 12 B/bin)" % (t, 1024*690*513*12/t/1e9))
