import sys, os, ctypes
sys.path.insert(0, os.getcwd())
import torch
from acids_transforms_amd.streaming import StreamingDGTSession
from acids_transforms_amd import _lib
dev = torch.device("cuda:0")
L = _lib.lib()
S = 256
g = torch.Generator(device=dev).manual_seed(1)
chunk = torch.randn(S, 256, device=dev, generator=g) * 0.1
sess = StreamingDGTSession(S, 256, 1024, 256, 44100, device=dev, use_graph=False, random_phase_below_tolerance=False)
for i in range(10):
    sess.step(chunk)
torch.cuda.synchronize()
buf = (ctypes.c_uint * 8)()
L.at_dev_rt_stats(buf, 1)
for i in range(10):
    sess.step(chunk)
torch.cuda.synchronize()
L.at_dev_rt_stats(buf, 1)
print("frames", buf[0], "ok", buf[1], "declined reseed", buf[3], "tie", buf[4])
m = sess.mag_out[0, 0]
print("tolerance", float(sess.dgt.tolerance), "max", float(m.max()), "live bins", int((m > float(sess.dgt.tolerance) * m.max()).sum()),
      "distinct magnitudes", int(torch.unique(m).numel()), "of", m.numel())
print(m[:16].tolist())
