"""Dev: per-step times of the bench step in a fresh process, with and without a device pre-warm (is the ~15-step ramp
the clock governor or something in the stack?).   python tools/ramp_probe.py [prewarm_ms] [kind]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import acids_transforms_amd as A

pre_ms = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
kind = sys.argv[2] if len(sys.argv) > 2 else "fill"
dev = torch.device("cuda:0")
B, L = 1024, 176400
x = torch.randn(B, L, device=dev) * 0.1
stft = A.STFT().to(dev)
mag = A.Magnitude(n_mels=128, mode="unipolar", contrast="log1p").to(dev)
mag.scale_data(stft(x[:8]))
torch.cuda.synchronize()
if pre_ms > 0:
    buf = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < pre_ms:
        if kind == "fill":
            for _ in range(8):
                buf.fill_(1)
        else:
            for _ in range(2):
                mag.forward_fused(stft, x[:128], return_spectrum=True)
        torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(61)]
keep = None
for i in range(60):
    ev[i].record()
    X, f = mag.forward_fused(stft, x, return_spectrum=True)
    y = stft.invert(X)
    keep = (X, f, y)
ev[60].record()
torch.cuda.synchronize()
t = [ev[i].elapsed_time(ev[i + 1]) for i in range(60)]
print("prewarm %.0f ms (%s): steps 0-4 %s | 5-24 avg %.3f | 25-59 avg %.3f" % (
    pre_ms, kind, " ".join("%.2f" % v for v in t[:5]), sum(t[5:25]) / 20, sum(t[25:]) / 35))
print("   ", " ".join("%.2f" % v for v in t))
