"""Step time against where the caching allocator put the step's outputs (spectrum, features, audio)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import acids_transforms_amd as A
dev = torch.device("cuda:0")
B, L = 1024, 176400
x = torch.randn(B, L, device=dev) * 0.1
stft = A.STFT().to(dev)
mag = A.Magnitude(n_mels=128, mode="unipolar", contrast="log1p").to(dev)
mag.scale_data(stft(x[:8]))
X = feat = y = None
for i in range(200):          # settle
    X, feat = mag.forward_fused(stft, x, return_spectrum=True)
    y = stft.invert(X)
torch.cuda.synchronize()
rows = []
for i in range(24):
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record()
    X, feat = mag.forward_fused(stft, x, return_spectrum=True)
    e[1].record()
    y = stft.invert(X)
    e[2].record()
    torch.cuda.synchronize()
    rows.append((X.data_ptr(), feat.data_ptr(), y.data_ptr(), e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2])))
for r in rows:
    print("X %#x (mod 2M %#8x)  feat %#x  y %#x   fwd %.4f  inv %.4f ms" % (r[0], r[0] & 0x1fffff, r[1], r[2], r[3], r[4]))
print("x", hex(x.data_ptr()))
