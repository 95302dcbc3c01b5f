"""Dev check: batches whose tensors pass 2^31 elements (6400 clips x 690 frames x 513 bins = 2.27e9 complex values):
the whole batch in one call must equal the same clips processed 800 at a time -- 64-bit indexing everywhere."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import acids_transforms_amd as A

dev = torch.device("cuda:0")
B, L, C = 6400, 176400, 800
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randn(B, L, device=dev, generator=g) * 0.1
stft = A.STFT().to(dev)
mag = A.Magnitude(n_mels=128).to(dev)
mag.scale_data(stft(x[:8]))
pol = A.Polar().to(dev)
pol.scale_data(stft(x[:8]))
car = A.Cartesian().to(dev)
car.scale_data(stft(x[:8]))
mf = A.MFCC().to(dev)

X = stft(x)
assert X.numel() > 2 ** 31, X.numel()
Xf, feat = mag.forward_fused(stft, x, return_spectrum=True)
y = stft.invert(X)
m2 = mag(X)
mfcc = mf(x)
checks = {"stft": 0.0, "fused X": 0.0, "fused feat": 0.0, "istft": 0.0, "magnitude": 0.0, "mfcc": 0.0}
for i in range(0, B, C):
    xs = x[i:i + C]
    Xs = stft(xs)
    checks["stft"] = max(checks["stft"], float((Xs - X[i:i + C]).abs().max()))
    Xfs, fs = mag.forward_fused(stft, xs, return_spectrum=True)
    checks["fused X"] = max(checks["fused X"], float((Xfs - Xf[i:i + C]).abs().max()))
    checks["fused feat"] = max(checks["fused feat"], float((fs - feat[i:i + C]).abs().max()))
    checks["istft"] = max(checks["istft"], float((stft.invert(Xs) - y[i:i + C]).abs().max()))
    checks["magnitude"] = max(checks["magnitude"], float((mag(Xs) - m2[i:i + C]).abs().max()))
    checks["mfcc"] = max(checks["mfcc"], float((mf(xs) - mfcc[i:i + C]).abs().max()))
print("max |whole batch - chunked| (bit-identical expected):", checks, flush=True)
assert all(v == 0.0 for v in checks.values()), checks
del Xf, feat, y, m2, mfcc
# stacked representations on the > 2^31-element spectrum (last chunk compared)
for name, tr in (("polar", pol), ("cartesian", car)):
    yy = tr(X)
    ys = tr(X[B - C:])
    d = float((yy[B - C:] - ys).abs().max())
    back = tr.invert(yy)
    d2 = float((back[B - C:] - tr.invert(ys)).abs().max())
    print(name, "forward diff", d, "invert diff", d2, flush=True)
    assert d == 0.0 and d2 == 0.0
    del yy, ys, back
print("ok")
