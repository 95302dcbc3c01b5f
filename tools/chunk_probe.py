#!/usr/bin/env python3
"""Does running forward and inverse on sub-batches keep the spectrum in the 256 MB Infinity Cache?
Times the bench step (fused STFT+mel forward, ISTFT inverse) on 1024 clips as one pair of launches and as
pairs of launches over sub-batches of C clips (a 64-clip spectrum is 181 MB)."""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import acids_transforms_amd as A  # noqa: E402

dev = torch.device("cuda")
B, L = 1024, 176400
x = torch.randn(B, L, device=dev) * 0.1
stft = A.STFT(sr=44100, n_fft=1024, hop_length=256).to(dev)
mag = A.Magnitude(sr=44100, n_fft=1024, n_mels=128, mode="unipolar", contrast="log1p").to(dev)
mag.scale_data(stft(x[:8]))


def step(chunk):
    for i in range(0, B, chunk):
        X, feat = mag.forward_fused(stft, x[i:i + chunk], return_spectrum=True)
        stft.invert(X)


for chunk in (1024, 512, 256, 128, 96, 64, 48, 32, 1024):
    for _ in range(3):
        step(chunk)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 10
    for _ in range(n):
        step(chunk)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print("sub-batch %4d clips (%6.1f MB spectrum): %.3f ms per 1024-clip step = %.1f Mframes/s"
          % (chunk, chunk * 690 * 513 * 8 / 1e6, ms, B * 690 / ms / 1e3), flush=True)
