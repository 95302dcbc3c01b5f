#!/usr/bin/env python3
"""Forward / inverse at hops other than n_fft/4 (the generic frame-at-a-time kernels), 1024 clips x 4 s."""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import acids_transforms_amd as A  # noqa: E402

dev = torch.device("cuda")
x = torch.randn(1024, 176400, device=dev) * 0.1
for n_fft, hop in ((1024, 256), (1024, 128), (1024, 512), (2048, 512), (512, 128)):
    st = A.STFT(n_fft=n_fft, hop_length=hop).to(dev)
    X = st(x)
    y = st.invert(X)
    torch.cuda.synchronize()

    def timed(fn, n=5):
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n

    tf, ti = timed(lambda: st(x)), timed(lambda: st.invert(X))
    gb = X.numel() * 8 / 1e9
    print("n_fft %4d hop %3d: %d frames/clip, spectrum %.1f GB, forward %.2f ms (%.2f TB/s written), inverse %.2f ms"
          % (n_fft, hop, X.shape[-2], gb, tf, gb / tf, ti), flush=True)
    del X, y
