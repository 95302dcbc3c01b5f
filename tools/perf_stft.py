"""Quick kernel timing (dev tool, not the bench contract): STFT fwd / ISTFT at BASELINE config 2 size."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import acids_transforms_amd as A

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
L = 176400
dev = torch.device("cuda:0")
x = torch.randn(B, L, device=dev) * 0.1
m = A.STFT().to(dev)
T = 1 + L // 256
frames = B * T


def timeit(fn, n=10, warm=3):
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


X = m(x)
t_f = timeit(lambda: m(x))
t_i = timeit(lambda: m.invert(X))
m.eager_phase = True
t_fp = timeit(lambda: m(x))
m.eager_phase = False
mag, ph = X.abs(), X.angle()
t_ip = timeit(lambda: m._istft(mag=mag, phase=ph))
for name, t, bytes_per_frame in [("stft_fwd", t_f, 5128), ("istft", t_i, 5128), ("stft_fwd+phase", t_fp, 5128 + 2052),
                                 ("istft_polar", t_ip, 5128)]:
    print("%-16s %8.3f ms  %8.1f Mframes/s  %6.2f TB/s algorithmic (%.0f%% of 8 TB/s)" % (
        name, t, frames / t / 1e3, frames * bytes_per_frame / t / 1e9, frames * bytes_per_frame / t / 1e9 / 8 * 100))
