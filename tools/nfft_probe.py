"""Dev timing: STFT forward / inverse at other FFT sizes (1024 clips x 4 s, hop = n_fft / 4)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import acids_transforms_amd as A

dev = torch.device("cuda:0")
B, L = 1024, 176400
x = torch.randn(B, L, device=dev) * 0.1


def timeit(fn, n=int(os.environ.get("PERF_N", "5")), warm=int(os.environ.get("PERF_WARM", "2"))):
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


for n_fft in [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["512", "1024", "2048", "4096"])]:
    st = A.STFT(n_fft=n_fft, hop_length=n_fft // 4).to(dev)
    X = st(x)
    T, F = X.shape[-2], X.shape[-1]
    tf, ti = timeit(lambda: st(x)), timeit(lambda: st.invert(X))
    fwd_bytes = B * T * (n_fft // 4 * 4 + F * 8)
    print("n_fft %5d  frames/clip %4d  forward %.3f ms (%.2f TB/s)  inverse %.3f ms (%.2f TB/s)"
          % (n_fft, T, tf, fwd_bytes / tf / 1e9, ti, fwd_bytes / ti / 1e9), flush=True)
