import os, sys
sys.path.insert(0, "/root/repo")
import torch
import acids_transforms_amd as A
dev = torch.device("cuda:0")
x = torch.randn(1024, 176400, device=dev) * 0.1
def timeit(fn, n=5, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
for n_fft in (512, 2048, 4096):
    st = A.STFT(n_fft=n_fft, hop_length=n_fft // 4).to(dev)
    X = st(x)
    for nm in (128, None):
        mg = A.Magnitude(n_fft=n_fft, n_mels=nm, mode=None).to(dev)
        t = timeit(lambda: mg(X))
        y = mg(X)
        ti = timeit(lambda: mg.invert(y))
        print("n_fft %d n_mels %s: forward %.3f ms inverse %.3f ms (banded fwd %s)" % (n_fft, nm, t, ti, mg._band_of("mel_bank") is not None), flush=True)
    del X
