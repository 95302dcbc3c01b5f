"""Does the spectrum survive in the 256 MB infinity cache between the forward and the inverse kernel when the batch is
walked in sub-batches?  Step of bench.py (fused STFT + Magnitude, then ISTFT) over 1024 clips x 4 s, in chunks of c clips."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import acids_transforms_amd as A

B, L = 1024, 176400
dev = torch.device("cuda:0")
x = torch.randn(B, L, device=dev) * 0.1
stft = A.STFT().to(dev)
mag = A.Magnitude(n_mels=128, mode="unipolar", contrast="log1p").to(dev)
X0 = stft(x[:8])
mag.scale_data(X0)
del X0
chunks = [int(c) for c in (sys.argv[1].split(",") if len(sys.argv) > 1 else "1024,256,128,64,48,32,16".split(","))]
N_WARM, N_IT = int(os.environ.get("PERF_WARM", "25")), int(os.environ.get("PERF_N", "40"))


def step(c):
    for i in range(0, B, c):
        X, feat = mag.forward_fused(stft, x[i:i + c], return_spectrum=True)
        y = stft.invert(X)
    return y


for c in chunks:
    for _ in range(N_WARM if c == chunks[0] else 5):
        step(c)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(N_IT):
        step(c)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / N_IT
    print("chunk %5d clips (spectrum %7.1f MB): %.3f ms per 1024 clips  %.3e frames/s" % (
        c, c * 690 * 513 * 8 / 1e6, ms, B * 690 / ms * 1e3), flush=True)
