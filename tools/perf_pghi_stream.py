"""Driver for the rocprofv3 passes over the PGHI and streaming kernels (tools/profile_pghi.sh):
offline PGHI on PGHI_B dense-noise clips x 4 s (BASELINE configs[2]) and STREAM_STEPS eager steps of the
256-stream per-hop and 1024-sample sessions (configs[4])."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import acids_transforms_amd as A
from acids_transforms_amd.streaming import StreamingDGTSession

dev = torch.device("cuda:0")
B = int(os.environ.get("PGHI_B", "1024"))
steps = int(os.environ.get("STREAM_STEPS", "20"))
g = torch.Generator(device=dev).manual_seed(1234)
if B > 0:
    d = A.DGT().to(dev)
    x = torch.randn(B, 176400, device=dev, generator=g) * 0.1
    m = d(x).abs()
    del x
    for _ in range(int(os.environ.get("PGHI_REPS", "2"))):
        ph = d.pghi(m, d.tolerance)
    torch.cuda.synchronize()
    del m, ph
for C in (256, 1024):
    if steps <= 0:
        break
    chunk = torch.randn(256, C, device=dev, generator=g) * 0.1
    sess = StreamingDGTSession(256, C, 1024, 256, 44100, device=dev, use_graph=False, mel_bands=128, mel_dtype="bf16")
    for _ in range(steps):
        sess.step(chunk)
    torch.cuda.synchronize()
print("done")
