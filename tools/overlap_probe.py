"""Dev probe: does running the write-heavy forward kernel and the read-heavy inverse kernel concurrently (two
half batches on two HIP streams) move more bytes per second than running them back to back on the full batch?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import acids_transforms_amd as A

dev = torch.device("cuda:0")
B, L = 1024, 176400
x = torch.randn(B, L, device=dev) * 0.1
st = A.STFT().to(dev)
mg = A.Magnitude(n_mels=128, mode="unipolar").to(dev)
mg.scale_data(st(x[:8]))


def seq(n=20):
    for _ in range(3):
        X, f = mg.forward_fused(st, x, return_spectrum=True); y = st.invert(X)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        X, f = mg.forward_fused(st, x, return_spectrum=True); y = st.invert(X)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3


def overlapped(n=20, parts=2):
    streams = [torch.cuda.Stream(device=dev) for _ in range(parts)]
    xs = x.chunk(parts)
    mods = [(A.STFT().to(dev), mg) for _ in range(parts)]
    def once():
        for s, xp, (stp, mgp) in zip(streams, xs, mods):
            with torch.cuda.stream(s):
                X, f = mgp.forward_fused(stp, xp, return_spectrum=True)
                y = stp.invert(X)
    for _ in range(3):
        once()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        once()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3


print("sequential, full batch        %.3f ms/step" % seq())
for p in (2, 4):
    print("%d streams x 1/%d batch         %.3f ms/step" % (p, p, overlapped(parts=p)))
print("sequential again              %.3f ms/step" % seq())
