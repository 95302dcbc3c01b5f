import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from acids_transforms_amd.streaming import StreamingDGTSession
from acids_transforms_amd._lib import variant
dev = torch.device("cuda:0")
S = 256
g = torch.Generator(device=dev).manual_seed(1)
chunk = torch.randn(S, 256, device=dev, generator=g) * 0.1
tone = (0.3 * torch.sin(2 * 3.14159265 * 441.0 * torch.arange(256 * 64, device=dev) / 44100.0)).repeat(S, 1)
for name, feed in (("same chunk every step", lambda i: chunk), ("441 Hz tone, consecutive chunks", lambda i: tone[:, 256 * (i % 64):256 * (i % 64 + 1)].contiguous())):
    outs = {}
    for kern in (0, 3):
        with variant("pghi_kernel", kern):
            sess = StreamingDGTSession(S, 256, 1024, 256, 44100, device=dev, use_graph=True, mel_bands=128, mel_dtype="bf16",
                                       random_phase_below_tolerance=False)
            ys = []
            for i in range(30):
                ys.append(sess.step(feed(i)).clone())
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(200):
                sess.step(feed(i))
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 200 * 1e3
            outs[kern] = torch.stack(ys)
            print("%-34s pghi_kernel %d  %.3f ms per step" % (name, kern, dt), flush=True)
            del sess
    print("   scan path == heap:", bool(torch.equal(outs[0], outs[3])))
    assert torch.equal(outs[0], outs[3])
