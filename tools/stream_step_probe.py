"""ms per streaming step (256 streams, bf16 mel, hipGraph) for chunk sizes 256 / 1024 / 4096, realtime PGHI on the rank
fast path (default) against the cooperative heap kernel alone (at_set_variant pghi_kernel = 3)."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from acids_transforms_amd._lib import variant
from acids_transforms_amd.streaming import StreamingDGTSession

dev = torch.device("cuda:0")
S = int(os.environ.get("STREAMS", "256"))
for C in (256, 1024, 4096):
    for kern in (0, 4, 3, 0, 4):
        with variant("pghi_kernel", kern):
            # a different chunk every step (32 of them in turn): fed the SAME hop-sized chunk again and again, every frame of
            # a stream is the frame before it, all candidates of the flood tie, and the heap path is what gets timed
            chunks = [torch.randn(S, C, device=dev) * 0.1 for _ in range(32)]
            sess = StreamingDGTSession(S, C, 1024, 256, 44100, device=dev, use_graph=True, mel_bands=128, mel_dtype="bf16")
            for i in range(20):
                sess.step(chunks[i % 32])
            torch.cuda.synchronize()
            n = 300 if C == 256 else 100
            t0 = time.perf_counter()
            for i in range(n):
                sess.step(chunks[i % 32])
            torch.cuda.synchronize()
            print("chunk %5d  pghi_kernel %d  %.3f ms per step" % (C, kern, (time.perf_counter() - t0) / n * 1e3), flush=True)
            del sess
