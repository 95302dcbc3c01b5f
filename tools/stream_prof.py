"""Dev: kernel mix of the streaming DGT round trip (256 streams x 1024-sample chunks)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from acids_transforms_amd.streaming import StreamingDGTSession
dev = torch.device("cuda:0")
S, C = int(os.environ.get("STREAMS", "256")), int(os.environ.get("CHUNK", "1024"))
chunk = torch.randn(S, C, device=dev) * 0.1
sess = StreamingDGTSession(S, C, 1024, 256, 44100, device=dev, use_graph=False)
for _ in range(10):
    sess.step(chunk)
torch.cuda.synchronize()
# 16 different chunks in turn (the same chunk again and again ties every candidate of the flood with its predecessor)
chunks = [torch.randn(S, C, device=dev) * 0.1 for _ in range(16)]
for i in range(40):
    sess.step(chunks[i % 16])
torch.cuda.synchronize()
