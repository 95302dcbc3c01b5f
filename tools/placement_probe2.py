"""Plain forward (at_stft_forward) into different output buffers / offsets: does the buffer's placement set the speed?"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import acids_transforms_amd as A
from acids_transforms_amd._lib import lib
from acids_transforms_amd.ops import ptr, stream_ptr, check
dev = torch.device("cuda:0")
B, L, T, F = 1024, 176400, 690, 513
x = torch.randn(B, L, device=dev) * 0.1
stft = A.STFT().to(dev)
win = stft.window[:1024].contiguous()
X0 = stft(x[:2])      # initialises the library for this device
n_el = B * T * F
slack = 1 << 20     # complex elements of slack behind every candidate (8 MB)
bufs = [torch.empty(n_el + slack, dtype=torch.complex64, device=dev) for _ in range(int(os.environ.get("PROBE_BUFS", "4")))]


def run(p):
    check(lib().at_stft_forward(ptr(x), B, L, L, T, 1024, 256, 1, ptr(win), ctypes.c_void_p(p), None, stream_ptr()), "fwd")


def timeit(p, n=15):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): run(p)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n


for _ in range(150): run(bufs[0].data_ptr())     # settle
torch.cuda.synchronize()
for rnd in range(3):
    print("round", rnd, "  ".join("buf%d @%#x: %.4f" % (i, b.data_ptr(), timeit(b.data_ptr())) for i, b in enumerate(bufs)), flush=True)
base = bufs[0].data_ptr()
for off in (0, 512, 4096, 65536, 1 << 20, (1 << 20) + 512, 2 << 20, 3 << 20):
    print("buf0 + %8d B: %.4f  %.4f" % (off, timeit(base + off), timeit(base + off)), flush=True)
print("x @%#x" % x.data_ptr())
