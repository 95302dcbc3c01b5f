import torch
a = torch.empty(1 << 30, device="cuda", dtype=torch.float32); b = torch.empty_like(a)
def t(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True); s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n
print("box: copy_ %.2f TB/s  fill_ %.2f TB/s" % (8 * (1 << 30) / t(lambda: b.copy_(a)) / 1e9, 4 * (1 << 30) / t(lambda: a.fill_(1.0)) / 1e9))
