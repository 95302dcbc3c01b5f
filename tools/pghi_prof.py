import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["ACIDS_PGHI_PROF"] = "1"
import torch
import acids_transforms_amd as A
from acids_transforms_amd import ops
dev = torch.device("cuda:0")
B = int(os.environ.get("PGHI_B", "256"))
d = A.DGT().to(dev)
x = torch.randn(B, 176400, device=dev) * 0.1
m = d(x).abs()
ph, npops, order = ops.pghi_offline(m, d._hostf("gamma"), 1024, 256, d._hostf("tolerance"), d._hostf("eps"), debug=True)
torch.cuda.synchronize()
o = order[0][:20].cpu().numpy().view("int64")
n = o[0]
names = ["pop1(load last/root, issue nb)", "bubble rounds", "final siftdown", "neighbour update", "pushes"]
tot = sum(o[1:6])
print("pops", n, "pushes", o[6], "ticks/pop total %.0f (shader cycles; %.2f us at 2.1 GHz)" % (tot / n, tot / n / 2100.0))
for i, nm in enumerate(names):
    print("  %-32s %7.1f ticks/pop  %4.1f%%" % (nm, o[1 + i] / n, 100.0 * o[1 + i] / tot))

print("mean heap depth at pop %.2f levels, largest heap %d entries" % (o[7] / n, o[8]))
