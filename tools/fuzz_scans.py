#!/usr/bin/env python3
"""unwrap / fdiff_* / fint_* on real input against the oracle (torch CPU restatement of utils/misc.py), bit for bit,
on random shapes and value ranges (jumps of many multiples of 2 pi, tiny differences, odd and even frame counts)."""
import os
import sys
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acids_transforms_amd.utils import misc as M  # noqa: E402
from oracle import oracle as O  # noqa: E402

dev = torch.device("cuda")
rng = np.random.RandomState(int(os.environ.get("FUZZ_SEED", "0")))
n_cases = int(os.environ.get("FUZZ_CASES", "80"))
checked = 0
for i in range(n_cases):
    shape = tuple(int(v) for v in rng.randint(1, 6, size=int(rng.randint(0, 3)))) + (int(rng.randint(1, 60)), int(rng.randint(1, 40)))
    if i % 4 == 3:      # one block per clip (>= 64 clips, rows of >= 256 bins that are not whole 64-byte segments): 2 / 4 columns per thread
        shape = (int(rng.randint(64, 80)), int(rng.randint(1, 40)), int(rng.choice([257, 300, 513, 1025, 2049, 2050, 4001])))
    scale = float(rng.choice([0.5, 3.0, 10.0, 100.0, 1e4]))
    x = torch.from_numpy((rng.randn(*shape) * scale).astype(np.float32))
    xd = x.to(dev)
    pairs = [("unwrap", M.unwrap(xd), O.unwrap(x))]
    for m in ("forward", "backward", "central"):
        pairs.append(("fdiff_" + m, getattr(M, "fdiff_" + m)(xd), O.fdiff(x, m)))
        pairs.append(("fint_" + m, getattr(M, "fint_" + m)(xd), O.fint(x, m)))
    for name, got, ref in pairs:
        assert got.shape == ref.shape, (name, shape)
        assert torch.equal(got.cpu(), ref), (name, shape, scale, float((got.cpu() - ref).abs().max()))
        checked += 1
print("%d cases, %d results identical to the oracle" % (n_cases, checked))
