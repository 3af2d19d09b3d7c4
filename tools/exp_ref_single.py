#!/usr/bin/env python3
"""(r4) the reference-order sweep on a handful of long dense columns: kernel duration per entry of the longest chain
(run under rocprofv3 --kernel-trace)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "cuda-recommender_amd"))
import numpy as np
import torch  # noqa
import mfx
from oracle import oracle as orc
rows, cols = 240000, int(sys.argv[1]) if len(sys.argv) > 1 else 4
rng = np.random.default_rng(0)
ptr = (np.arange(cols + 1, dtype=np.uint64) * rows).astype(np.uint32)
idx = np.tile(np.arange(rows, dtype=np.uint32), cols)
val = rng.uniform(1, 5, rows * cols).astype(np.float32)
u = rng.uniform(0.001, 0.1, rows).astype(np.float32)
for rep in range(3):
    v = mfx.rank_one_sweep(ptr, idx, val, u, 0.05, -1)
ref = orc.rank_one_sweep(ptr, idx, val, u, 0.05, 1)
print("bit-identical to the oracle:", np.array_equal(v.view(np.uint32), ref.view(np.uint32)))
