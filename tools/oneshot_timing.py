#!/usr/bin/env python3
"""One-shot (host-buffer) timing of kernel_wrapper_ccdpp_NV at the bench shape: what a caller of the
reference-style entry point pays including the one-time layout build and the PCIe upload."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-recommender_amd"))
import numpy as np, torch
import mfx
from mfx import synth_torch
d = synth_torch.synth_ratings_device(480189, 17770, 99072112, seed=1234, device="cuda:0")
host = synth_torch.to_rating_data(d)
del d
torch.cuda.empty_cache()
for t in (1, 5, 15):
    p = mfx.parameter(); p.k, p.lambda_, p.maxiter = 64, 0.05, t
    W, H = mfx.initial_col(64, host.rows), np.zeros((64, host.cols), np.float32)
    t0 = time.time()
    rep = mfx.kernel_wrapper_ccdpp_NV(host, mfx.test_data_of(host), W, H, p)
    wall = time.time() - t0
    gpu = sum(r.rank_time + r.update_time for r in rep)
    print(f"maxiter={t}: wall {wall:.2f} s, of which iterations {gpu:.3f} s (+ rmse {sum(r.rmse_time for r in rep):.3f} s); "
          f"end-to-end {host.nnz * t / wall:.3g} nnz/s per outer iteration; rmse {rep[-1].rmse:.5f}")
