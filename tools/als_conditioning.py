#!/usr/bin/env python3
"""ALS half-sweep accuracy against a float64 solve of the same normal equations, on well- and ill-conditioned
shapes (rank k larger than the rows/columns that carry information): GPU path vs CPU oracle vs float64."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, ROOT+"/cuda-recommender_amd")
import numpy as np, torch, mfx
from oracle import oracle as orc
def f64_half(ptr, idx, val, X, k, lam):
    n = ptr.shape[0]-1; Y = np.zeros((n,k))
    X64 = X.astype(np.float64)
    for s in range(n):
        lo, hi = ptr[s], ptr[s+1]
        if hi == lo: continue
        xs = X64[idx[lo:hi]]
        A = xs.T @ xs + lam*np.eye(k); b = xs.T @ val[lo:hi].astype(np.float64)
        Y[s] = np.linalg.solve(A, b)
    return Y
for (rows, cols, nnz, k, lam) in [(40000,1,12000,36,0.05),(17,30000,153000,64,0.05),(3,30000,26999,20,0.05),(1500,40,18000,70,0.5),(3000,400,150000,64,0.05)]:
    d = mfx.dataset.synth_ratings(rows, cols, nnz, seed=5, skew=0.5, test_frac=0.0)
    H0 = mfx.initial_col(d.cols, k)
    ref = f64_half(d.csr_row_ptr, d.csr_col_idx, d.csr_val, H0, k, lam)
    a = mfx.als_half(d.csr_row_ptr, d.csr_col_idx, d.csr_val, H0, k, lam)
    o = orc.als_half(d.csr_row_ptr, d.csr_col_idx, d.csr_val, H0, k, lam, 2)
    sc = np.abs(ref).max()
    print(f"{rows}x{cols} k={k}: |gpu-f64|={np.abs(a-ref).max()/sc:.2e}  |oracle-f64|={np.abs(o-ref).max()/sc:.2e}  |gpu-oracle|={np.abs(a-o).max()/sc:.2e}")
