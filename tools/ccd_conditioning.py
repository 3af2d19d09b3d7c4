#!/usr/bin/env python3
"""CCD++ on a poorly determined problem (tiny lambda, rows / columns with a handful of ratings): GPU path and
the fp32 CPU oracle against the SAME algorithm carried out in float64 (numpy, vectorised).  Coordinates whose
denominator lambda*|Omega| + sum v^2 is tiny amplify rounding differences by ~1/lambda; the question is
whether the GPU result is further from the float64 trajectory than the reference arithmetic is."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "cuda-recommender_amd"))
import numpy as np, torch, mfx
from mfx import synth_torch
from oracle import oracle as orc

def ccd_f64(d, W0, k, lam, t_outer, inner=1):
    rows = np.repeat(np.arange(d.rows), np.diff(d.csr_row_ptr.astype(np.int64)))
    cols = d.csr_col_idx.astype(np.int64)
    r = d.csr_val.astype(np.float64).copy()
    cnt_r = np.bincount(rows, minlength=d.rows).astype(np.float64)
    cnt_c = np.bincount(cols, minlength=d.cols).astype(np.float64)
    W = W0.astype(np.float64).copy(); H = np.zeros((k, d.cols))
    for it in range(t_outer):
        for t in range(k):
            u, v = W[t], H[t]
            if it > 0: r += u[rows] * v[cols]
            for _ in range(inner):
                g = np.bincount(cols, weights=u[rows] * r, minlength=d.cols); h = np.bincount(cols, weights=u[rows] ** 2, minlength=d.cols)
                v = np.where(cnt_c > 0, g / (lam * cnt_c + h + (cnt_c == 0)), 0.0)
                g = np.bincount(rows, weights=v[cols] * r, minlength=d.rows); h = np.bincount(rows, weights=v[cols] ** 2, minlength=d.rows)
                u = np.where(cnt_r > 0, g / (lam * cnt_r + h + (cnt_r == 0)), 0.0)
            r -= u[rows] * v[cols]
            W[t], H[t] = u, v
    return W, H

if __name__ == "__main__":
    for (rows, cols, nnz, k, lam, t) in [(300000, 2000, 4200000, 4, 0.01, 1), (300000, 2000, 4200000, 4, 0.01, 3), (300000, 2000, 4200000, 4, 0.5, 3)]:
        d = synth_torch.to_rating_data(synth_torch.synth_ratings_device(rows, cols, nnz, seed=11, device="cuda:0", sigma_rows=1.2, sigma_cols=1.8))
        W0 = mfx.initial_col(k, d.rows)
        Wt, Ht = ccd_f64(d, W0, k, lam, t)
        Wo, Ho, *_ = orc.ccdr1(d, W0, k, lam, t, 1, orc.max_threads())
        p = mfx.parameter(); p.k, p.lambda_, p.maxiter, p.maxinneriter = k, lam, t, 1
        s = mfx.CcdSolver(d, mfx.test_data_of(d), p); s.set_factors(W0.copy()); s.iterate(t); Wg, Hg = s.get_factors(); s.close()
        sc = max(np.abs(Wt).max(), np.abs(Ht).max())
        f = lambda A, B: float(np.abs(A - B).max() / sc)
        print(f"{rows}x{cols} nnz={nnz} k={k} lambda={lam} outer={t}: |gpu-f64| W {f(Wg, Wt):.2e} H {f(Hg, Ht):.2e}   |oracle-f64| W {f(Wo, Wt):.2e} H {f(Ho, Ht):.2e}"
              f"   |gpu-oracle| W {f(Wg, Wo):.2e} H {f(Hg, Ho):.2e}")
