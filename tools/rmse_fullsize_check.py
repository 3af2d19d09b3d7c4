#!/usr/bin/env python3
"""Test RMSE per outer iteration at the Netflix shape: GPU path vs the fp32 CPU oracle (all host cores)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "cuda-recommender_amd"))
import numpy as np, torch, mfx
from mfx import synth_torch
from oracle import oracle as orc
k, lam, t = int(sys.argv[1]) if len(sys.argv) > 1 else 8, 0.05, 3
d = synth_torch.to_rating_data(synth_torch.synth_ratings_device(480189, 17770, 99072112, seed=1234, device="cuda:0"))
W0 = mfx.initial_col(k, d.rows)
t0 = time.time()
Wr, Hr, rmse_ref, *_ = orc.ccdr1(d, W0, k, lam, t, 1, orc.max_threads())
print(f"oracle: {time.time() - t0:.1f} s on {orc.max_threads()} threads", flush=True)
p = mfx.parameter(); p.k, p.lambda_, p.maxiter, p.maxinneriter = k, lam, t, 1
s = mfx.CcdSolver(d, mfx.test_data_of(d), p); s.set_factors(W0.copy()); rep = s.iterate(t); W, H = s.get_factors(); s.close()
rm = np.array([r.rmse for r in rep])
sc = max(np.abs(Wr).max(), np.abs(Hr).max())

def ccd_f64_torch(d, W0, k, lam, t_outer, dev="cuda:0"):
    """The same algorithm in float64 on the device (torch only as a calculator), with the test RMSE per iteration."""
    f8 = torch.float64
    rows = torch.repeat_interleave(torch.arange(d.rows, device=dev), torch.from_numpy(np.diff(d.csr_row_ptr.astype(np.int64))).to(dev))
    cols = torch.from_numpy(d.csr_col_idx.astype(np.int64)).to(dev)
    r = torch.from_numpy(d.csr_val.astype(np.float64)).to(dev)
    cnt_r = torch.bincount(rows, minlength=d.rows).to(f8); cnt_c = torch.bincount(cols, minlength=d.cols).to(f8)
    W = torch.from_numpy(W0.astype(np.float64)).to(dev); H = torch.zeros((k, d.cols), dtype=f8, device=dev)
    tr = torch.from_numpy(d.test_row.astype(np.int64)).to(dev); tc = torch.from_numpy(d.test_col.astype(np.int64)).to(dev)
    tv = torch.from_numpy(d.test_val.astype(np.float64)).to(dev)
    out = []
    for it in range(t_outer):
        for t in range(k):
            u, v = W[t], H[t]
            if it > 0: r += u[rows] * v[cols]
            ur = u[rows]
            g = torch.bincount(cols, weights=ur * r, minlength=d.cols); h = torch.bincount(cols, weights=ur * ur, minlength=d.cols)
            v = torch.where(cnt_c > 0, g / (lam * cnt_c + h + (cnt_c == 0)), torch.zeros_like(g))
            vc = v[cols]
            g = torch.bincount(rows, weights=vc * r, minlength=d.rows); h = torch.bincount(rows, weights=vc * vc, minlength=d.rows)
            u = torch.where(cnt_r > 0, g / (lam * cnt_r + h + (cnt_r == 0)), torch.zeros_like(g))
            r -= u[rows] * vc
            W[t], H[t] = u, v
        pred = (W[:, tr] * H[:, tc]).sum(0)
        out.append(float(torch.sqrt(((pred - tv) ** 2).mean())))
    return W.cpu().numpy(), H.cpu().numpy(), np.array(out)

Wt, Ht, rmse_t = ccd_f64_torch(d, W0, k, lam, t)
print("rmse f64   ", rmse_t)
print(f"vs float64: max |rmse diff| gpu {np.abs(rm - rmse_t).max():.2e}, oracle {np.abs(rmse_ref - rmse_t).max():.2e}; "
      f"factors/scale gpu {max(np.abs(W - Wt).max(), np.abs(H - Ht).max()) / sc:.2e}, oracle {max(np.abs(Wr - Wt).max(), np.abs(Hr - Ht).max()) / sc:.2e}")
print("rmse gpu   ", rm)
print("rmse oracle", rmse_ref)
print(f"max |rmse diff| {np.abs(rm - rmse_ref).max():.2e}; factors: max |W diff|/scale {np.abs(W - Wr).max() / sc:.2e}, |H diff|/scale {np.abs(H - Hr).max() / sc:.2e}")
