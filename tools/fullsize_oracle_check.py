#!/usr/bin/env python3
"""GPU path vs the fp32 CPU oracle at BASELINE configs[2]/[3] size (480189 x 17770, 99 M ratings, k = 64).

    python tools/fullsize_oracle_check.py [--k 64] [--iters 3] [--sigma-cols 1.8] [--solver ccd|als] [--f64]

Prints one JSON line per run: per-iteration test RMSE of the GPU path and of the oracle (the
bit-exact restatement of the reference's ccdr1_OMP / ALS_OMP, all usable host cores), their gap, the
factor gap relative to scale, and (--f64, CCD only) the same algorithm in float64 as a third opinion.
--sigma-cols sets the item-popularity skew of the generator: 1.8 (bench default) gives columns with
about 1.2e6 ratings, 1.12 gives a longest column of about 2.3e5 -- the real Netflix figure.
The committed output lives in profiles/r02_fullsize_oracle.txt; tests/test_gpu_fullsize.py asserts
the same comparison.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "cuda-recommender_amd"))


def ccd_f64_torch(torch, np, d, W0, k, lam, t_outer, dev="cuda:0"):
    """The same algorithm in float64 on the device (torch only as a calculator)."""
    f8 = torch.float64
    rows = torch.repeat_interleave(torch.arange(d.rows, device=dev),
                                   torch.from_numpy(np.diff(d.csr_row_ptr.astype(np.int64))).to(dev))
    cols = torch.from_numpy(d.csr_col_idx.astype(np.int64)).to(dev)
    r = torch.from_numpy(d.csr_val.astype(np.float64)).to(dev)
    cnt_r = torch.bincount(rows, minlength=d.rows).to(f8)
    cnt_c = torch.bincount(cols, minlength=d.cols).to(f8)
    W = torch.from_numpy(W0.astype(np.float64)).to(dev)
    H = torch.zeros((k, d.cols), dtype=f8, device=dev)
    tr = torch.from_numpy(d.test_row.astype(np.int64)).to(dev)
    tc = torch.from_numpy(d.test_col.astype(np.int64)).to(dev)
    tv = torch.from_numpy(d.test_val.astype(np.float64)).to(dev)
    out = []
    for it in range(t_outer):
        for t in range(k):
            u, v = W[t], H[t]
            if it > 0:
                r += u[rows] * v[cols]
            ur = u[rows]
            g = torch.bincount(cols, weights=ur * r, minlength=d.cols)
            h = torch.bincount(cols, weights=ur * ur, minlength=d.cols)
            v = torch.where(cnt_c > 0, g / (lam * cnt_c + h + (cnt_c == 0)), torch.zeros_like(g))
            vc = v[cols]
            g = torch.bincount(rows, weights=vc * r, minlength=d.rows)
            h = torch.bincount(rows, weights=vc * vc, minlength=d.rows)
            u = torch.where(cnt_r > 0, g / (lam * cnt_r + h + (cnt_r == 0)), torch.zeros_like(g))
            r -= u[rows] * vc
            W[t], H[t] = u, v
        pred = torch.zeros(tr.numel(), dtype=f8, device=dev)
        for t in range(k):
            pred += W[t][tr] * H[t][tc]
        out.append(float(torch.sqrt(((pred - tv) ** 2).mean())))
    return W.cpu().numpy(), H.cpu().numpy(), np.array(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=480189)
    ap.add_argument("--cols", type=int, default=17770)
    ap.add_argument("--nnz", type=int, default=99072112)
    ap.add_argument("--k", type=int, default=64)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--lam", type=float, default=0.05)
    ap.add_argument("--sigma-cols", type=float, default=1.8)
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--solver", choices=["ccd", "als"], default="ccd")
    ap.add_argument("--f64", action="store_true")
    a = ap.parse_args()
    import numpy as np
    import torch
    import mfx
    from mfx import synth_torch
    from oracle import oracle as orc

    d = synth_torch.to_rating_data(synth_torch.synth_ratings_device(a.rows, a.cols, a.nnz, seed=a.seed, device="cuda:0",
                                                                    sigma_cols=a.sigma_cols))
    torch.cuda.empty_cache()
    longest_col = int(np.diff(d.csc_col_ptr.astype(np.int64)).max())
    longest_row = int(np.diff(d.csr_row_ptr.astype(np.int64)).max())
    threads = orc.max_threads()
    out = {"solver": a.solver, "shape": [d.rows, d.cols, int(d.nnz)], "k": a.k, "lambda": a.lam, "iters": a.iters,
           "sigma_cols": a.sigma_cols, "longest_col": longest_col, "longest_row": longest_row, "oracle_threads": threads}
    p = mfx.parameter()
    p.k, p.lambda_, p.maxiter, p.maxinneriter = a.k, a.lam, a.iters, 1
    if a.solver == "ccd":
        W0 = mfx.initial_col(a.k, d.rows)
        t0 = time.time()
        Wr, Hr, rmse_ref, *_ = orc.ccdr1(d, W0, a.k, a.lam, a.iters, 1, threads)
        out["oracle_seconds"] = round(time.time() - t0, 1)
        s = mfx.CcdSolver(d, mfx.test_data_of(d), p)
        s.set_factors(W0.copy())
        rep = s.iterate(a.iters)
        W, H = s.get_factors()
        s.close()
    else:
        H0 = mfx.initial_col(d.cols, a.k)
        t0 = time.time()
        Wr, Hr, rmse_ref, _ = orc.als(d, H0, a.k, a.lam, a.iters, threads)
        out["oracle_seconds"] = round(time.time() - t0, 1)
        s = mfx.AlsSolver(d, mfx.test_data_of(d), p)
        s.set_factors(H0.copy())
        rep = s.iterate(a.iters)
        W, H = s.get_factors()
        s.close()
    rm = np.array([r.rmse for r in rep])
    sc = float(max(np.abs(Wr).max(), np.abs(Hr).max()))
    out.update({"rmse_gpu": [round(x, 7) for x in rm], "rmse_oracle": [round(float(x), 7) for x in rmse_ref],
                "rmse_gap": [float("%.3g" % abs(x - y)) for x, y in zip(rm, rmse_ref)],
                "W_gap_over_scale": float("%.3g" % (np.abs(W - Wr).max() / sc)),
                "H_gap_over_scale": float("%.3g" % (np.abs(H - Hr).max() / sc)),
                "golden_compare_errors_W": int(np.sum(np.abs(W - Wr) > 0.1 * np.abs(Wr))),
                "golden_compare_errors_H": int(np.sum(np.abs(H - Hr) > 0.1 * np.abs(Hr)))})
    if a.f64 and a.solver == "ccd":
        Wt, Ht, rmse_t = ccd_f64_torch(torch, np, d, W0, a.k, a.lam, a.iters)
        out.update({"rmse_f64": [round(float(x), 7) for x in rmse_t],
                    "gpu_vs_f64": float("%.3g" % np.abs(rm - rmse_t).max()),
                    "oracle_vs_f64": float("%.3g" % np.abs(rmse_ref - rmse_t).max()),
                    "gpu_factors_vs_f64": float("%.3g" % (max(np.abs(W - Wt).max(), np.abs(H - Ht).max()) / sc)),
                    "oracle_factors_vs_f64": float("%.3g" % (max(np.abs(Wr - Wt).max(), np.abs(Hr - Ht).max()) / sc))})
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
