#!/usr/bin/env python3
"""Benchmark sweep harness: the protocol of the reference's scripts/times.sh (K x T x dataset x 3
repeats, -l 0.1; scripts/times.sh:5-66) on top of libmfx, with JSON-lines output instead of
log files.  Datasets are directories in the reference's binary format, or `synth:<rows>x<cols>x<nnz>`.

    python tools/sweep_times.py --out results.jsonl synth:6040x3706x1000000 /data/netflix
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-recommender_amd"))
import numpy as np  # noqa: E402

import mfx  # noqa: E402

KS = [1, 5, 10, 15, 20, 25, 30, 40, 50]  # scripts/times.sh:5
TS = [1, 3, 5, 7]                        # scripts/times.sh:6


def load(spec):
    if spec.startswith("synth:"):
        r, c, z = (int(x) for x in spec[6:].split("x"))
        return mfx.dataset.synth_ratings(r, c, z, seed=1234, skew=0.9)
    if spec.startswith("synthdev:"):  # generated on the GPU (numpy's rejection sampling takes minutes beyond a few million ratings)
        import torch  # noqa: F401
        from mfx import synth_torch
        r, c, z = (int(x) for x in spec[9:].split("x"))
        return synth_torch.to_rating_data(synth_torch.synth_ratings_device(r, c, z, seed=1234, device="cuda:0"))
    return mfx.dataset.read_dataset_dir(spec)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("datasets", nargs="+")
    ap.add_argument("--out", default="-")
    ap.add_argument("--iters", type=int, default=10, help="-t (times.sh uses 20/15/10 by dataset)")
    ap.add_argument("--repeats", type=int, default=3)
    ap.add_argument("--lam", type=float, default=0.1)
    ap.add_argument("--als", action="store_true", help="the ALS half of the protocol (scripts/times.sh:41-66)")
    ap.add_argument("--ks", type=int, nargs="*", default=KS)
    ap.add_argument("--ts", type=int, nargs="*", default=TS)
    ap.add_argument("--summary", default="", help="also write a table: median GPU seconds per (dataset, K, T) over the repeats")
    a = ap.parse_args()
    table = {}
    out = sys.stdout if a.out == "-" else open(a.out, "a")
    d_nnz = {}
    for spec in a.datasets:
        d = load(spec)
        d_nnz[spec] = d.nnz
        T = mfx.test_data_of(d)
        for k in a.ks:
            for t_inner in ([1] if a.als else a.ts):
                for rep in range(a.repeats):
                    p = mfx.parameter()
                    p.k, p.lambda_, p.maxiter, p.maxinneriter = k, a.lam, a.iters, t_inner
                    t0 = time.time()
                    if a.als:
                        W, H = np.zeros((d.rows, k), np.float32), mfx.initial_col(d.cols, k)
                        reports = mfx.kernel_wrapper_als_NV(d, T, W, H, p)
                        status = mfx.kernel_wrapper_als_NV.last_status
                    else:
                        W, H = mfx.initial_col(k, d.rows), np.zeros((k, d.cols), np.float32)
                        reports = mfx.kernel_wrapper_ccdpp_NV(d, T, W, H, p)
                        status = mfx.kernel_wrapper_ccdpp_NV.last_status
                    wall = time.time() - t0
                    gpu = sum(r.rank_time + r.update_time for r in reports)
                    out.write(json.dumps({"dataset": spec, "solver": "als" if a.als else "ccd", "k": k, "T": t_inner,
                                          "repeat": rep, "iters": a.iters, "status": status, "wall_s": round(wall, 4),
                                          "gpu_s": round(gpu, 5), "nnz_per_s_per_iter": round(d.nnz * a.iters / gpu, 1) if gpu else None,
                                          "rmse": [round(r.rmse, 6) for r in reports]}) + "\n")
                    out.flush()
                    table.setdefault((spec, k, t_inner), []).append((gpu, wall, reports[-1].rmse if reports else float("nan")))
    if a.summary:
        with open(a.summary, "a") as f:
            f.write("# dataset  K  T  iters  median GPU s (solver's rank + update time)  median wall s (upload + setup + solve)  nnz/s per outer iteration  final test RMSE\n")
            for (spec, k, t_inner), v in table.items():
                g = sorted(x[0] for x in v)[len(v) // 2]
                w = sorted(x[1] for x in v)[len(v) // 2]
                f.write(f"{spec} {k} {t_inner} {a.iters} {g:.4f} {w:.3f} {d_nnz[spec] * a.iters / g:.3e} {v[-1][2]:.5f}\n")


if __name__ == "__main__":
    main()
