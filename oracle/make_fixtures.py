#!/usr/bin/env python3
"""Generates tests/golden/*.npz by running the REFERENCE's own CPU sources
(oracle/_ref/ref_cpu, built from /root/reference/src by oracle/Makefile) on tiny seeded
datasets.  TEST INFRASTRUCTURE ONLY; runs in the build container only (the reference does
not travel).  The fixtures hold data: inputs (the dataset arrays, initial factors) and the
reference's outputs (final factors, residuals, single-step vectors, printed RMSE values).

    python oracle/make_fixtures.py            # regenerate everything
"""
from __future__ import annotations

import os
import re
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "cuda-recommender_amd"))
from mfx import dataset as ds  # noqa: E402

REF_BIN = os.path.join(HERE, "_ref", "ref_cpu")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def edge_case(seed: int) -> ds.RatingData:
    """Empty rows and columns, one column rated by (almost) every row, one row that rated
    (almost) every column, a single-entry row and a single-entry column."""
    rng = np.random.default_rng(seed)
    rows, cols = 120, 50
    base = ds.synth_ratings(rows, cols, 700, seed=seed, skew=0.8, test_frac=0.05,
                            empty_row_frac=0.1, empty_col_frac=0.1)
    r = np.repeat(np.arange(rows), np.diff(base.csr_row_ptr.astype(np.int64)))
    c = base.csr_col_idx.astype(np.int64)
    v = base.csr_val
    keep = (r != 7) & (c != 3) & (r != 11) & (c != 5)
    r, c, v = r[keep], c[keep], v[keep]
    long_col_rows = np.setdiff1d(np.arange(rows), [7, 11, 13])
    long_row_cols = np.setdiff1d(np.arange(cols), [3, 5, 9])
    r = np.concatenate([r, long_col_rows, np.full(long_row_cols.size, 7), [11]])
    c = np.concatenate([c, np.full(long_col_rows.size, 3), long_row_cols, [5]])
    v = np.concatenate([v, rng.uniform(1, 5, long_col_rows.size + long_row_cols.size + 1).astype(np.float32)])
    key, first = np.unique(r * cols + c, return_index=True)
    r, c, v = r[first], c[first], v[first].astype(np.float32)
    return ds.from_coo(rows, cols, r, c, v, base.test_row, base.test_col, base.test_val)


CASES = {
    # name: (dataset factory, k, lambda)
    "tiny": (lambda: ds.synth_ratings(60, 40, 600, seed=11, skew=0.7, test_frac=0.05), 8, 0.05),
    "small": (lambda: ds.synth_ratings(300, 200, 6000, seed=12, skew=1.0, test_frac=0.02), 8, 0.05),
    "edge": (lambda: edge_case(13), 5, 0.1),
}


def run_ref(mode, ddir, odir, k, lam, maxiter, maxinner, threads):
    os.makedirs(odir, exist_ok=True)
    cmd = [REF_BIN, mode, ddir, odir, str(k), repr(lam), str(maxiter), str(maxinner), str(threads)]
    out = subprocess.run(cmd, check=True, capture_output=True, text=True).stdout
    return out


def rd(odir, name, shape=None):
    a = np.fromfile(os.path.join(odir, name), dtype="<f4")
    return a.reshape(shape) if shape is not None else a


def main() -> None:
    if not os.path.exists(REF_BIN):
        subprocess.run(["make", "-C", HERE], check=True)
    os.makedirs(GOLDEN, exist_ok=True)
    for name, (factory, k, lam) in CASES.items():
        d = factory()
        d.validate()
        out = {}
        with tempfile.TemporaryDirectory() as tmp:
            ddir = os.path.join(tmp, "data")
            ds.write_dataset_dir(ddir, d)
            for tag, mode, t, T in (("ccd_T1", "ccd", 3, 1), ("ccd_T3", "ccd", 2, 3), ("als", "als", 3, 1)):
                res = {}
                for threads in (4, 1):  # the reference's results must not depend on -n
                    odir = os.path.join(tmp, f"{tag}_{threads}")
                    txt = run_ref(mode, ddir, odir, k, lam, t, T, threads)
                    m, n = d.rows, d.cols
                    shW = (k, m) if mode == "ccd" else (m, k)
                    shH = (k, n) if mode == "ccd" else (n, k)
                    res[threads] = dict(
                        W0=rd(odir, "W0.bin", shW), H0=rd(odir, "H0.bin", shH),
                        W=rd(odir, "W.bin", shW), H=rd(odir, "H.bin", shH),
                        rmse=np.array([float(x) for x in re.findall(r"RMSE=([0-9]+\.[0-9]+|nan|-nan|inf)", txt)]),
                        final_rmse=np.array([float(re.search(r"Test RMSE = ([0-9]+\.[0-9]+|nan|-nan)", txt).group(1))]))
                    # the reference's save_mat_t output for these factors, as raw bytes (the -save / -predict file format)
                    res[threads]["model"] = np.fromfile(os.path.join(odir, "model.bin"), dtype=np.uint8)
                    if mode == "ccd":
                        res[threads]["csc_val_final"] = rd(odir, "csc_val_final.bin")
                        res[threads]["csr_val_final"] = rd(odir, "csr_val_final.bin")
                for key in res[4]:
                    assert np.array_equal(res[4][key].view(np.uint8), res[1][key].view(np.uint8)), (name, tag, key)
                    out[f"{tag}__{key}"] = res[4][key]
                out[f"{tag}__maxiter"] = np.array([t]); out[f"{tag}__maxinner"] = np.array([T])
                assert len(res[4]["rmse"]) == t, txt
            odir = os.path.join(tmp, "steps")
            run_ref("steps", ddir, odir, k, lam, 1, 1, 1)
            for nm in ("step_v1", "step_u1", "step_csc_sub", "step_csr_sub", "step_csc_add", "step_csr_add"):
                out[nm] = rd(odir, nm + ".bin")
            out["step_gram"] = rd(odir, "step_gram.bin", (k, k))
            out["step_inv"] = rd(odir, "step_inv.bin", (k, k))
            out["step_als_row"] = np.array([int(open(os.path.join(odir, "step_als_row.txt")).read())])
            out["step_rmse_init_ccd"] = np.array([float(open(os.path.join(odir, "step_rmse_init_ccd.txt")).read())])
        out.update(rows=np.array([d.rows]), cols=np.array([d.cols]), k=np.array([k]),
                   lam=np.array([lam], np.float32),
                   csr_row_ptr=d.csr_row_ptr, csr_col_idx=d.csr_col_idx, csr_val=d.csr_val,
                   csc_col_ptr=d.csc_col_ptr, csc_row_idx=d.csc_row_idx, csc_val=d.csc_val,
                   test_row=d.test_row, test_col=d.test_col, test_val=d.test_val)
        path = os.path.join(GOLDEN, f"{name}.npz")
        np.savez_compressed(path, **out)
        print(f"{name}: rows={d.rows} cols={d.cols} nnz={d.nnz} nnz_test={d.nnz_test} k={k} "
              f"-> {path} ({os.path.getsize(path)} bytes)")
        print("   ccd_T1 rmse:", out["ccd_T1__rmse"], " als rmse:", out["als__rmse"])


if __name__ == "__main__":
    main()
