#!/usr/bin/env python3
"""Randomised parity sweep (not part of the test suite; run on a GPU box):
random shapes / densities / skews / ranks / layouts / schedules, CCD++ and ALS against the CPU oracle.

    python3 tools/fuzz_parity.py [--cases N] [--seed S] [--seconds T]
"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "cuda-recommender_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch  # noqa: F401  (one HIP runtime in the process: before libmfx)
import mfx
from oracle import oracle as orc

ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=200)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--seconds", type=float, default=240.0)
ap.add_argument("--only", type=int, default=-1, help="run just this case index (same random stream), with diagnostics")
ap.add_argument("--layouts", default="", help="comma-separated subset of the layout / schedule kinds to draw from (e.g. reforder)")
ap.add_argument("--big", action="store_true", help="mid-size shapes (4-20 M ratings) where the automatic layout choice picks LDS / cache panels")
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
t_end = time.time() + a.seconds
fails, done, kinds, drift = [], 0, {}, 0
for case in range(a.cases):
    if time.time() > t_end:
        break
    rng = np.random.default_rng([a.seed, case])  # every case reproducible on its own (--only)
    if a.big:
        rows = int(rng.choice([20000, 90000, 300000, 700000]))
        cols = int(rng.choice([2000, 9000, 40000, 150000]))
        nnz = int(min(0.2 * rows * cols, rng.choice([4.2e6, 8e6, 2e7])))
    else:
        rows = int(rng.choice([1, 3, 17, 64, 300, 1500, 9000, 40000]))
        cols = int(rng.choice([1, 2, 40, 257, 1000, 5000, 30000]))
        dens = float(rng.choice([0.0005, 0.003, 0.02, 0.1, 0.3]))
        nnz = int(min(0.3 * rows * cols, max(1, rows * cols * dens), 400000))
    if nnz < 1:
        continue
    skip = a.only >= 0 and case != a.only
    if a.big:  # generated on the device: numpy's rejection sampling takes minutes at these sizes
        from mfx import synth_torch
        dseed, sr, sc = int(rng.integers(1 << 30)), float(rng.choice([0.3, 1.2])), float(rng.choice([0.3, 1.8]))
        d = None if skip else synth_torch.to_rating_data(synth_torch.synth_ratings_device(rows, cols, nnz, seed=dseed, device="cuda:0",
                                                                                          sigma_rows=sr, sigma_cols=sc))
    else:
        dseed, sk, tf, ef = int(rng.integers(1 << 30)), float(rng.choice([0.0, 0.5, 1.2])), float(rng.choice([0.0, 0.02])), float(rng.choice([0.0, 0.1]))
        d = None if skip else mfx.dataset.synth_ratings(rows, cols, nnz, seed=dseed, skew=sk, test_frac=tf, empty_row_frac=ef)
    k = int(rng.choice([2, 4]) if a.big else rng.choice([1, 2, 5, 8, 16, 33]))
    lam = float(rng.choice([0.01, 0.05, 0.5]))
    T = int(rng.choice([1, 1, 2, 3]))
    t = int(rng.choice([1, 2, 3]))
    p = mfx.parameter(); p.k, p.lambda_, p.maxiter, p.maxinneriter = k, lam, t, T
    lay = rng.choice(["auto", "auto", "lds", "cache"] if a.big else ["auto", "plain", "lds", "cache", "hostbuilt", "wave", "written_flat", "scatter", "scatter", "reforder", "reforder"])
    if a.layouts:
        lay = rng.choice(a.layouts.split(","))
    if lay == "plain": p.panel_rows = -1
    elif lay == "lds": p.panel_rows = int(rng.choice([16, 100, 1000, 7000]))
    elif lay == "cache": p.panel_rows = -int(rng.choice([16, 100, 5000]))
    elif lay == "hostbuilt": p.layout_build = 1
    elif lay == "wave": p.schedule, p.kernel_variant = 0, 0
    elif lay == "written_flat": p.schedule, p.kernel_variant, p.panel_rows = 0, 1, int(rng.choice([0, 50, -50]))
    elif lay == "scatter":  # persistent workgroups over random chunk ranges (r3)
        p.kernel_variant, p.panel_rows = 2, int(rng.choice([0, 7, 64, 500, 6816]))
        os.environ["MFX_SCATTER_WGS"] = str(int(rng.choice([1, 2, 3, 5, 16, 256, 100000])))
        # (r4) half of them through the sharded path (1-rank RCCL communicator) with the column pass launched by panel groups
        use_comm = bool(rng.integers(2))
        os.environ["MFX_OVERLAP_GROUPS"] = str(int(rng.choice([1, 2, 3, 4, 7, 16])))
        os.environ["MFX_COMM_RESERVE_CUS"] = str(int(rng.choice([0, 16, 200])))
    elif lay == "reforder":  # the reference's summation order: bit-identical to the oracle (r3)
        p.schedule, p.kernel_variant = 0, -1
        # (r4) the owner passes (k_ref_quad / k_ref_split, with random split thresholds and with / without the LDS table) or the
        # as-written sequence
        os.environ["MFX_REF_FUSED"] = str(int(rng.choice([1, 1, 1, 0])))
        thr = int(rng.choice([0, 0, 1, 8, 64, 300, 2048]))
        if thr: os.environ["MFX_REF_LONG"] = str(thr)
        else: os.environ.pop("MFX_REF_LONG", None)
        tabb = int(rng.choice([-1, -1, 0, 64, 1000, 20000]))  # LDS table of k_ref_quad: default rule, never, or this many bytes (partly covered tables)
        if tabb >= 0: os.environ["MFX_REF_QUAD_TAB"] = str(tabb)
        else: os.environ.pop("MFX_REF_QUAD_TAB", None)
    p.tiles_per_span = int(rng.choice([0, 2, 4, 16]))
    p.wg_waves = int(rng.choice([0, 4, 8, 16]))
    p.graph = int(rng.choice([0, -1]))
    if lay != "scatter":
        use_comm = False
    # (r4) the in-pass finalize, now also on the plain layout and its read-only sweeps: opt-in, fuzzed on a third of the cases
    if rng.integers(3) == 0:
        os.environ["MFX_FUSE_FINALIZE"] = "1"
    else:
        os.environ.pop("MFX_FUSE_FINALIZE", None)
    for env, attr in (("FUZZ_PR", "panel_rows"), ("FUZZ_TPS", "tiles_per_span"), ("FUZZ_WG", "wg_waves"), ("FUZZ_GRAPH", "graph"),
                      ("FUZZ_SCHEDULE", "schedule"), ("FUZZ_T", "maxiter")):  # overrides for bisecting a failing case (--only)
        if env in os.environ:
            setattr(p, attr, int(os.environ[env]))
    t = p.maxiter
    if skip:
        continue
    tag = f"case {case}: {rows}x{cols} nnz={d.nnz} k={k} lam={lam} T={T} t={t} {lay} pr={p.panel_rows} tps={p.tiles_per_span} wg={p.wg_waves}"
    try:
        W0 = mfx.initial_col(k, d.rows)
        Wr, Hr, rmse_ref, _, csc_ref, csr_ref = orc.ccdr1(d, W0, k, lam, t, T, orc.max_threads() if a.big else 2)
        comm = mfx.Comm(mfx.Comm.unique_id(), 0, 1, 0) if use_comm else None
        gcnt = np.ascontiguousarray(np.diff(d.csc_col_ptr.astype(np.int64)).astype(np.uint32)) if use_comm else None
        s = mfx.CcdSolver(d, mfx.test_data_of(d), p, comm=comm, global_col_nnz=gcnt, global_test_nnz=d.nnz_test)
        info = s.layout_info()
        s.set_factors(W0.copy())
        rep = s.iterate(t)
        W, H = s.get_factors()
        csc, csr = s.get_residual(d.nnz)
        s.close()
        if comm is not None:
            comm.close()
            kinds["sharded"] = kinds.get("sharded", 0) + 1
        kinds[info["csc"]["kind"]] = kinds.get(info["csc"]["kind"], 0) + 1
        scale = max(1e-6, float(np.abs(Wr).max()), float(np.abs(Hr).max()))
        err = max(float(np.abs(W - Wr).max()), float(np.abs(H - Hr).max())) / scale
        rerr = float(np.abs(np.array([r.rmse for r in rep]) - rmse_ref).max()) if d.nnz_test else 0.0
        res = max(float(np.abs(csc - csc_ref).max()), float(np.abs(csr - csr_ref).max())) if d.nnz else 0.0
        if a.only >= 0:
            dW, dH = np.abs(W - Wr), np.abs(H - Hr)
            iw, ih = np.unravel_index(np.argmax(dW), dW.shape), np.unravel_index(np.argmax(dH), dH.shape)
            print("worst W", iw, W[iw], Wr[iw], "row nnz", int(d.csr_row_ptr[iw[1] + 1] - d.csr_row_ptr[iw[1]]))
            print("worst H", ih, H[ih], Hr[ih], "col nnz", int(d.csc_col_ptr[ih[1] + 1] - d.csc_col_ptr[ih[1]]))
            print("count W entries off by > 1e-3*scale:", int((dW > 1e-3 * scale).sum()), " H:", int((dH > 1e-3 * scale).sum()), "layout", info)
        good = err < 1e-2 and rerr < 1e-4 and res < 5e-3 * max(1.0, float(np.abs(csc_ref).max()) if d.nnz else 1.0)
        if lay == "reforder":  # not a tolerance: the same bits (NaN patterns included)
            same = all(np.array_equal(x.view(np.uint32), y.view(np.uint32)) for x, y in ((W, Wr), (H, Hr), (csc, csc_ref), (csr, csr_ref)))
            if not same:
                fails.append(f"REFORDER {tag}: not bit-identical to the oracle (factor err {err:.2e}, residual err {res:.2e})")
            good = True
        if not good and d.nnz:
            # Disagreement with the fp32 oracle: on poorly determined problems (tiny lambda, segments with a
            # handful of ratings, 10^4-term sequential fp32 sums in the reference) the ORACLE is the one that
            # drifts.  Judge both against the same algorithm in float64 (tools/ccd_conditioning.py).
            from ccd_conditioning import ccd_f64
            Wt, Ht = ccd_f64(d, W0, k, lam, t, T)
            e_gpu = max(float(np.abs(W - Wt).max()), float(np.abs(H - Ht).max())) / scale
            e_orc = max(float(np.abs(Wr - Wt).max()), float(np.abs(Hr - Ht).max())) / scale
            # (no further from float64 than the reference's own fp32 arithmetic is: the disagreement is conditioning)
            good = e_gpu <= max(e_orc, 1e-4)
            tag += f" [vs float64: gpu {e_gpu:.2e}, oracle {e_orc:.2e}]"
            if good:
                drift = drift + 1
        if not good:
            fails.append(f"CCD {tag}: factor err {err:.2e} rmse err {rerr:.2e} residual err {res:.2e}")
        # ALS on the same data (every 3rd case): one half-sweep against a float64 solve of the same normal
        # equations on a sample of segments.  (Whole ALS runs are compared with the oracle in the test suite
        # on well-posed shapes; on shapes with fewer informative rows than k two fp32 trajectories drift apart
        # whatever the arithmetic, see tools/als_conditioning.py.)
        if case % 3 == 0 and d.nnz > 0:
            ka = int(rng.choice([3, 8, 20, 36, 40, 64, 70, 128]))
            H0 = mfx.initial_col(d.cols, ka)
            Y = mfx.als_half(d.csr_row_ptr, d.csr_col_idx, d.csr_val, H0, ka, lam)
            Yo = orc.als_half(d.csr_row_ptr, d.csr_col_idx, d.csr_val, H0, ka, lam, 2)
            # (r4) the as-written mode (als_exact.hip: entries in the outer loop, four waves per system): the oracle's bits
            Yx = mfx.als_half(d.csr_row_ptr, d.csr_col_idx, d.csr_val, H0, ka, lam, variant=0)
            if not np.array_equal(Yx.view(np.uint32), Yo.view(np.uint32)):
                fails.append(f"ALS-AS-WRITTEN k={ka} {tag}: not bit-identical to the oracle (max diff {float(np.abs(Yx - Yo).max()):.2e})")
            kinds["als_as_written"] = kinds.get("als_as_written", 0) + 1
            segs = rng.choice(d.rows, size=min(d.rows, 800), replace=False)
            H64 = H0.astype(np.float64)
            e_gpu = e_orc = 0.0
            sc = 1e-12
            for srow in segs:
                lo, hi = int(d.csr_row_ptr[srow]), int(d.csr_row_ptr[srow + 1])
                if hi == lo:
                    if np.any(Y[srow] != 0):
                        fails.append(f"ALS k={ka} {tag}: empty row {srow} not zero")
                    continue
                xs = H64[d.csr_col_idx[lo:hi]]
                ref = np.linalg.solve(xs.T @ xs + lam * np.eye(ka), xs.T @ d.csr_val[lo:hi].astype(np.float64))
                sc = max(sc, float(np.abs(ref).max()))
                e_gpu = max(e_gpu, float(np.abs(Y[srow] - ref).max()))
                e_orc = max(e_orc, float(np.abs(Yo[srow] - ref).max()))
            if not (np.isfinite(e_gpu) and e_gpu <= max(5.0 * e_orc, 2e-4 * sc)):
                fails.append(f"ALS k={ka} {tag}: |gpu-f64| {e_gpu / sc:.2e} vs |oracle-f64| {e_orc / sc:.2e}")
    except Exception as ex:  # noqa: BLE001
        if "exceeds the 32-bit virtual-segment range" in str(ex):  # an explicit panel size the library refuses (by design)
            kinds["rejected"] = kinds.get("rejected", 0) + 1
        else:
            fails.append(f"EXC {tag}: {type(ex).__name__}: {ex}")
    done += 1
    if done % (2 if a.big else 20) == 0:
        print(f"{done} cases, {len(fails)} failures, layouts {kinds}", flush=True)
print(f"fuzz: {done} cases, {len(fails)} failures, {drift} settled against float64 (oracle drift), layouts {kinds}")
for f in fails[:40]:
    print("  " + f)
sys.exit(1 if fails else 0)
