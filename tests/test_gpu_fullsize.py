"""Full BASELINE.json size (configs[2] / configs[3]: 480189 x 17770, Z = 99 072 112, k = 64) on the GPU: compared
with the CPU oracle itself (it costs 6 s for CCD++ -t 3 and 40 s for one ALS iteration on the box's 16 host
threads) and checked through size-independent properties:

  * both residual copies hold bit-identical values under the CSR<->CSC permutation (every element
    sees the same operands in the same order in the two fused passes);
  * residual identity  r_ij = R_ij - sum_t W[t,i] H[t,j]  for EVERY stored rating (fp64 check);
  * reported test RMSE equals an independent fp64 evaluation from the returned factors and
    decreases monotonically; a second run is bitwise identical (no atomics anywhere);
  * the as-written schedule reaches the same factors as the fused one.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROWS, COLS, NNZ, K = 480189, 17770, 99072112, 64


@pytest.fixture(scope="module")
def big():
    import torch
    import mfx
    from mfx import synth_torch
    assert mfx.device_count() >= 1 and torch.cuda.is_available()
    d = synth_torch.synth_ratings_device(ROWS, COLS, NNZ, seed=2024, device="cuda:0")
    torch.cuda.synchronize()
    return mfx, torch, d


def _solve(mfx, d, iters, schedule=1, variant=1):
    p = mfx.parameter()
    p.k, p.lambda_, p.schedule, p.kernel_variant = K, 0.05, schedule, variant
    s = mfx.CcdSolver(None, None, p, device_arrays=d)
    s.set_factors(mfx.initial_col(K, ROWS))
    rep = s.iterate(iters)
    W, H = s.get_factors()
    return s, rep, W, H


def test_fullsize_properties(big):
    mfx, torch, d = big
    s, rep, W, H = _solve(mfx, d, 2)
    csc, csr = s.get_residual(NNZ)
    s.close()
    rmse = [r.rmse for r in rep]
    assert rmse[1] < rmse[0] < 2.0 and all(np.isfinite(rmse))

    dev = torch.device("cuda:0")
    Wt, Ht = torch.from_numpy(W).to(dev), torch.from_numpy(H).to(dev)
    csc_t, csr_t = torch.from_numpy(csc).to(dev), torch.from_numpy(csr).to(dev)
    # (1) the two copies agree bit for bit under the permutation
    assert torch.equal(csr_t[d["csc_of_csr"]].view(torch.int32), csc_t.view(torch.int32))
    # (2) residual identity over all stored ratings, fp64, in chunks
    row_of = torch.repeat_interleave(torch.arange(ROWS, device=dev), (d["csr_row_ptr"][1:] - d["csr_row_ptr"][:-1]).long())
    col_of = d["csr_col_idx"].long()
    worst = 0.0
    step = 1 << 23
    for b in range(0, NNZ, step):
        e = min(NNZ, b + step)
        pred = torch.zeros(e - b, dtype=torch.float64, device=dev)
        for t in range(K):
            pred += (Wt[t][row_of[b:e]] * Ht[t][col_of[b:e]]).double()
        worst = max(worst, float((d["csr_val"][b:e].double() - pred - csr_t[b:e].double()).abs().max()))
    assert worst < 5e-4, worst
    # (4) reported RMSE vs an independent fp64 evaluation
    ti, tj = d["test_row"].long(), d["test_col"].long()
    pred = torch.zeros(ti.numel(), dtype=torch.float64, device=dev)
    for t in range(K):
        pred += (Wt[t][ti] * Ht[t][tj]).double()
    ref = float(torch.sqrt(((pred - d["test_val"].double()) ** 2).mean()))
    assert abs(ref - rmse[-1]) < 1e-9

    # (3) bitwise reproducible
    s2, rep2, W2, H2 = _solve(mfx, d, 2)
    s2.close()
    assert np.array_equal(W.view(np.uint32), W2.view(np.uint32)) and np.array_equal(H.view(np.uint32), H2.view(np.uint32))
    assert [r.rmse for r in rep2] == rmse


def test_fullsize_as_written_schedule_agrees(big):
    mfx, torch, d = big
    s1, rep1, W1, H1 = _solve(mfx, d, 1, schedule=1)
    s1.close()
    s0, rep0, W0, H0 = _solve(mfx, d, 1, schedule=0, variant=1)
    s0.close()
    assert abs(rep0[0].rmse - rep1[0].rmse) < 1e-5
    assert np.max(np.abs(W0 - W1)) < 1e-3 * np.max(np.abs(W1)) and np.max(np.abs(H0 - H1)) < 1e-3 * np.max(np.abs(H1))


def test_fullsize_als_normal_equations(big):
    """BASELINE configs[3] (Netflix shape, k = 64, ALS): after one iteration every sampled row of W and
    column of H must satisfy its normal equations (X_Omega^T X_Omega + lambda I) y = X_Omega^T r, checked
    against an independent fp64 solve; RMSE must fall between iterations.  (H-half is checked against
    the W it was computed from, i.e. the returned W; the W-half against the H of the previous
    iteration, so the row check runs on a 1-iteration solve.)"""
    mfx, torch, d = big
    from mfx import synth_torch
    host = synth_torch.to_rating_data(d)
    p = mfx.parameter(); p.k, p.lambda_ = K, 0.05
    H0 = mfx.initial_col(COLS, K)
    s = mfx.AlsSolver(host, mfx.test_data_of(host), p)
    s.set_factors(H0.copy())
    rep = s.iterate(1)
    W, H = s.get_factors()
    rep2 = s.iterate(1)
    s.close()
    assert rep2[0].rmse < rep[0].rmse < 1.5
    rng = np.random.default_rng(0)
    lam = 0.05

    def check(ptr, idx, val, X, Y, picks):
        worst = 0.0
        for c in picks:
            lo, hi = int(ptr[c]), int(ptr[c + 1])
            if hi == lo:
                assert np.all(Y[c] == 0)
                continue
            Xo = X[idx[lo:hi].astype(np.int64)].astype(np.float64)
            A = Xo.T @ Xo + lam * np.eye(K)
            y = np.linalg.solve(A, Xo.T @ val[lo:hi].astype(np.float64))
            worst = max(worst, float(np.max(np.abs(Y[c] - y)) / max(1e-12, np.max(np.abs(y)))))
        return worst

    rows = rng.choice(ROWS, 400, replace=False)
    cols = np.concatenate([rng.choice(COLS, 150, replace=False), [int(np.argmax(np.diff(host.csc_col_ptr.astype(np.int64))))]])
    assert check(host.csr_row_ptr, host.csr_col_idx, host.csr_val, H0, W, rows) < 2e-3      # W solved against H0
    assert check(host.csc_col_ptr, host.csc_row_idx, host.csc_val, W, H, cols) < 2e-3       # H solved against that W


def test_fullsize_follows_the_float64_trajectory(big):
    """The whole solve against the SAME algorithm in float64 (torch on the device, used as a calculator):
    test RMSE per outer iteration within 2e-5, factors within 1e-3 of scale after 2 outer iterations at
    k = 8.  (The fp32 CPU reference itself is 5e-5 ... 2e-4 away from float64 at this size: its sequential
    fp32 sums over columns with up to 10^6 ratings -- tools/rmse_fullsize_check.py)"""
    mfx, torch, d = big
    k, lam, iters = 8, 0.05, 2
    dev = torch.device("cuda:0")
    f8 = torch.float64
    W0 = mfx.initial_col(k, ROWS)
    p = mfx.parameter()
    p.k, p.lambda_ = k, lam
    s = mfx.CcdSolver(None, None, p, device_arrays=d)
    s.set_factors(W0)
    rep = s.iterate(iters)
    Wg, Hg = s.get_factors()
    s.close()

    rows = torch.repeat_interleave(torch.arange(ROWS, device=dev), (d["csr_row_ptr"][1:] - d["csr_row_ptr"][:-1]).long())
    cols = d["csr_col_idx"].long()
    r = d["csr_val"].to(f8)
    cnt_r, cnt_c = torch.bincount(rows, minlength=ROWS).to(f8), torch.bincount(cols, minlength=COLS).to(f8)
    W, H = torch.from_numpy(W0).to(dev).to(f8), torch.zeros((k, COLS), dtype=f8, device=dev)
    tr, tc, tv = d["test_row"].long(), d["test_col"].long(), d["test_val"].to(f8)
    rmse64 = []
    for it in range(iters):
        for t in range(k):
            u, v = W[t], H[t]
            if it > 0:
                r += u[rows] * v[cols]
            ur = u[rows]
            g, h = torch.bincount(cols, weights=ur * r, minlength=COLS), torch.bincount(cols, weights=ur * ur, minlength=COLS)
            v = torch.where(cnt_c > 0, g / (lam * cnt_c + h + (cnt_c == 0)), torch.zeros_like(g))
            vc = v[cols]
            g, h = torch.bincount(rows, weights=vc * r, minlength=ROWS), torch.bincount(rows, weights=vc * vc, minlength=ROWS)
            u = torch.where(cnt_r > 0, g / (lam * cnt_r + h + (cnt_r == 0)), torch.zeros_like(g))
            r -= u[rows] * vc
            W[t], H[t] = u, v
        rmse64.append(float(torch.sqrt((((W[:, tr] * H[:, tc]).sum(0) - tv) ** 2).mean())))
    rm = np.array([x.rmse for x in rep])
    assert np.all(np.abs(rm - np.array(rmse64)) < 2e-5), (rm, rmse64)
    scale = float(max(W.abs().max(), H.abs().max()))
    assert float((torch.from_numpy(Wg).to(dev).to(f8) - W).abs().max()) < 1e-3 * scale
    assert float((torch.from_numpy(Hg).to(dev).to(f8) - H).abs().max()) < 1e-3 * scale


# ----------------------------------------------------------------------------------------------------------------
# The headline config against the ORACLE (the bit-exact restatement of the reference's ccdr1_OMP / ALS_OMP), at the
# north star's own terms: k = 64, lambda = 0.05, per-iteration test RMSE -- in TWO modes:
#
#   (a) the reference-order parity mode (schedule 0, kernel_variant -1; csrc/ccd_reforder.hip): every column / row sum
#       accumulated strictly left to right in unfused fp32, as src/CCD.cpp:6-16 does.  Bar: W and H BIT-IDENTICAL to
#       the oracle's after three outer iterations at the full size (480 189 x 17 770, 99 M ratings, longest column
#       237 488), RMSE equal to 1e-9.  This is the demonstration that the GPU implements the reference's algorithm
#       exactly -- every multiply, the order of the two residual updates, the division, lambda * count.
#   (b) the product path (fused passes, tree-shaped / segmented-scan sums).  It differs from (a) ONLY in the order in
#       which the fp32 terms of a sum are added, and its distance to the oracle is therefore its distance to (a):
#       asserted equal to 1e-9.  Measured (profiles/r02_fullsize_oracle.txt, profiles/r03_fullsize_reforder.txt):
#         item popularity sigma 1.8 (longest column 237 488; the real Netflix figure is 232 944):
#             |RMSE_product - RMSE_reference| = 1.6e-6, 1.28e-4, 5.1e-6 over the three outer iterations
#             product vs the same algorithm in float64: 1.4e-8;  reference order vs float64: 1.28e-4  (the sequential
#             fp32 sums over 2e5-entry columns are the side that moves)
#         sigma 1.12 (longest column 95 875): 4e-7, 2.4e-5, 8e-8
#       So with (a) the north star's "test-RMSE within 1e-4 of the CPU reference" is met exactly (0), and the fast
#       path's summation-order deviation is bounded at 2e-4 for the Netflix-like skew (1e-4 at the milder one) --
#       DESIGN.md section 2.
@pytest.mark.parametrize("sigma_cols,rmse_tol,factor_tol", [(1.8, 2e-4, 5e-2), (1.12, 1e-4, 2e-2)])
def test_fullsize_k64_ccd_vs_oracle(sigma_cols, rmse_tol, factor_tol):
    import torch
    import mfx
    from mfx import synth_torch
    from oracle import oracle as orc
    dev = synth_torch.synth_ratings_device(ROWS, COLS, NNZ, seed=1234, device="cuda:0", sigma_cols=sigma_cols)
    d = synth_torch.to_rating_data(dev)
    lam, t = 0.05, 3
    W0 = mfx.initial_col(K, ROWS)
    out = {}
    for mode, (schedule, variant) in (("product", (1, 1)), ("reference_order", (0, -1))):
        p = mfx.parameter()
        p.k, p.lambda_, p.maxiter, p.schedule, p.kernel_variant = K, lam, t, schedule, variant
        s = mfx.CcdSolver(None, None, p, device_arrays=dev)
        s.set_factors(W0.copy())
        rep = s.iterate(t)
        W, H = s.get_factors()
        s.close()
        out[mode] = (W, H, np.array([r.rmse for r in rep]))
    del dev
    torch.cuda.empty_cache()
    Wr, Hr, rmse_ref, *_ = orc.ccdr1(d, W0, K, lam, t, 1, orc.max_threads())
    # (a) reference order: the oracle's bits
    Wx, Hx, rx = out["reference_order"]
    assert np.array_equal(Wx.view(np.uint32), Wr.view(np.uint32)), float(np.abs(Wx - Wr).max())
    assert np.array_equal(Hx.view(np.uint32), Hr.view(np.uint32)), float(np.abs(Hx - Hr).max())
    assert np.all(np.abs(rx - rmse_ref) < 1e-9), (rx, rmse_ref)
    # (b) product path: a summation-order deviation of the recorded size, nothing else
    W, H, rm = out["product"]
    assert np.all(np.abs(rm - rmse_ref) < rmse_tol), (rm, rmse_ref)
    assert np.all(np.abs(np.abs(rm - rx) - np.abs(rm - rmse_ref)) < 1e-9)
    assert rmse_ref[2] < rmse_ref[1] < rmse_ref[0] and rm[2] < rm[1] < rm[0]
    scale = float(max(np.abs(Wr).max(), np.abs(Hr).max()))
    assert np.abs(W - Wr).max() < factor_tol * scale and np.abs(H - Hr).max() < factor_tol * scale
    print("fullsize k=64 sigma_cols=%.2f |rmse_product - rmse_reference| = %s ; reference-order mode: bit-identical"
          % (sigma_cols, np.abs(rm - rmse_ref)))
    # (r4) the reference's OWN acceptance yardstick, golden_compare (src/extras.cpp:218-238: an entry fails when
    # |x - ref| > 0.1 |ref|), applied to product-path factors against the reference's: what `-CUDA -OMP` prints at this
    # size.  Reported, not asserted at zero: entries close to zero fail a relative 10 % bar on a 1e-4 absolute difference.
    for name, A, B in (("W", W, Wr), ("H", H, Hr)):
        bad = int(np.count_nonzero(np.abs(A.astype(np.float64) - B) > 0.1 * np.abs(B.astype(np.float64))))
        print("fullsize k=64 sigma_cols=%.2f golden_compare(product, reference) on %s: %s"
              % (sigma_cols, name, "Check... PASS!" if bad == 0 else "Check... NO PASS! [%.4f%%] #Error = %d out of %d entries." % (100.0 * bad / A.size, bad, A.size)))
        assert bad < 0.15 * A.size  # (round 2 counted ~6 % of W at the Netflix-like skew)


def test_fullsize_k64_als_vs_oracle(big, tmp_path):
    """BASELINE configs[3]: one full ALS iteration at the Netflix shape, k = 64, against oracle.als (measured:
    RMSE gap 5.8e-7, W within 5.9e-5 and H within 2.1e-4 of scale).  The oracle runs in a fresh process: inside
    this long-lived one the same call takes 240 s instead of 41 s (its per-row allocations in a heap that two
    hundred earlier tests have churned)."""
    import subprocess
    import sys
    mfx, torch, dev = big
    from mfx import synth_torch
    d = synth_torch.to_rating_data(dev)
    p = mfx.parameter()
    p.k, p.lambda_ = K, 0.05
    H0 = mfx.initial_col(COLS, K)
    s = mfx.AlsSolver(d, mfx.test_data_of(d), p)
    s.set_factors(H0.copy())
    rep = s.iterate(1)
    W, H = s.get_factors()
    s.close()
    names = ("csr_row_ptr", "csr_col_idx", "csr_val", "csc_col_ptr", "csc_row_idx", "csc_val", "test_row", "test_col", "test_val")
    for nm in names:
        np.save(tmp_path / (nm + ".npy"), getattr(d, nm))
    np.save(tmp_path / "H0.npy", H0)
    root = __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))
    code = (
        "import sys, numpy as np\n"
        f"sys.path.insert(0, {root!r}); sys.path.insert(0, {root!r} + '/cuda-recommender_amd')\n"
        "from mfx.dataset import RatingData\n"
        "from oracle import oracle as orc\n"
        f"t = {str(tmp_path)!r}\n"
        f"a = [np.load(t + '/' + n + '.npy') for n in {names!r}]\n"
        f"d = RatingData({ROWS}, {COLS}, a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8])\n"
        f"W, H, rmse, _ = orc.als(d, np.load(t + '/H0.npy'), {K}, 0.05, 1, orc.max_threads())\n"
        "np.save(t + '/Wr.npy', W); np.save(t + '/Hr.npy', H); np.save(t + '/rmse.npy', rmse)\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr
    Wr, Hr, rmse_ref = np.load(tmp_path / "Wr.npy"), np.load(tmp_path / "Hr.npy"), np.load(tmp_path / "rmse.npy")
    for f in tmp_path.iterdir():
        f.unlink()
    assert abs(rep[0].rmse - rmse_ref[0]) < 1e-4, (rep[0].rmse, rmse_ref)
    scale = float(max(np.abs(Wr).max(), np.abs(Hr).max()))
    assert np.abs(W - Wr).max() < 2e-3 * scale and np.abs(H - Hr).max() < 2e-3 * scale


# ---- the other BASELINE.json configurations, at their own shapes ----------------------------------------------

@pytest.mark.parametrize("rows,cols,nnz,k,t,alg", [
    (943, 1682, 100000, 10, 5, "ccd"),        # configs[0]: MovieLens-100K shape, k = 10
    (943, 1682, 100000, 10, 3, "als"),
    (6040, 3706, 1000209, 40, 5, "ccd"),      # configs[1]: MovieLens-1M shape, k = 40
    (6040, 3706, 1000209, 40, 2, "als"),
])
def test_movielens_shapes_vs_oracle(rows, cols, nnz, k, t, alg):
    """configs[0] / configs[1] of BASELINE.json (synthetic ratings of the MovieLens shapes -- the data sets
    themselves are not in the image): every outer iteration's test RMSE within 1e-4 of the oracle's, factors within
    2e-3 of scale, at the reference's own defaults (lambda 0.05, T = 1)."""
    import mfx
    from oracle import oracle as orc
    d = mfx.dataset.synth_ratings(rows, cols, nnz, seed=100 + k, skew=0.9, test_frac=0.05)
    p = mfx.parameter()
    p.k, p.lambda_, p.maxiter = k, 0.05, t
    if alg == "ccd":
        W0 = mfx.initial_col(k, rows)
        Wr, Hr, rmse_ref, *_ = orc.ccdr1(d, W0, k, 0.05, t, 1, 2)
        s = mfx.CcdSolver(d, mfx.test_data_of(d), p)
        s.set_factors(W0.copy())
    else:
        H0 = mfx.initial_col(cols, k)
        Wr, Hr, rmse_ref, _ = orc.als(d, H0, k, 0.05, t, 2)
        s = mfx.AlsSolver(d, mfx.test_data_of(d), p)
        s.set_factors(H0.copy())
    rep = s.iterate(t)
    W, H = s.get_factors()
    s.close()
    rm = np.array([r.rmse for r in rep])
    assert np.all(np.abs(rm - rmse_ref) < 1e-4), (rm, rmse_ref)
    scale = float(max(np.abs(Wr).max(), np.abs(Hr).max()))
    assert np.abs(W - Wr).max() < 2e-3 * scale and np.abs(H - Hr).max() < 2e-3 * scale


def test_config5_shard_vs_oracle():
    """configs[4] (10 M x 1 M, 1e9 ratings, k = 128, row-sharded over 8 GPUs): ONE rank's shard at full size --
    rows [0, 1.25 M) of the 10 M, all 1 M columns, 1.25e8 ratings (12.5 per column, 100 per row: the hyper-sparse
    regime, scatter layout on both sides) -- solved stand-alone (a one-rank job over its own rows) and compared
    with the oracle on the same arrays.  k = 8 and two outer iterations keep the oracle at a few seconds; the
    layouts, kernels and launch shapes are those of the k = 128 run (k only sets the number of rank-one
    passes).  Also: a second run reproduces the first bit for bit (64-bit fixed-point LDS accumulation)."""
    import torch
    import mfx
    from mfx import synth_torch
    from oracle import oracle as orc
    rows, cols, nnz, k, t = 1250000, 1000000, 125000000, 8, 2
    dev = synth_torch.synth_ratings_device(rows, cols, nnz, seed=5, device="cuda:0", sigma_rows=0.5, sigma_cols=1.0)
    d = synth_torch.to_rating_data(dev)
    W0 = mfx.initial_col(k, rows)
    p = mfx.parameter()
    p.k, p.lambda_, p.maxiter = k, 0.05, t
    outs = []
    for _ in range(2):
        s = mfx.CcdSolver(None, None, p, device_arrays=dev)
        info = s.layout_info()
        assert info["csc"]["kind"] == "scatter" and info["csr"]["kind"] == "scatter", info
        s.set_factors(W0.copy())
        rep = s.iterate(t)
        W, H = s.get_factors()
        csc, csr = s.get_residual(d.nnz)
        s.close()
        outs.append((W, H, csc, csr, np.array([r.rmse for r in rep])))
    # the reference-order parity mode on the same shard (plain layout, sequential sums): the oracle's bits
    p.schedule, p.kernel_variant = 0, -1
    s = mfx.CcdSolver(None, None, p, device_arrays=dev)
    s.set_factors(W0.copy())
    s.iterate(t)
    Wx, Hx = s.get_factors()
    s.close()
    del dev
    torch.cuda.empty_cache()
    (W, H, csc, csr, rm), second = outs
    assert all(np.array_equal(a.view(np.uint32), b.view(np.uint32)) for a, b in zip(outs[0][:4], second[:4]))
    Wr, Hr, rmse_ref, _, csc_ref, csr_ref = orc.ccdr1(d, W0, k, 0.05, t, 1, orc.max_threads())
    assert np.array_equal(Wx.view(np.uint32), Wr.view(np.uint32)) and np.array_equal(Hx.view(np.uint32), Hr.view(np.uint32))
    assert np.all(np.abs(rm - rmse_ref) < 1e-4), (rm, rmse_ref)
    scale = float(max(np.abs(Wr).max(), np.abs(Hr).max()))
    assert np.abs(W - Wr).max() < 2e-3 * scale and np.abs(H - Hr).max() < 2e-3 * scale
    assert np.abs(csc - csc_ref).max() < 1e-3 and np.abs(csr - csr_ref).max() < 1e-3


def test_config5_shard_k128_vs_reference_order():
    """configs[4] at its own rank: k = 128 on one rank's full-size shard (1.25 M x 1 M, 1.25e8 ratings), two outer
    iterations, product path (scatter layout, fixed-point LDS sums) against the REFERENCE-ORDER mode on the GPU -- which
    is the CPU reference bit for bit (test_config5_shard_vs_oracle pins that at this shape with k = 8; the oracle
    itself would need ~10 minutes of host time for 2 x 128 rank-one updates of 1.25e8 ratings).  Bar: the north star's
    1e-4 on the test RMSE of every outer iteration, factors within 2e-3 of scale."""
    import torch
    import mfx
    from mfx import synth_torch
    rows, cols, nnz, k, t = 1250000, 1000000, 125000000, 128, 2
    dev = synth_torch.synth_ratings_device(rows, cols, nnz, seed=5, device="cuda:0", sigma_rows=0.5, sigma_cols=1.0)
    W0 = mfx.initial_col(k, rows)
    out = {}
    for mode, (schedule, variant) in (("product", (1, 1)), ("reference_order", (0, -1))):
        p = mfx.parameter()
        p.k, p.lambda_, p.maxiter, p.schedule, p.kernel_variant = k, 0.05, t, schedule, variant
        s = mfx.CcdSolver(None, None, p, device_arrays=dev)
        if mode == "product":
            info = s.layout_info()
            assert info["csc"]["kind"] == "scatter" and info["csr"]["kind"] == "scatter", info
        s.set_factors(W0.copy())
        rep = s.iterate(t)
        out[mode] = s.get_factors() + (np.array([r.rmse for r in rep]),)
        s.close()
    del dev
    torch.cuda.empty_cache()
    (W, H, rm), (Wx, Hx, rx) = out["product"], out["reference_order"]
    assert np.all(np.abs(rm - rx) < 1e-4), (rm, rx)
    # (no monotonicity claim: 128 ranks on ~100 ratings per user overfit the planted rank-8 model, the test RMSE rises)
    scale = float(max(np.abs(Wx).max(), np.abs(Hx).max()))
    assert np.abs(W - Wx).max() < 2e-3 * scale and np.abs(H - Hx).max() < 2e-3 * scale
    print("config5 shard k=128: |rmse_product - rmse_reference_order| =", np.abs(rm - rx))


def test_config5_sharded_solve_at_the_real_message_size(monkeypatch):
    """configs[4]'s per-inner-iteration exchange at its real size: n = 1 000 000 item columns, so every rank's column
    partials are a 2 x 1 M fp32 = 8 MB all-reduce buffer -- scatter layout on both sides (hyper-sparse), slabs ->
    k_scatter_combine -> dense (g, h) -> all-reduce over the shards -> finalize from the reduced buffer with the GLOBAL
    |Omega_c| -> local u-pass.  Two nnz-balanced user-row shards (threads of this process on one GPU, loopback
    communicator: RCCL refuses two ranks on one device) of a 500 000 x 1 000 000 matrix with 5e7 ratings, against the
    UNSHARDED oracle on the whole matrix, k = 4, two outer iterations."""
    import threading
    import torch
    import mfx
    from mfx import synth_torch
    from oracle import oracle as orc
    monkeypatch.setenv("MFX_OVERLAP_GROUPS", "2")  # (r4) with the exchange of the first half of the columns under the second half's pass
    rows, cols, nnz, k, t, nshards = 500_000, 1_000_000, 50_000_000, 4, 2, 2
    dev = synth_torch.synth_ratings_device(rows, cols, nnz, seed=55, device="cuda:0", sigma_rows=0.5, sigma_cols=1.0)
    d = synth_torch.to_rating_data(dev)
    del dev
    torch.cuda.empty_cache()
    W0 = mfx.initial_col(k, rows)
    bounds = mfx.partition_rows(d, nshards)
    gcnt = np.ascontiguousarray(np.diff(d.csc_col_ptr.astype(np.int64)).astype(np.uint32))
    out, errs = [None] * nshards, []

    def run(r):
        try:
            lo, hi = int(bounds[r]), int(bounds[r + 1])
            shard = mfx.extract_shard(d, lo, hi)
            comm = mfx.Comm(None, r, nshards, 0, local_group=9050)
            p = mfx.parameter()
            p.k, p.lambda_, p.maxiter, p.profile = k, 0.05, t, 1
            s = mfx.CcdSolver(shard, mfx.test_data_of(shard), p, comm=comm, global_col_nnz=gcnt, global_test_nnz=d.nnz_test)
            info = s.layout_info()
            s.set_factors(np.ascontiguousarray(W0[:, lo:hi]))
            status = comm.agree(0)
            assert status == 0
            rep = s.iterate(t)
            out[r] = (s.get_factors(), [x.rmse for x in rep], info, s.kernel_times())
            s.close(); comm.close()
        except Exception as e:
            errs.append(e)
            raise

    th = [threading.Thread(target=run, args=(r,)) for r in range(nshards)]
    [x.start() for x in th]
    [x.join(timeout=600) for x in th]
    assert not errs and all(o is not None for o in out), errs
    for (_, _), _, info, times in out:
        assert info["csc"]["kind"] == "scatter" and info["csr"]["kind"] == "scatter", info
        assert times["rccl_allreduce"][1] >= 2 * k * t and "ccd_scatter_combine" in times and times["ccd_scatter_v_pass"][1] == 2 * k * t
    Wr, Hr, rmse_ref, *_ = orc.ccdr1(d, W0, k, 0.05, t, 1, orc.max_threads())
    W = np.concatenate([o[0][0] for o in out], axis=1)
    scale = float(max(np.abs(Wr).max(), np.abs(Hr).max()))
    assert np.abs(W - Wr).max() < 2e-3 * scale
    for (Wl, Hl), rm, _, _ in out:
        assert np.abs(Hl - Hr).max() < 2e-3 * scale
        assert np.all(np.abs(np.array(rm) - rmse_ref) < 1e-4), (rm, rmse_ref)
    assert np.array_equal(out[0][0][1].view(np.uint32), out[1][0][1].view(np.uint32))  # the H replicas agree bit for bit


def test_fullsize_fused_finalize_is_bit_identical(big, monkeypatch):
    """Opt-in path (MFX_FUSE_FINALIZE=1; off by default, it measured slower).  Netflix shape, k = 64, two outer iterations (256 fused passes, ~3 000 workgroups each, on all eight XCDs): the
    in-pass finalize reads other workgroups' partial sums of the same launch; one stale read would change bits."""
    mfx, torch, d = big
    outs = []
    for fuse in ("1", "0"):
        monkeypatch.setenv("MFX_FUSE_FINALIZE", fuse)
        s, rep, W, H = _solve(mfx, d, 2)
        s.close()
        outs.append((W, H, np.array([r.rmse for r in rep])))
    assert np.array_equal(outs[0][0].view(np.uint32), outs[1][0].view(np.uint32))
    assert np.array_equal(outs[0][1].view(np.uint32), outs[1][1].view(np.uint32))
    assert np.array_equal(outs[0][2], outs[1][2])
