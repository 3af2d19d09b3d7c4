"""Pins the CPU oracle (oracle/mf_oracle.cpp) to the reference.

tests/golden/*.npz were produced by the reference's own src/{CCD,ALS,tools,extras}.cpp
(compiled in place by oracle/Makefile, driven by oracle/make_fixtures.py).  The reference has
no tests or golden vectors of its own, so these fixtures ARE the pin: every
function of the restatement must reproduce them bit for bit, for any thread count.
"""
import numpy as np
import pytest

from conftest import bits
from oracle import oracle as orc


def test_initial_col_matches_reference_init(golden):
    name, g, d = golden
    k = int(g["k"][0])
    # CCD layout: initial_col(W, k, rows); ALS layout: initial_col(W, rows, k)  (src/main.cpp:86-98)
    assert np.array_equal(bits(orc.initial_col(k, d.rows)), bits(g["ccd_T1__W0"]))
    assert np.array_equal(bits(orc.initial_col(k, d.cols)), bits(g["ccd_T1__H0"]))
    assert np.array_equal(bits(orc.initial_col(d.cols, k)), bits(g["als__H0"]))


@pytest.mark.parametrize("threads", [1, 3])
def test_single_steps_bit_exact(golden, threads):
    name, g, d = golden
    lam = float(g["lam"][0])
    W0 = g["ccd_T1__W0"]; H0 = g["ccd_T1__H0"]; k = int(g["k"][0])
    v1 = orc.rank_one_sweep(d.csc_col_ptr, d.csc_row_idx, d.csc_val, W0[0].copy(), lam, threads)
    assert np.array_equal(bits(v1), bits(g["step_v1"]))
    u1 = orc.rank_one_sweep(d.csr_row_ptr, d.csr_col_idx, d.csr_val, v1, lam, threads)
    assert np.array_equal(bits(u1), bits(g["step_u1"]))
    csc, csr = d.csc_val.copy(), d.csr_val.copy()
    orc.update_rating(d.csc_col_ptr, d.csc_row_idx, csc, u1, v1, False, threads)
    orc.update_rating(d.csr_row_ptr, d.csr_col_idx, csr, v1, u1, False, threads)
    assert np.array_equal(bits(csc), bits(g["step_csc_sub"]))
    assert np.array_equal(bits(csr), bits(g["step_csr_sub"]))
    t1 = 1 % k
    orc.update_rating(d.csc_col_ptr, d.csc_row_idx, csc, W0[t1].copy(), H0[t1].copy(), True, threads)
    orc.update_rating(d.csr_row_ptr, d.csr_col_idx, csr, H0[t1].copy(), W0[t1].copy(), True, threads)
    assert np.array_equal(bits(csc), bits(g["step_csc_add"]))
    assert np.array_equal(bits(csr), bits(g["step_csr_add"]))
    assert orc.calrmse(d, np.ascontiguousarray(W0), np.ascontiguousarray(H0), k, False) == float(g["step_rmse_init_ccd"][0])


@pytest.mark.parametrize("tag", ["ccd_T1", "ccd_T3"])
@pytest.mark.parametrize("threads", [1, 4])
def test_ccdr1_bit_exact(golden, tag, threads):
    name, g, d = golden
    k, lam = int(g["k"][0]), float(g["lam"][0])
    t, T = int(g[tag + "__maxiter"][0]), int(g[tag + "__maxinner"][0])
    W, H, rmse, _, csc, csr = orc.ccdr1(d, g[tag + "__W0"], k, lam, t, T, threads)
    assert np.array_equal(bits(W), bits(g[tag + "__W"]))
    assert np.array_equal(bits(H), bits(g[tag + "__H"]))
    assert np.array_equal(bits(csc), bits(g[tag + "__csc_val_final"]))
    assert np.array_equal(bits(csr), bits(g[tag + "__csr_val_final"]))
    # the reference prints RMSE with %lf (6 decimals): src/CCD.cpp:158-159
    assert np.allclose(rmse, g[tag + "__rmse"], rtol=0, atol=5.1e-7)
    assert abs(rmse[-1] - float(g[tag + "__final_rmse"][0])) <= 5.1e-7


def test_gramian_and_inverse_bit_exact(golden):
    name, g, d = golden
    k, lam = int(g["k"][0]), float(g["lam"][0])
    row = int(g["step_als_row"][0])
    lo, hi = int(d.csr_row_ptr[row]), int(d.csr_row_ptr[row + 1])
    H0 = np.ascontiguousarray(g["als__H0"])
    A = orc.gramian(np.ascontiguousarray(d.csr_col_idx[lo:hi]), H0, k)
    assert np.array_equal(bits(A), bits(g["step_gram"]))
    A[np.arange(k), np.arange(k)] = A[np.arange(k), np.arange(k)] + np.float32(lam)
    assert np.array_equal(bits(orc.chol_inverse(A)), bits(g["step_inv"]))


@pytest.mark.parametrize("threads", [1, 4])
def test_als_bit_exact(golden, threads):
    name, g, d = golden
    k, lam = int(g["k"][0]), float(g["lam"][0])
    t = int(g["als__maxiter"][0])
    W, H, rmse, _ = orc.als(d, g["als__H0"], k, lam, t, threads)
    assert np.array_equal(bits(W), bits(g["als__W"]))
    assert np.array_equal(bits(H), bits(g["als__H"]))
    assert np.allclose(rmse, g["als__rmse"], rtol=0, atol=5.1e-7)


def test_als_half_equals_full_first_half(golden):
    name, g, d = golden
    k, lam = int(g["k"][0]), float(g["lam"][0])
    H0 = np.ascontiguousarray(g["als__H0"])
    W1 = orc.als_half(d.csr_row_ptr, d.csr_col_idx, d.csr_val, H0, k, lam, 2)
    W, _, _, _ = orc.als(d, H0, k, lam, 1, 2)
    H1 = orc.als_half(d.csc_col_ptr, d.csc_row_idx, d.csc_val, W1, k, lam, 2)
    _, H, _, _ = orc.als(d, H0, k, lam, 1, 2)
    assert np.array_equal(bits(W1), bits(W)) and np.array_equal(bits(H1), bits(H))
    empty = np.diff(d.csr_row_ptr.astype(np.int64)) == 0
    assert np.all(W1[empty] == 0)  # zero-row rule, src/ALS.cpp:151-157


@pytest.mark.parametrize("tag", ["ccd_T1", "ccd_T3"])
def test_ccdr1_ext_with_everything_off_is_ccdr1_and_its_rank_trace_ends_on_the_iteration_rmse(golden, tag):
    """orc_ccdr1_ext (the flags the reference parses and ignores, given their LIBPMF meaning) with do_nmf = 0 and
    eps = 0 must be the pinned ccdr1 bit for bit; and its per-rank calrmse_r1 trace (src/tools.cpp:261-270) must end
    every outer iteration on the reference's own per-iteration test RMSE -- the incrementally updated test residual
    IS the residual calrmse recomputes from the factors (fp32 vs f64 accumulation apart)."""
    name, g, d = golden
    k, lam = int(g["k"][0]), float(g["lam"][0])
    t, T = int(g[tag + "__maxiter"][0]), int(g[tag + "__maxinner"][0])
    W, H, rmse, trace, done = orc.ccdr1_ext(d, g[tag + "__W0"], k, lam, t, T, 2)
    assert np.array_equal(bits(W), bits(g[tag + "__W"])) and np.array_equal(bits(H), bits(g[tag + "__H"]))
    assert np.allclose(rmse, g[tag + "__rmse"], rtol=0, atol=5.1e-7) and list(done) == [k] * t  # (the log prints 6 decimals)
    if d.nnz_test:
        assert np.all(np.abs(trace[:, -1] - g[tag + "__rmse"]) < 3e-6), (trace[:, -1], g[tag + "__rmse"])
