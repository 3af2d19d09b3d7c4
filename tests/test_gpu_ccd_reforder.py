"""CCD++ in the reference's own summation order (mfx_params.schedule = 0, kernel_variant = -1;
csrc/ccd_reforder.hip): every sum of RankOneUpdate_Original_float (src/CCD.cpp:6-16) is accumulated strictly
left to right in unfused fp32 from (0, lambda * count), so the GPU result must equal the CPU reference
BIT FOR BIT -- factors and both residual copies -- not merely within a tolerance.  The bar here is
np.array_equal on the uint32 views; only the test RMSE (fp64 sum over the test set, whose order is not
part of the reference's result) is compared at 1e-12 against the oracle, and to the six decimals the reference
printed where the fixture holds its log line.
"""
import numpy as np
import pytest

from conftest import CASES, bits, load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mfx():
    import mfx as m
    assert m.device_count() >= 1, "no HIP device: these tests must run on the GPU box"
    return m


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle
    return oracle


def _params(mfx, k, lam, t, T):
    p = mfx.parameter()
    p.k, p.lambda_, p.maxiter, p.maxinneriter = k, lam, t, T
    p.schedule, p.kernel_variant = 0, -1
    return p


@pytest.mark.parametrize("name", CASES)
def test_rank_one_sweep_reference_order_golden_bit_exact(mfx, name):
    """One v-sweep and one u-sweep against the vectors the reference itself produced (step_v1 / step_u1)."""
    g, d = load_golden(name)
    lam = float(g["lam"][0])
    v1 = mfx.rank_one_sweep(d.csc_col_ptr, d.csc_row_idx, d.csc_val, g["ccd_T1__W0"][0].copy(), lam, -1)
    assert np.array_equal(bits(v1), bits(g["step_v1"]))
    u1 = mfx.rank_one_sweep(d.csr_row_ptr, d.csr_col_idx, d.csr_val, g["step_v1"].copy(), lam, -1)
    assert np.array_equal(bits(u1), bits(g["step_u1"]))


def test_rank_one_sweep_reference_order_long_and_degenerate_segments(mfx, orc):
    """Segments of every length class of the kernel: empty, 1 ... 63 (partial tile), exactly 64 / 256 (tile / stage
    boundaries), 257, a few thousand, 20 011 and 70 001 entries (many pipeline stages, carries between all of them),
    with signed operands and values (cancellation makes the order visible in the low bits)."""
    rng = np.random.default_rng(11)
    lens = np.concatenate([[0, 0, 20011, 0, 64, 256, 257, 255, 63, 65, 128, 1, 0, 70001, 512, 511, 513, 4099],
                           np.ones(300, np.int64), rng.integers(0, 200, 500), [7, 0]])
    ptr = np.zeros(lens.size + 1, np.uint32); ptr[1:] = np.cumsum(lens)
    nnz, nvec = int(ptr[-1]), 5000
    idx = rng.integers(0, nvec, nnz).astype(np.uint32)
    val = rng.uniform(-5, 5, nnz).astype(np.float32)
    vec = rng.uniform(-1, 1, nvec).astype(np.float32)
    for lam in (0.1, 0.0):
        ref = orc.rank_one_sweep(ptr, idx, val, vec, lam, 2)
        out = mfx.rank_one_sweep(ptr, idx, val, vec, lam, -1)
        nonempty = lens > 0
        assert np.array_equal(bits(out[nonempty]), bits(ref[nonempty])), lam
        assert np.all(out[~nonempty] == 0)


@pytest.mark.parametrize("tag", ["ccd_T1", "ccd_T3"])
@pytest.mark.parametrize("name", CASES)
def test_ccdpp_reference_order_golden_bit_exact(mfx, name, tag):
    """Whole solves against the reference's own ccdr1_OMP output: W, H and the two mutated residual arrays."""
    g, d = load_golden(name)
    k, lam = int(g["k"][0]), float(g["lam"][0])
    t, T = int(g[tag + "__maxiter"][0]), int(g[tag + "__maxinner"][0])
    p = _params(mfx, k, lam, t, T)
    p.profile = 1  # keep the per-kernel totals for kernel_times()
    s = mfx.CcdSolver(d, mfx.test_data_of(d), p)
    s.set_factors(np.array(g[tag + "__W0"], np.float32, copy=True))
    rep = s.iterate(t)
    W, H = s.get_factors()
    csc, csr = s.get_residual(d.nnz)
    times = s.kernel_times()
    s.close()
    assert np.array_equal(bits(W), bits(g[tag + "__W"])) and np.array_equal(bits(H), bits(g[tag + "__H"]))
    if tag == "ccd_T1":
        assert np.array_equal(bits(csc), bits(g["ccd_T1__csc_val_final"]))
        assert np.array_equal(bits(csr), bits(g["ccd_T1__csr_val_final"]))
    rmse = np.array([r.rmse for r in rep])
    assert np.all(np.abs(rmse - g[tag + "__rmse"]) < 6e-7), (rmse, g[tag + "__rmse"])  # the fixture holds the log's six decimals
    assert mfx.golden_compare(W, g[tag + "__W"], k, d.rows, quiet=True) == 0
    assert "ccd_ref_order_sweep" in times  # the mode ran its own kernel, not a neighbour's


@pytest.mark.parametrize("T", [1, 2])
def test_ccdpp_reference_order_ml1m_shape_vs_oracle_bit_exact(mfx, orc, T):
    """BASELINE configs[1] shape (6040 x 3706, 1e6 ratings, k = 40) with empty rows / columns: bit-identical to
    oracle.ccdr1 after three outer iterations, residual copies included."""
    d = mfx.dataset.synth_ratings(6040, 3706, 1_000_000, seed=7, skew=0.9, test_frac=0.01,
                                  empty_row_frac=0.01, empty_col_frac=0.02)
    k, lam, t = 40, 0.05, 3
    W0 = mfx.initial_col(k, d.rows)
    Wr, Hr, rmse_ref, _, csc_ref, csr_ref = orc.ccdr1(d, W0, k, lam, t, T, orc.max_threads())
    s = mfx.CcdSolver(d, mfx.test_data_of(d), _params(mfx, k, lam, t, T))
    s.set_factors(W0.copy())
    rep = s.iterate(t)
    W, H = s.get_factors()
    csc, csr = s.get_residual(d.nnz)
    s.close()
    assert np.array_equal(bits(W), bits(Wr)) and np.array_equal(bits(H), bits(Hr))
    assert np.array_equal(bits(csc), bits(csc_ref)) and np.array_equal(bits(csr), bits(csr_ref))
    assert np.all(np.abs(np.array([r.rmse for r in rep]) - rmse_ref) < 1e-12)


@pytest.mark.parametrize("rows,cols,nnz", [(1, 1, 1), (1, 40, 17), (50, 1, 23), (3, 2, 5), (7, 5, 0)])
def test_reference_order_degenerate_shapes_bit_exact(mfx, orc, rows, cols, nnz):
    """One row, one column, a handful of ratings, no ratings at all: still the oracle's bits (empty segments -> exactly 0)."""
    rng = np.random.default_rng(rows * 100 + cols)
    if nnz:
        cells = rng.choice(rows * cols, nnz, replace=False)
        d = mfx.dataset.from_coo(rows, cols, cells // cols, cells % cols, rng.uniform(1, 5, nnz).astype(np.float32),
                                 [0], [0], np.array([3.0], np.float32))
    else:
        d = mfx.dataset.from_coo(rows, cols, [], [], np.zeros(0, np.float32), [1], [2], np.array([3.0], np.float32))
    k, t = 3, 3
    W0 = mfx.initial_col(k, d.rows)
    Wr, Hr, rmse_ref, _, csc_ref, csr_ref = orc.ccdr1(d, W0, k, 0.05, t, 2, 2)
    s = mfx.CcdSolver(d, mfx.test_data_of(d), _params(mfx, k, 0.05, t, 2))
    s.set_factors(W0.copy())
    rep = s.iterate(t)
    W, H = s.get_factors()
    csc, csr = s.get_residual(d.nnz)
    s.close()
    assert np.array_equal(bits(W), bits(Wr)) and np.array_equal(bits(H), bits(Hr))
    assert np.array_equal(bits(csc), bits(csc_ref)) and np.array_equal(bits(csr), bits(csr_ref))
    assert np.all(np.abs(np.array([r.rmse for r in rep]) - rmse_ref) < 1e-12)


def test_reference_order_follows_the_stored_order_of_unsorted_segments(mfx, orc):
    """The reference adds a column's terms in the order the file holds them (its loader does not sort): with the
    entries of every row / column shuffled the sums change in the low bits -- and the parity mode must follow the
    shuffled order, not some canonical one: bit-identical to the oracle on the SAME arrays, and different from the
    solve on the sorted arrays."""
    d0 = mfx.dataset.synth_ratings(700, 500, 30000, seed=77, skew=0.7, test_frac=0.02)
    d = d0.copy()
    rng = np.random.default_rng(5)
    for ptr, idx, val in ((d.csr_row_ptr, d.csr_col_idx, d.csr_val), (d.csc_col_ptr, d.csc_row_idx, d.csc_val)):
        for sgm in range(ptr.shape[0] - 1):
            lo, hi = int(ptr[sgm]), int(ptr[sgm + 1])
            if hi - lo > 1:
                o = rng.permutation(hi - lo)
                idx[lo:hi] = idx[lo:hi][o]
                val[lo:hi] = val[lo:hi][o]
    k, t = 5, 3
    W0 = mfx.initial_col(k, d.rows)
    out = []
    for data in (d, d0):
        Wr, Hr, *_ = orc.ccdr1(data, W0, k, 0.05, t, 1, 2)
        s = mfx.CcdSolver(data, mfx.test_data_of(data), _params(mfx, k, 0.05, t, 1))
        s.set_factors(W0.copy())
        s.iterate(t)
        W, H = s.get_factors()
        s.close()
        assert np.array_equal(bits(W), bits(Wr)) and np.array_equal(bits(H), bits(Hr))
        out.append(W)
    assert not np.array_equal(bits(out[0]), bits(out[1]))  # the order is visible in the bits


def test_reference_order_mode_argument_checks(mfx):
    d = mfx.dataset.synth_ratings(300, 200, 5000, seed=1, test_frac=0.02)
    p = _params(mfx, 4, 0.05, 1, 1)
    p.schedule = 1  # the fused schedule has no reference order
    with pytest.raises(mfx.MfxError, match="schedule = 0"):
        mfx.CcdSolver(d, mfx.test_data_of(d), p)
    p = _params(mfx, 4, 0.05, 1, 1)
    p.kernel_variant = -2
    with pytest.raises(mfx.MfxError, match="kernel_variant"):
        mfx.CcdSolver(d, mfx.test_data_of(d), p)


# ---- (r4) the mode's two forms: the owner passes on the default path's schedule (k_ref_quad / k_ref_split: subtraction, add-back,
# first sweep and division in one launch per copy) and the as-written sequence (MFX_REF_FUSED=0: one launch per reference kernel)

def _solve(mfx, d, W0, k, lam, t, T):
    s = mfx.CcdSolver(d, mfx.test_data_of(d), _params(mfx, k, lam, t, T))
    s.set_factors(W0.copy())
    rep = s.iterate(t)
    W, H = s.get_factors()
    csc, csr = s.get_residual(d.nnz)
    s.close()
    return W, H, csc, csr, np.array([r.rmse for r in rep])


@pytest.mark.parametrize("T", [1, 3])
def test_reference_order_owner_passes_equal_the_as_written_sequence(mfx, orc, monkeypatch, T):
    """Same bits from both forms and from the oracle, on a matrix whose columns AND rows reach the split kernel
    (segments of >= 4096 entries on both sides, next to thousands of short ones and empty ones)."""
    d = mfx.dataset.synth_ratings(9000, 6000, 1_500_000, seed=21, skew=1.3, test_frac=0.01, empty_row_frac=0.01, empty_col_frac=0.01)
    assert np.diff(d.csc_col_ptr).max() >= 4096 and np.diff(d.csr_row_ptr).max() >= 4096
    k, lam, t = 6, 0.05, 2
    W0 = mfx.initial_col(k, d.rows)
    Wr, Hr, rmse_ref, _, csc_ref, csr_ref = orc.ccdr1(d, W0, k, lam, t, T, orc.max_threads())
    out = {}
    for form in ("1", "0", "split", "table"):
        monkeypatch.setenv("MFX_REF_FUSED", "0" if form == "0" else "1")
        if form == "split":  # (by default a side takes the split kernel only when one of its segments has >= 8192 entries)
            monkeypatch.setenv("MFX_REF_LONG", "4096")
        if form == "table":  # k_ref_quad's LDS table holding a part of the operands only (as at the Netflix shape), and no split kernel
            monkeypatch.setenv("MFX_REF_LONG", "100000000")
            monkeypatch.setenv("MFX_REF_QUAD_TAB", "30000")
        out[form] = _solve(mfx, d, W0, k, lam, t, T)
        monkeypatch.delenv("MFX_REF_LONG", raising=False)
        monkeypatch.delenv("MFX_REF_QUAD_TAB", raising=False)
    for form, (W, H, csc, csr, rmse) in out.items():
        assert np.array_equal(bits(W), bits(Wr)) and np.array_equal(bits(H), bits(Hr)), form
        assert np.array_equal(bits(csc), bits(csc_ref)) and np.array_equal(bits(csr), bits(csr_ref)), form
        assert np.all(np.abs(rmse - rmse_ref) < 1e-12), form


def test_reference_order_sweep_op_both_forms(mfx, orc, monkeypatch):
    """mfx_rank_one_sweep(variant = -1) through k_ref_quad / k_ref_split and through k_sweep_ref / k_sweep_ref2: the oracle's bits,
    with segments on both sides of both thresholds (4096 and 32768 entries) and four-segment items of mixed lengths."""
    rng = np.random.default_rng(5)
    lens = np.concatenate([[40000, 0, 4096, 4095, 4097, 33000, 1, 2, 3, 4, 5, 63, 64, 65, 8191], rng.integers(0, 400, 900), [12000, 0, 0, 7]])
    ptr = np.zeros(lens.size + 1, np.uint32); ptr[1:] = np.cumsum(lens)
    nnz, nvec = int(ptr[-1]), 7000
    idx = rng.integers(0, nvec, nnz).astype(np.uint32)
    val = rng.uniform(-5, 5, nnz).astype(np.float32)
    vec = rng.uniform(-1, 1, nvec).astype(np.float32)
    ref = orc.rank_one_sweep(ptr, idx, val, vec, 0.07, 2)
    for form in ("1", "0"):  # (the 40 000-entry segment puts this side's long segments on the split kernel)
        monkeypatch.setenv("MFX_REF_FUSED", form)
        out = mfx.rank_one_sweep(ptr, idx, val, vec, 0.07, -1)
        assert np.array_equal(bits(out[lens > 0]), bits(ref[lens > 0])), form
        assert np.all(out[lens == 0] == 0), form
