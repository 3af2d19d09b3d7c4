"""GPU edge cases through the C ABI: degenerate shapes and parameters the reference's code paths
imply (empty segments -> 0, src/CCD.cpp:8; zero rows in ALS, src/ALS.cpp:151-157) plus the
limits of this implementation's layouts (panel boundaries, 16-bit local indices, rank windows)."""
import numpy as np
import pytest

from conftest import bits

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mfx():
    import mfx as m
    assert m.device_count() >= 1
    return m


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle
    return oracle


def _p(mfx, k, t=2, T=1, **kw):
    p = mfx.parameter()
    p.k, p.lambda_, p.maxiter, p.maxinneriter = k, 0.05, t, T
    for a, b in kw.items():
        setattr(p, a, b)
    return p


def _check(mfx, orc, d, k, t=2, T=1, tol=2e-3, **kw):
    W0 = mfx.initial_col(k, d.rows)
    Wr, Hr, rmse_ref, *_ = orc.ccdr1(d, W0, k, 0.05, t, T, 2)
    s = mfx.CcdSolver(d, mfx.test_data_of(d), _p(mfx, k, t, T, **kw))
    s.set_factors(W0.copy())
    rep = s.iterate(t)
    W, H = s.get_factors()
    csc, csr = s.get_residual(d.nnz)
    s.close()
    scale = max(1e-6, float(np.abs(Wr).max()), float(np.abs(Hr).max()))
    assert np.max(np.abs(W - Wr)) < tol * scale and np.max(np.abs(H - Hr)) < tol * scale
    if d.nnz_test:
        assert np.all(np.abs(np.array([r.rmse for r in rep]) - rmse_ref) < 1e-4)
    return W, H, csc, csr


def test_all_empty_matrix(mfx, orc):
    d = mfx.dataset.from_coo(7, 5, [], [], np.zeros(0, np.float32), [1], [2], np.array([3.0], np.float32))
    W, H, csc, csr = _check(mfx, orc, d, 3)
    assert np.all(W == 0) and np.all(H == 0)  # every row/column is empty -> exactly 0
    W, H, csc, csr = _check(mfx, orc, d, 3, kernel_variant=2)  # scatter layout: one all-padding chunk
    assert np.all(W == 0) and np.all(H == 0)
    Y = mfx.als_half(d.csr_row_ptr, d.csr_col_idx, d.csr_val, np.ones((5, 3), np.float32), 3, 0.1)
    assert np.all(Y == 0)


@pytest.mark.parametrize("rows,cols,nnz", [(1, 1, 1), (1, 40, 17), (50, 1, 23), (3, 2, 5)])
def test_tiny_shapes(mfx, orc, rows, cols, nnz):
    rng = np.random.default_rng(rows * 100 + cols)
    key = rng.choice(rows * cols, size=nnz, replace=False)
    d = mfx.dataset.from_coo(rows, cols, key // cols, key % cols, rng.uniform(1, 5, nnz).astype(np.float32),
                             [0], [0], np.array([2.5], np.float32))
    for kw in ({}, {"panel_rows": -1}, {"panel_rows": 16}, {"panel_rows": -16}, {"schedule": 0, "kernel_variant": 0},
               {"panel_rows": 16, "layout_build": 1}, {"kernel_variant": 2}, {"kernel_variant": 2, "panel_rows": 3, "schedule": 0}):
        _check(mfx, orc, d, 2, **kw)


@pytest.mark.parametrize("k", [1, 7, 130])
def test_rank_extremes(mfx, orc, k):
    d = mfx.dataset.synth_ratings(300, 120, 6000, seed=k, skew=0.9, test_frac=0.02)
    _check(mfx, orc, d, k, t=2)


def test_no_test_set_and_T_gt_1(mfx, orc):
    d = mfx.dataset.synth_ratings(400, 150, 9000, seed=3, skew=1.0, test_frac=0.0)
    assert d.nnz_test == 0
    W0 = mfx.initial_col(4, d.rows)
    s = mfx.CcdSolver(d, None, _p(mfx, 4, 2, 3))
    s.set_factors(W0.copy())
    rep = s.iterate(2)
    W, H = s.get_factors()
    s.close()
    assert all(r.rmse == 0.0 for r in rep)
    Wr, Hr, *_ = orc.ccdr1(d, W0, 4, 0.05, 2, 3, 2)
    assert np.max(np.abs(W - Wr)) < 2e-3 * np.abs(Wr).max()


@pytest.mark.parametrize("panel_rows", [999, 1000, 1001, 65535, -999, -1000, -1001])
def test_panel_boundaries(mfx, orc, panel_rows):
    """Gathered dimension exactly at / one past a panel boundary; the largest 16-bit panel."""
    d = mfx.dataset.synth_ratings(1000, 1000, 40000, seed=9, skew=0.6, test_frac=0.01)
    _check(mfx, orc, d, 3, panel_rows=panel_rows)


def test_many_one_entry_segments_overflow_the_lds_rank_window(mfx, orc):
    """Every row holds one rating: a workgroup chunk touches tens of thousands of ranks, far more
    than the 1024-entry LDS window -> the bounds-checked kernel instantiation with its global
    fallback must be selected and agree with the oracle."""
    rng = np.random.default_rng(4)
    rows, cols = 120000, 64
    r = np.arange(rows)
    c = rng.integers(0, cols, rows)
    d = mfx.dataset.from_coo(rows, cols, r, c, rng.uniform(1, 5, rows).astype(np.float32),
                             r[:100], c[:100], np.full(100, 3.0, np.float32))
    for kw in ({"panel_rows": 0}, {"panel_rows": 32}, {"panel_rows": -1}, {"panel_rows": -32},
               {"panel_rows": 32, "layout_build": 1}):
        _check(mfx, orc, d, 2, **kw)


def test_hyper_sparse_shard_layouts(mfx, orc):
    """600 k x 40 k with 4.2 M ratings: LDS-sized panels would leave ~1 entry per (panel, segment) pair.
    By default such a shape takes the SCATTER layout on both sides (ccd_scatter.hip); with the host builder
    (layout_build = 1) the round-1 layouts remain: 2 MB cache panels on the column side (global indices, L2
    gather), plain on the row side -- whose gathered vector is below 2 MB.  Results as the oracle's either way,
    fused and as-written, T = 1 and T = 2, and bitwise reproducible."""
    d = mfx.dataset.synth_ratings(600000, 40000, 4200000, seed=21, skew=0.3, test_frac=0.002)
    s = mfx.CcdSolver(d, mfx.test_data_of(d), _p(mfx, 2))
    info = s.layout_info()
    s.close()
    assert info["csc"]["kind"] == "scatter" and info["csr"]["kind"] == "scatter", info
    # (r4) phase alignment: at least CUs / 8 LDS-sized panels -> the count is rounded up to a multiple of CUs / 8
    # (600 k rows: 89 panels of 6816 -> 96 of 6250); fewer panels than that keep their count, in equal sizes (40 k columns: 6 x 6667)
    assert info["csc"]["panels"] == 96 and info["csc"]["panel_rows"] == 6250 and info["csr"]["panels"] == 6 and info["csr"]["panel_rows"] == 6667, info
    s = mfx.CcdSolver(d, mfx.test_data_of(d), _p(mfx, 2, layout_build=1))
    info = s.layout_info()
    s.close()
    assert info["csc"]["kind"] == "cache" and info["csc"]["panels"] == 3 and info["csc"]["panel_rows"] == 262144, info
    assert info["csr"]["kind"] == "plain", info
    a = _check(mfx, orc, d, 2, t=2)
    b = _check(mfx, orc, d, 2, t=2)
    assert all(np.array_equal(bits(x), bits(y)) for x, y in zip(a, b))  # 64-bit fixed-point accumulation: order-free
    _check(mfx, orc, d, 2, t=2, layout_build=1)
    _check(mfx, orc, d, 2, t=2, T=2)
    _check(mfx, orc, d, 2, t=2, schedule=0, kernel_variant=2)
    _check(mfx, orc, d, 2, t=2, panel_rows=-262144)   # cache panels forced, device-built


def test_small_hyper_sparse_matrices_stay_on_lds_panels(mfx, orc):
    """300 k x 20 k with 5 M ratings: 3.3 / 6 entries per (panel, segment) pair -- below the 8 at which big matrices
    switch to the scatter layout -- but only 1.5 M / 0.84 M pairs: k_finalize stays cheap and the LDS panels are the
    faster layout (choose_layout, ccd_solver.hip).  Oracle results either way."""
    d = mfx.dataset.synth_ratings(300000, 20000, 5000000, seed=33, skew=0.5, test_frac=0.002)
    assert _kinds(mfx, d) == ("lds", "lds")
    _check(mfx, orc, d, 2, t=2)
    _check(mfx, orc, d, 2, t=2, kernel_variant=2)  # the scatter layout, forced, on the same data


def _kinds(mfx, d, **kw):
    s = mfx.CcdSolver(d, None, _p(mfx, 2, **kw))
    info = s.layout_info()
    s.close()
    return info["csc"]["kind"], info["csr"]["kind"]


def test_scatter_segment_ids_as_byte_steps(mfx, orc):
    """The scatter layout stores the id of the streamed dimension as one byte per entry (the step from the
    previous entry of the tile's sorted order) plus one base per tile, decoded in the kernel by two packed prefix
    scans; kernel_variant = 3 keeps the explicit 32-bit ids.  Same ids, so bit-identical factors and residuals.
    A pattern with a run of > 255 rows without an entry falls back to the 32-bit ids on the side it concerns
    (and only there), still with the oracle's results."""
    d = mfx.dataset.synth_ratings(3000, 900, 60000, seed=8, skew=0.6, test_frac=0.02, empty_row_frac=0.02)
    assert _kinds(mfx, d, kernel_variant=2) == ("scatter", "scatter") and _kinds(mfx, d, kernel_variant=3) == ("scatter32", "scatter32")
    for kw in ({}, {"panel_rows": 150, "tiles_per_span": 2}, {"maxinneriter": 2}, {"schedule": 0}):
        T = kw.pop("maxinneriter", 1)
        a = _check(mfx, orc, d, 3, t=2, T=T, kernel_variant=2, **kw)
        b = _check(mfx, orc, d, 3, t=2, T=T, kernel_variant=3, **kw)
        assert all(np.array_equal(bits(x), bits(y)) for x, y in zip(a, b)), kw
    # rows 400 .. 1399 empty: the row ids step by 1000 inside the panels of the row-major copy
    rng = np.random.default_rng(3)
    r = np.concatenate([rng.integers(0, 400, 9000), rng.integers(1400, 2000, 9000)])
    c = rng.integers(0, 300, 18000)
    key = np.unique(r * 300 + c)
    r, c = key // 300, key % 300
    v = rng.uniform(1, 5, r.shape[0]).astype(np.float32)
    g = mfx.dataset.from_coo(2000, 300, r[50:], c[50:], v[50:], r[:50], c[:50], v[:50])
    kinds = _kinds(mfx, g, kernel_variant=2)
    assert kinds == ("scatter", "scatter32"), kinds   # the column-major copy streams column ids: dense
    _check(mfx, orc, g, 3, t=2, kernel_variant=2)
    _check(mfx, orc, g, 3, t=2, kernel_variant=2, panel_rows=64)


@pytest.mark.parametrize("kw", [{}, {"schedule": 0}, {"maxinneriter": 3}, {"panel_rows": 40}, {"panel_rows": 7, "tiles_per_span": 2}])
@pytest.mark.parametrize("name", ["tiny", "small", "edge"])
def test_scatter_layout_on_golden(mfx, name, kw):
    """kernel_variant = 2 forces the scatter layout on the reference-generated fixtures (empty rows / columns, a
    full row, a full column, 1-entry segments; several panels with panel_rows = 7 / 40): factors, RMSE and
    both residual copies against the golden output."""
    from conftest import load_golden
    g, d = load_golden(name)
    k, lam = int(g["k"][0]), float(g["lam"][0])
    T = kw.get("maxinneriter", 1)
    tag = "ccd_T3" if T == 3 else "ccd_T1"
    t = int(g[tag + "__maxiter"][0])
    p = _p(mfx, k, t, T, kernel_variant=2, **{a: b for a, b in kw.items() if a != "maxinneriter"})
    p.lambda_ = lam
    s = mfx.CcdSolver(d, mfx.test_data_of(d), p)
    assert d.nnz == 0 or s.layout_info()["csc"]["kind"] == "scatter"
    s.set_factors(np.array(g[tag + "__W0"], np.float32, copy=True))
    rep = s.iterate(t)
    W, H = s.get_factors()
    csc, csr = s.get_residual(d.nnz)
    s.close()
    scale = float(max(np.abs(g[tag + "__W"]).max(), np.abs(g[tag + "__H"]).max()))
    assert np.abs(W - g[tag + "__W"]).max() < 2e-3 * scale and np.abs(H - g[tag + "__H"]).max() < 2e-3 * scale
    assert np.all(np.abs(np.array([r.rmse for r in rep]) - g[tag + "__rmse"]) < 1e-4)
    if tag == "ccd_T1":
        assert np.max(np.abs(csc - g["ccd_T1__csc_val_final"])) < 2e-4 and np.max(np.abs(csr - g["ccd_T1__csr_val_final"])) < 2e-4


@pytest.mark.parametrize("wgs", ["1", "2", "3", "7", "64", "100000"])
def test_scatter_persistent_workgroup_ranges(mfx, orc, monkeypatch, wgs):
    """The scatter pass runs persistent workgroups over contiguous chunk ranges of the panel-major stream
    (ccd_scatter.hip); a range may start and end inside a panel, hold several whole panels, or be a single chunk.
    MFX_SCATTER_WGS pins the workgroup count (default: one per CU), so 1 = one workgroup walks every panel, 2 / 3 / 7
    = ranges that split panels at odd places, 100000 = one workgroup per chunk (round 2's launch shape).  The
    fixed-point sums make the result independent of how the chunks are grouped: factors and residual copies must be
    BIT-IDENTICAL across all of them, and match the oracle."""
    d = mfx.dataset.synth_ratings(3000, 2500, 60_000, seed=31, skew=0.6, test_frac=0.02, empty_row_frac=0.02, empty_col_frac=0.02)
    kw = dict(kernel_variant=2, panel_rows=200, tiles_per_span=2)  # 13 / 15 panels, a few chunks each
    monkeypatch.setenv("MFX_SCATTER_WGS", "100000")
    ref = _check(mfx, orc, d, 3, t=2, T=2, **kw)
    monkeypatch.setenv("MFX_SCATTER_WGS", wgs)
    got = _check(mfx, orc, d, 3, t=2, T=2, **kw)
    assert all(np.array_equal(bits(a), bits(b)) for a, b in zip(ref, got))


@pytest.mark.parametrize("groups,reserve", [(2, 0), (4, 16), (5, 8), (16, 0)])
def test_scatter_overlap_groups_are_bit_identical(mfx, monkeypatch, groups, reserve):
    """(r4) Sharded solve, scatter layout: the column pass launched panel group by panel group, with each group's combine ->
    all-reduce -> finalize on a second stream under the next group's pass, on fewer workgroups (CUs left to the
    collective).  The sums are fixed-point integers, so ANY grouping gives the bits of the unsplit pass: W, H and both
    residual copies of a 1-rank RCCL solve with G groups equal those of the same solve with one group and of the
    unsharded solve.  200 k x 30 k with explicit panels of 1500 (20 column panels) so that every group count is real."""
    d = mfx.dataset.synth_ratings(200000, 30000, 1500000, seed=77, skew=0.3, test_frac=0.002)
    k, t = 3, 2
    W0 = mfx.initial_col(k, d.rows)
    cnt = np.ascontiguousarray(np.diff(d.csc_col_ptr.astype(np.int64)).astype(np.uint32))

    def solve(with_comm, T=1):
        p = _p(mfx, k, t, T, kernel_variant=2, panel_rows=1500)
        comm = mfx.Comm(mfx.Comm.unique_id(), 0, 1, 0) if with_comm else None
        s = mfx.CcdSolver(d, mfx.test_data_of(d), p, comm=comm, global_col_nnz=cnt if comm else None,
                          global_test_nnz=d.nnz_test)
        s.set_factors(W0.copy())
        rep = s.iterate(t)
        W, H = s.get_factors()
        csc, csr = s.get_residual(d.nnz)
        s.close()
        if comm:
            comm.close()
        return W, H, csc, csr, np.array([r.rmse for r in rep])

    monkeypatch.setenv("MFX_OVERLAP_GROUPS", "1")
    ref = solve(False)
    one = solve(True)
    monkeypatch.setenv("MFX_OVERLAP_GROUPS", str(groups))
    monkeypatch.setenv("MFX_COMM_RESERVE_CUS", str(reserve))
    split = solve(True)
    for a, b, c in zip(ref[:5], one[:5], split[:5]):
        assert np.array_equal(bits(a), bits(b)) and np.array_equal(bits(a), bits(c))
    split2 = solve(True, T=2)  # the read-only column sweeps of T = 2 run over the grouped store as well
    monkeypatch.setenv("MFX_OVERLAP_GROUPS", "1")
    one2 = solve(True, T=2)
    for a, b in zip(one2[:5], split2[:5]):
        assert np.array_equal(bits(a), bits(b))


def test_scatter_non_finite_terms_are_not_silently_wrong(mfx):
    """the scatter pass accumulates in 64-bit fixed point, which cannot hold NaN / Inf / |x| >= 2^27.
    Such a term must not come back as a finite but wrong sum: the kernel flags the slab, the combine poisons the
    panel's sums with NaN, and -- like the flat path and the reference -- the factors / RMSE go non-finite.
    Every entry that is non-finite on the flat path must be non-finite in scatter mode too."""
    d = mfx.dataset.synth_ratings(400, 300, 8000, seed=9, test_frac=0.02).copy()
    q = 1234
    for bad in (np.float32(np.inf), np.float32(np.nan), np.float32(3e38)):
        e = d.copy()
        e.csr_val[q] = bad
        row = int(np.searchsorted(e.csr_row_ptr, q, side="right") - 1)
        col = int(e.csr_col_idx[q])
        lo, hi = int(e.csc_col_ptr[col]), int(e.csc_col_ptr[col + 1])
        e.csc_val[lo + int(np.where(e.csc_row_idx[lo:hi] == row)[0][0])] = bad
        out = {}
        for name, kw in (("flat", {}), ("scatter", {"kernel_variant": 2, "panel_rows": 64})):
            s = mfx.CcdSolver(e, mfx.test_data_of(e), _p(mfx, 3, 1, 1, **kw))
            s.set_factors(mfx.initial_col(3, e.rows))
            rep = s.iterate(1)
            out[name] = s.get_factors() + (rep[0].rmse,)
            s.close()
        for a, b in zip(out["flat"][:2], out["scatter"][:2]):
            assert not np.all(np.isfinite(b)), bad  # (3e38: a finite rating whose terms exceed the fixed-point range)
            if not np.isfinite(bad):
                assert not np.all(np.isfinite(a))
                assert np.all(~np.isfinite(b[~np.isfinite(a)])), bad


def test_scatter_sums_of_in_range_terms_do_not_wrap(mfx):
    """the 64-bit fixed-point accumulators (scale 2^36) hold |x| < 2^27.  Checking every TERM against 2^27 is not
    enough -- a long column of in-range terms sums past it and wraps to a FINITE wrong value.  The per-term bound is
    2^27 / (entries of the fullest row / column) now, so no sum can wrap: a column of 3000 ratings of 1e6 (terms u r ~ 1e5,
    each fine on its own, their sum ~ 2e8 > 1.3e8) makes the scatter path go non-finite -- loudly -- instead."""
    rows, cols = 4000, 300
    base = mfx.dataset.synth_ratings(rows, cols, 20000, seed=19, test_frac=0.01)
    r = np.repeat(np.arange(rows), np.diff(base.csr_row_ptr.astype(np.int64)))
    c = base.csr_col_idx.astype(np.int64)
    v = base.csr_val.copy()
    keep = c != 7
    heavy = np.arange(3000)
    r, c, v = np.concatenate([r[keep], heavy]), np.concatenate([c[keep], np.full(heavy.size, 7)]), np.concatenate([v[keep], np.full(heavy.size, 1e6, np.float32)])
    d = mfx.dataset.from_coo(rows, cols, r, c, v.astype(np.float32), base.test_row, base.test_col, base.test_val)
    out = {}
    for name, kw in (("flat", {}), ("scatter", {"kernel_variant": 2, "panel_rows": 64})):
        s = mfx.CcdSolver(d, mfx.test_data_of(d), _p(mfx, 2, 1, 1, **kw))
        s.set_factors(mfx.initial_col(2, d.rows))
        s.iterate(1)
        out[name] = s.get_factors()
        s.close()
    assert np.all(np.isfinite(out["flat"][1]))            # fp32 sums hold 2e8 without trouble ...
    Hs = out["scatter"][1]
    assert not np.isfinite(Hs[0, 7])                       # ... the fixed-point ones cannot, and say so
    ok = np.isfinite(Hs[0])
    assert np.allclose(Hs[0][ok], out["flat"][1][0][ok], rtol=2e-3, atol=1e-5)  # whatever stayed finite is right


def test_bad_arguments_are_errors(mfx):
    d = mfx.dataset.synth_ratings(50, 40, 500, seed=1)
    with pytest.raises(mfx.MfxError, match="maxinneriter"):
        mfx.CcdSolver(d, None, _p(mfx, 2, 1, 0))
    with pytest.raises(mfx.MfxError, match="k must be"):
        mfx.CcdSolver(d, None, _p(mfx, 0))
    with pytest.raises(mfx.MfxError, match="wg_waves|spans_per_wg"):
        mfx.CcdSolver(d, None, _p(mfx, 2, panel_rows=16, wg_waves=5))
    bad = d.copy()
    bad.csc_row_idx[0] = 10 ** 6  # index out of range must be caught on the host, not fault on the GPU
    with pytest.raises(mfx.MfxError, match="out of range|index"):
        mfx.CcdSolver(bad, None, _p(mfx, 2))
    for build in (1, 2):   # host and device layout builders both refuse it
        with pytest.raises(mfx.MfxError):
            mfx.CcdSolver(bad, None, _p(mfx, 2, layout_build=build))
    with pytest.raises(mfx.MfxError):
        mfx.CcdSolver(bad, None, _p(mfx, 2, kernel_variant=2))


def test_out_of_range_indices_are_errors_everywhere(mfx):
    """ALS's orientations and the test-set COO are uploaded and gathered with on the GPU, so a bad index
    must be an MFX_ERR_INVALID on the host side, never a device fault."""
    d = mfx.dataset.synth_ratings(50, 40, 500, seed=1)
    for field, value in (("csr_col_idx", 40), ("csc_row_idx", 50)):
        bad = d.copy()
        getattr(bad, field)[3] = value
        with pytest.raises(mfx.MfxError):
            mfx.AlsSolver(bad, None, _p(mfx, 4))
    for field, value in (("test_row", 50), ("test_col", 40)):
        bad = d.copy()
        getattr(bad, field)[0] = value
        with pytest.raises(mfx.MfxError):
            mfx.CcdSolver(d, mfx.test_data_of(bad), _p(mfx, 2))
        with pytest.raises(mfx.MfxError):
            mfx.AlsSolver(d, mfx.test_data_of(bad), _p(mfx, 4))
    bad = d.copy()
    bad.csr_row_ptr[5] = bad.csr_row_ptr[6] + 1   # non-monotone pointer array
    with pytest.raises(mfx.MfxError):
        mfx.CcdSolver(bad, None, _p(mfx, 2))
    with pytest.raises(mfx.MfxError):
        mfx.AlsSolver(bad, None, _p(mfx, 4))
    with pytest.raises(mfx.MfxError):   # half-step operator: index beyond the gathered factor
        ptr = np.array([0, 2], np.uint32); idx = np.array([0, 7], np.uint32); val = np.ones(2, np.float32)
        mfx.als_half(ptr, idx, val, np.ones((4, 3), np.float32), 3, 0.1)


def _shuffle_within_segments(ptr, idx, val, rng):
    idx, val = idx.copy(), val.copy()
    for s in range(ptr.shape[0] - 1):
        lo, hi = int(ptr[s]), int(ptr[s + 1])
        if hi - lo > 1:
            o = rng.permutation(hi - lo)
            idx[lo:hi] = idx[lo:hi][o]
            val[lo:hi] = val[lo:hi][o]
    return idx, val


@pytest.mark.parametrize("kw", [{"panel_rows": 64}, {"panel_rows": -64}, {"panel_rows": -1}, {"panel_rows": 64, "layout_build": 1}])
def test_unsorted_indices_inside_segments(mfx, orc, kw):
    """Entries of a row / column in arbitrary order (the reference's loader does not sort either): a
    panel is then visited several times per segment, so the layout must keep its full provenance array
    (no run compression) -- factors, test RMSE and both residual copies as the oracle's on the SAME order."""
    d = mfx.dataset.synth_ratings(700, 500, 30000, seed=77, skew=0.7, test_frac=0.02).copy()
    rng = np.random.default_rng(5)
    d.csr_col_idx, d.csr_val = _shuffle_within_segments(d.csr_row_ptr, d.csr_col_idx, d.csr_val, rng)
    d.csc_row_idx, d.csc_val = _shuffle_within_segments(d.csc_col_ptr, d.csc_row_idx, d.csc_val, rng)
    W0 = mfx.initial_col(5, d.rows)
    Wr, Hr, rmse_ref, _, csc_ref, csr_ref = orc.ccdr1(d, W0, 5, 0.05, 3, 2, 2)
    s = mfx.CcdSolver(d, mfx.test_data_of(d), _p(mfx, 5, 3, 2, **kw))
    s.set_factors(W0.copy())
    rep = s.iterate(3)
    W, H = s.get_factors()
    csc, csr = s.get_residual(d.nnz)
    s.close()
    scale = max(float(np.abs(Wr).max()), float(np.abs(Hr).max()))
    assert np.max(np.abs(W - Wr)) < 2e-3 * scale and np.max(np.abs(H - Hr)) < 2e-3 * scale
    assert np.all(np.abs(np.array([r.rmse for r in rep]) - rmse_ref) < 1e-4)
    assert np.max(np.abs(csc - csc_ref)) < 1e-3 and np.max(np.abs(csr - csr_ref)) < 1e-3
