"""GPU parity tests for the CCD++ hot path: every call goes through libmfx.so's C ABI and is
checked against the CPU oracle (itself pinned bit-exact to the reference by
test_oracle_pinned.py) and against the committed golden vectors.

Tolerances (fp32): the residual update is elementwise and must be BIT-EXACT; anything that sums
over a row/column differs from the CPU only by summation order, bounded per sweep by
~len * 2^-24 relative -- 2e-5 is asserted; whole solves amplify that over k*iters updates, so
factors are compared at 2e-3 relative-to-scale and test RMSE at the north-star's 1e-4.
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


@pytest.fixture(scope="module")
def medium(mfx):
    # ML-1M shaped (BASELINE.json configs[1]): 6040 x 3706, Z = 1e6, some empty rows/cols
    return mfx.dataset.synth_ratings(6040, 3706, 1_000_000, seed=7, skew=0.9, test_frac=0.01,
                                     empty_row_frac=0.01, empty_col_frac=0.02)


def relerr(a, b):
    scale = max(1e-30, float(np.max(np.abs(b))))
    return float(np.max(np.abs(a.astype(np.float64) - b.astype(np.float64)))) / scale


# ------------------------------------------------------------------ single operators
OP_VARIANTS = [0, 1, 2, 16, 100, -16, -100]  # wave-per-segment, flat/L2 gather, flat/LDS auto, LDS panels of 16 / 100 entries, cache panels


@pytest.mark.parametrize("variant", OP_VARIANTS)
@pytest.mark.parametrize("name", CASES)
def test_update_rating_bit_exact_golden(mfx, name, variant):
    g, d = load_golden(name)
    u1, v1 = g["step_u1"].copy(), g["step_v1"].copy()
    csc, csr = d.csc_val.copy(), d.csr_val.copy()
    mfx.update_rating(d.csc_col_ptr, d.csc_row_idx, csc, u1, v1, False, variant)
    mfx.update_rating(d.csr_row_ptr, d.csr_col_idx, csr, v1, u1, False, variant)
    assert np.array_equal(bits(csc), bits(g["step_csc_sub"]))
    assert np.array_equal(bits(csr), bits(g["step_csr_sub"]))
    k = int(g["k"][0]); t1 = 1 % k
    W0, H0 = g["ccd_T1__W0"], g["ccd_T1__H0"]
    mfx.update_rating(d.csc_col_ptr, d.csc_row_idx, csc, W0[t1].copy(), H0[t1].copy(), True, variant)
    mfx.update_rating(d.csr_row_ptr, d.csr_col_idx, csr, H0[t1].copy(), W0[t1].copy(), True, variant)
    assert np.array_equal(bits(csc), bits(g["step_csc_add"]))
    assert np.array_equal(bits(csr), bits(g["step_csr_add"]))


@pytest.mark.parametrize("variant", OP_VARIANTS)
@pytest.mark.parametrize("name", CASES)
def test_rank_one_sweep_golden(mfx, name, variant):
    g, d = load_golden(name)
    lam = float(g["lam"][0])
    v1 = mfx.rank_one_sweep(d.csc_col_ptr, d.csc_row_idx, d.csc_val, g["ccd_T1__W0"][0].copy(), lam, variant)
    assert relerr(v1, g["step_v1"]) < 2e-5
    empty = np.diff(d.csc_col_ptr.astype(np.int64)) == 0
    assert np.all(v1[empty] == 0)  # empty column -> exactly 0 (src/CCD.cpp:8)
    u1 = mfx.rank_one_sweep(d.csr_row_ptr, d.csr_col_idx, d.csr_val, g["step_v1"].copy(), lam, variant)
    assert relerr(u1, g["step_u1"]) < 2e-5


@pytest.mark.parametrize("variant", [0, 1, 2, 1000])
def test_single_ops_medium(mfx, orc, medium, variant):
    d = medium
    rng = np.random.default_rng(3)
    u = rng.uniform(0.001, 0.101, d.rows).astype(np.float32)
    v_ref = orc.rank_one_sweep(d.csc_col_ptr, d.csc_row_idx, d.csc_val, u, 0.05, 4)
    v = mfx.rank_one_sweep(d.csc_col_ptr, d.csc_row_idx, d.csc_val, u, 0.05, variant)
    assert relerr(v, v_ref) < 2e-5
    u_ref = orc.rank_one_sweep(d.csr_row_ptr, d.csr_col_idx, d.csr_val, v_ref, 0.05, 4)
    u2 = mfx.rank_one_sweep(d.csr_row_ptr, d.csr_col_idx, d.csr_val, v_ref, 0.05, variant)
    assert relerr(u2, u_ref) < 2e-5
    a, b = d.csr_val.copy(), d.csr_val.copy()
    orc.update_rating(d.csr_row_ptr, d.csr_col_idx, a, v_ref, u_ref, False, 4)
    mfx.update_rating(d.csr_row_ptr, d.csr_col_idx, b, v_ref, u_ref, False, variant)
    assert np.array_equal(bits(a), bits(b))


def test_flat_kernel_long_and_degenerate_segments(mfx, orc):
    """One segment far longer than a span (carries across many spans), runs of 1-entry segments
    (several heads per lane), empty segments, a tail that is not a multiple of 4."""
    rng = np.random.default_rng(5)
    lens = np.concatenate([[0, 0, 20011, 0], np.ones(700, np.int64), [3, 2, 1, 0, 5, 4099, 1, 1, 0, 257, 255, 1023],
                           rng.integers(0, 40, 300), [7]])
    ptr = np.zeros(lens.size + 1, np.uint32); ptr[1:] = np.cumsum(lens)
    nnz, nvec = int(ptr[-1]), 5000
    idx = rng.integers(0, nvec, nnz).astype(np.uint32)
    val = rng.uniform(1, 5, nnz).astype(np.float32)
    vec = rng.uniform(-1, 1, nvec).astype(np.float32)
    ref = orc.rank_one_sweep(ptr, idx, val, vec, 0.1, 2)
    # (explicit span lengths are exercised through the resident solver: test_ccdpp_ml1m_shape_vs_oracle)
    per_seg = rng.uniform(-1, 1, lens.size).astype(np.float32)
    a = val.copy()
    orc.update_rating(ptr, idx, a, vec, per_seg, True, 2)
    for variant in (1, 2, 16, 333, -333):  # 333-entry panels: 16 panels, most of them cutting the long segment
        out = mfx.rank_one_sweep(ptr, idx, val, vec, 0.1, variant)
        assert relerr(out, ref) < 2e-5, variant
        assert np.all(out[lens == 0] == 0)
        b = val.copy()
        mfx.update_rating(ptr, idx, b, vec, per_seg, True, variant)
        assert np.array_equal(bits(a), bits(b)), variant


@pytest.mark.parametrize("name", CASES)
def test_test_rmse_matches_calrmse(mfx, orc, name):
    g, d = load_golden(name)
    k = int(g["k"][0])
    W0, H0 = np.ascontiguousarray(g["ccd_T1__W0"]), np.ascontiguousarray(g["ccd_T1__H0"])
    got = mfx.test_rmse(mfx.test_data_of(d), W0, H0, d.rows, d.cols, k, False)
    assert abs(got - float(g["step_rmse_init_ccd"][0])) < 1e-12
    Wa, Ha = np.ascontiguousarray(g["als__W"]), np.ascontiguousarray(g["als__H"])
    got = mfx.test_rmse(mfx.test_data_of(d), Wa, Ha, d.rows, d.cols, k, True)
    assert abs(got - orc.calrmse(d, Wa, Ha, k, True)) < 1e-12


# ------------------------------------------------------------------ whole solves, golden
def _params(mfx, k, lam, t, T, schedule, variant, tiles=0, panel_rows=0, wg_waves=0, layout_build=0):
    p = mfx.parameter()
    p.k, p.lambda_, p.maxiter, p.maxinneriter = k, lam, t, T
    p.schedule, p.kernel_variant, p.tiles_per_span = schedule, variant, tiles
    p.panel_rows, p.wg_waves, p.layout_build = panel_rows, wg_waves, layout_build
    return p


# (schedule, kernel_variant, panel_rows): fused with LDS panels (auto = one panel here), fused with
# tiny panels (many panels per matrix), fused gathering from L2, as-written wave, as-written flat,
# fused / as-written with tiny CACHE panels (panel-major order, global indices, gather from L2)
SCHEDULES = [(1, 1, 0), (1, 1, 24), (1, 1, -1), (0, 0, 0), (0, 1, 0), (0, 1, 24), (1, 1, -24), (0, 1, -24)]


@pytest.mark.parametrize("schedule,variant,panel_rows", SCHEDULES)
@pytest.mark.parametrize("tag", ["ccd_T1", "ccd_T3"])
@pytest.mark.parametrize("name", CASES)
def test_ccdpp_matches_reference_golden(mfx, name, tag, schedule, variant, panel_rows):
    """kernel_wrapper_ccdpp_NV vs the reference's own ccdr1_OMP output (tests/golden)."""
    g, d = load_golden(name)
    k, lam = int(g["k"][0]), float(g["lam"][0])
    t, T = int(g[tag + "__maxiter"][0]), int(g[tag + "__maxinner"][0])
    W = np.array(g[tag + "__W0"], np.float32, copy=True)
    H = np.array(g[tag + "__H0"], np.float32, copy=True)  # content must be ignored (H starts at 0)
    reports = mfx.kernel_wrapper_ccdpp_NV(d, mfx.test_data_of(d), W, H,
                                          _params(mfx, k, lam, t, T, schedule, variant, panel_rows=panel_rows))
    assert mfx.kernel_wrapper_ccdpp_NV.last_status == 0
    assert relerr(W, g[tag + "__W"]) < 2e-3 and relerr(H, g[tag + "__H"]) < 2e-3
    rmse = np.array([r.rmse for r in reports])
    assert np.all(np.abs(rmse - g[tag + "__rmse"]) < 1e-4), (rmse, g[tag + "__rmse"])
    assert mfx.golden_compare(W, g[tag + "__W"], k, d.rows, quiet=True) == 0  # the reference's own 10 % bar
    assert abs(mfx.calculate_rmse_directly(W, H, mfx.test_data_of(d), k, False, quiet=True) - float(g[tag + "__final_rmse"][0])) < 1e-4


@pytest.mark.parametrize("schedule,variant,panel_rows", SCHEDULES)
@pytest.mark.parametrize("name", CASES)
def test_residual_state_matches_reference(mfx, name, schedule, variant, panel_rows):
    """After the run both residual copies equal the reference's mutated csc/csr value arrays."""
    g, d = load_golden(name)
    k, lam = int(g["k"][0]), float(g["lam"][0])
    s = mfx.CcdSolver(d, mfx.test_data_of(d), _params(mfx, k, lam, 3, 1, schedule, variant, panel_rows=panel_rows))
    s.set_factors(np.ascontiguousarray(g["ccd_T1__W0"]))
    s.iterate(3)
    csc, csr = s.get_residual(d.nnz)
    s.close()
    assert np.max(np.abs(csc - g["ccd_T1__csc_val_final"])) < 2e-4
    assert np.max(np.abs(csr - g["ccd_T1__csr_val_final"])) < 2e-4


# The one-time layout build has two implementations -- a device pipeline (default for grouped patterns)
# and the host builder (fallback, and the checker): they must produce the SAME stored order, hence
# bitwise identical factors, RMSE and residual copies, for every layout kind.
@pytest.mark.parametrize("panel_rows", [0, 24, -24, -1])
@pytest.mark.parametrize("tag", ["ccd_T1", "ccd_T3"])
@pytest.mark.parametrize("name", CASES)
def test_device_built_layout_equals_host_built(mfx, name, tag, panel_rows):
    g, d = load_golden(name)
    k, lam = int(g["k"][0]), float(g["lam"][0])
    t, T = int(g[tag + "__maxiter"][0]), int(g[tag + "__maxinner"][0])
    out = []
    for layout_build in (1, 2):
        s = mfx.CcdSolver(d, mfx.test_data_of(d), _params(mfx, k, lam, t, T, 1, 1, panel_rows=panel_rows, layout_build=layout_build))
        info = s.layout_info()
        s.set_factors(np.array(g[tag + "__W0"], np.float32, copy=True))
        reports = s.iterate(t)
        W, H = s.get_factors()
        csc, csr = s.get_residual(d.nnz)
        s.close()
        out.append((info, [r.rmse for r in reports], W, H, csc, csr))
    (i1, r1, W1, H1, c1, q1), (i2, r2, W2, H2, c2, q2) = out
    assert i1 == i2 and r1 == r2
    assert np.array_equal(bits(W1), bits(W2)) and np.array_equal(bits(H1), bits(H2))
    assert np.array_equal(bits(c1), bits(c2)) and np.array_equal(bits(q1), bits(q2))
    assert relerr(W2, g[tag + "__W"]) < 2e-3 and relerr(H2, g[tag + "__H"]) < 2e-3
    assert np.all(np.abs(np.array(r2) - g[tag + "__rmse"]) < 1e-4)


# ------------------------------------------------------------------ whole solves, ML-1M shape
@pytest.mark.parametrize("schedule,variant,tiles,panel_rows,wg_waves,layout_build", [
    (1, 1, 0, 0, 0, 0), (1, 1, 2, 1000, 4, 0), (1, 1, 16, 500, 16, 0), (1, 1, 4, -1, 0, 0), (1, 1, 8, 2048, 8, 0),
    (0, 0, 0, 0, 0, 0), (0, 1, 4, 700, 8, 0), (1, 1, 4, -700, 0, 0), (1, 1, 0, -2000, 0, 0),
    (1, 1, 0, 0, 0, 1), (1, 1, 8, 2048, 8, 1), (0, 1, 0, -2000, 0, 1)])
def test_ccdpp_ml1m_shape_vs_oracle(mfx, orc, medium, schedule, variant, tiles, panel_rows, wg_waves, layout_build):
    d, k, lam, t = medium, 40, 0.05, 3
    W0 = mfx.initial_col(k, d.rows)
    Wr, Hr, rmse_ref, _, csc_ref, csr_ref = orc.ccdr1(d, W0, k, lam, t, 1, orc.max_threads())
    s = mfx.CcdSolver(d, mfx.test_data_of(d), _params(mfx, k, lam, t, 1, schedule, variant, tiles, panel_rows, wg_waves, layout_build))
    s.set_factors(W0.copy())
    reports = s.iterate(t)
    W, H = s.get_factors()
    csc, csr = s.get_residual(d.nnz)
    s.close()
    rmse = np.array([r.rmse for r in reports])
    assert np.all(np.abs(rmse - rmse_ref) < 1e-4), (rmse, rmse_ref)
    assert relerr(W, Wr) < 2e-3 and relerr(H, Hr) < 2e-3
    assert mfx.golden_compare(H, Hr, k, d.cols, quiet=True) <= 0.001 * k * d.cols
    assert np.max(np.abs(csc - csc_ref)) < 1e-3 and np.max(np.abs(csr - csr_ref)) < 1e-3


def test_ccdpp_runs_are_bitwise_reproducible_and_split_invariant(mfx, medium):
    """No atomics anywhere: two solves give identical bits; iterate(1)x3 == iterate(3)."""
    d, k = medium, 16
    W0 = mfx.initial_col(k, d.rows)
    outs = []
    for split in (False, False, True):
        s = mfx.CcdSolver(d, mfx.test_data_of(d), _params(mfx, k, 0.05, 3, 1, 1, 1))
        s.set_factors(W0.copy())
        if split:
            for _ in range(3):
                s.iterate(1)
        else:
            s.iterate(3)
        outs.append(s.get_factors() + s.get_residual(d.nnz))
        s.close()
    for a, b in zip(outs[0], outs[1]):
        assert np.array_equal(bits(a), bits(b))
    for a, b in zip(outs[0], outs[2]):
        assert np.array_equal(bits(a), bits(b))


def test_k1_and_rank_wraparound(mfx, orc):
    """k = 1: the rank being added back is the rank just subtracted (next == t)."""
    d = mfx.dataset.synth_ratings(500, 300, 20000, seed=21, skew=0.8, test_frac=0.02)
    W0 = mfx.initial_col(1, d.rows)
    Wr, Hr, rmse_ref, *_ = orc.ccdr1(d, W0, 1, 0.05, 4, 2, 2)
    s = mfx.CcdSolver(d, mfx.test_data_of(d), _params(mfx, 1, 0.05, 4, 2, 1, 1))
    s.set_factors(W0.copy())
    rep = s.iterate(4)
    W, H = s.get_factors()
    s.close()
    assert relerr(W, Wr) < 1e-3 and relerr(H, Hr) < 1e-3
    assert np.all(np.abs(np.array([r.rmse for r in rep]) - rmse_ref) < 1e-4)


def test_errors_are_reported_not_fatal(mfx, medium):
    p = _params(mfx, 4, 0.05, 1, 1, 1, 1)
    s = mfx.CcdSolver(medium, None, p)
    with pytest.raises(mfx.MfxError):
        s.iterate(1)  # factors not set
    s.close()
    p.device = 99
    with pytest.raises(mfx.MfxError, match="device"):
        mfx.CcdSolver(medium, None, p)


def test_sharded_path_single_rank_rccl(mfx, medium, monkeypatch):
    """The multi-GPU code path (combine -> RCCL all-reduce -> finalize from the reduced buffer, fp64
    all-reduce of the squared error) with a 1-rank communicator must reproduce the unsharded solve
    (of the same kernels: the flat-stream passes -- r4's segment-owner passes of small single-GPU solves add in another order)."""
    monkeypatch.setenv("MFX_OWNER_PASSES", "0")
    d, k = medium, 8
    W0 = mfx.initial_col(k, d.rows)
    s0 = mfx.CcdSolver(d, mfx.test_data_of(d), _params(mfx, k, 0.05, 2, 2, 1, 1))
    s0.set_factors(W0.copy()); r0 = s0.iterate(2); f0 = s0.get_factors(); s0.close()
    comm = mfx.Comm(mfx.Comm.unique_id(), 0, 1, 0)
    cnt = np.ascontiguousarray(np.diff(d.csc_col_ptr.astype(np.int64)).astype(np.uint32))
    s1 = mfx.CcdSolver(d, mfx.test_data_of(d), _params(mfx, k, 0.05, 2, 2, 1, 1), comm=comm, global_col_nnz=cnt,
                       global_test_nnz=d.nnz_test)
    s1.set_factors(W0.copy()); r1 = s1.iterate(2); f1 = s1.get_factors(); s1.close(); comm.close()
    assert np.array_equal(bits(f0[0]), bits(f1[0])) and np.array_equal(bits(f0[1]), bits(f1[1]))
    assert [r.rmse for r in r0] == [r.rmse for r in r1]


@pytest.mark.parametrize("nshards,schedule,variant,T", [(2, 1, 1, 1), (3, 1, 1, 2), (4, 0, 1, 1), (2, 0, 0, 1),
                                                        (3, 1, 2, 1), (2, 1, 2, 2), (2, 0, 2, 1)])  # variant 2: scatter layout (config 5's sharded path)
def test_sharded_solve_multi_rank_loopback(mfx, orc, medium, nshards, schedule, variant, T):
    """N user-row-block shards, one solver per shard (threads of this process, loopback communicator:
    RCCL refuses two ranks on one GPU), through the real HIP kernels: local (g,h) partials, all-reduce,
    division by lambda * GLOBAL |Omega_c|, replicated H, sharded W, global RMSE.  Must match the
    unsharded CPU oracle like the single-GPU solve does."""
    import threading
    d, k, lam, t = medium, 12, 0.05, 2
    W0 = mfx.initial_col(k, d.rows)
    Wr, Hr, rmse_ref, *_ = orc.ccdr1(d, W0, k, lam, t, T, orc.max_threads())
    bounds = mfx.partition_rows(d, nshards)
    gcnt = np.ascontiguousarray(np.diff(d.csc_col_ptr.astype(np.int64)).astype(np.uint32))
    out, errs = [None] * nshards, []
    group = 1000 + nshards * 10 + schedule * 2 + variant

    def run(r):
        try:
            lo, hi = int(bounds[r]), int(bounds[r + 1])
            shard = mfx.extract_shard(d, lo, hi)
            comm = mfx.Comm(None, r, nshards, 0, local_group=group)
            s = mfx.CcdSolver(shard, mfx.test_data_of(shard), _params(mfx, k, lam, t, T, schedule, variant),
                              comm=comm, global_col_nnz=gcnt, global_test_nnz=d.nnz_test)
            s.set_factors(np.ascontiguousarray(W0[:, lo:hi]))
            rep = s.iterate(t)
            out[r] = (s.get_factors(), [x.rmse for x in rep])
            s.close(); comm.close()
        except Exception as e:  # surface failures instead of dead-locking the other ranks' rendezvous
            errs.append(e)
            raise

    th = [threading.Thread(target=run, args=(r,)) for r in range(nshards)]
    [x.start() for x in th]
    [x.join(timeout=120) for x in th]
    assert not errs and all(o is not None for o in out), errs
    W = np.concatenate([o[0][0] for o in out], axis=1)
    assert relerr(W, Wr) < 2e-3
    for (Wl, Hl), rm in out:
        assert relerr(Hl, Hr) < 2e-3 and np.all(np.abs(np.array(rm) - rmse_ref) < 1e-4)
    assert all(np.array_equal(bits(out[0][0][1]), bits(o[0][1])) for o in out)  # H replicas identical


def test_failing_shard_does_not_strand_the_others(mfx, medium):
    """The multi-shard protocol (mfx.h, mfx_comm_agree / mfx_comm_abort): a rank whose setup fails reports it
    through agree() and nobody enters a collective; a rank that fails later aborts the communicator and the
    ranks already waiting inside an all-reduce come back with an error instead of hanging."""
    import threading
    d, k = medium, 4
    nshards = 3
    bounds = mfx.partition_rows(d, nshards)
    gcnt = np.ascontiguousarray(np.diff(d.csc_col_ptr.astype(np.int64)).astype(np.uint32))
    W0 = mfx.initial_col(k, d.rows)

    # (1) setup failure on rank 1: everybody learns the worst status, nobody iterates
    seen = [None] * nshards

    def setup(r):
        comm = mfx.Comm(None, r, nshards, 0, local_group=7001)
        status = 0
        try:
            if r == 1:
                raise mfx.MfxError("injected setup failure")
            lo, hi = int(bounds[r]), int(bounds[r + 1])
            shard = mfx.extract_shard(d, lo, hi)
            s = mfx.CcdSolver(shard, mfx.test_data_of(shard), _params(mfx, k, 0.05, 1, 1, 1, 1), comm=comm,
                              global_col_nnz=gcnt, global_test_nnz=d.nnz_test)
            s.close()
        except mfx.MfxError:
            status = -1
        seen[r] = comm.agree(status)
        comm.close()

    th = [threading.Thread(target=setup, args=(r,)) for r in range(nshards)]
    [x.start() for x in th]
    [x.join(timeout=120) for x in th]
    assert not any(x.is_alive() for x in th) and seen == [-1, -1, -1], seen

    # (2) rank 2 dies after agree: it aborts, the others' iterate() returns an error
    res = [None] * nshards

    def run(r):
        lo, hi = int(bounds[r]), int(bounds[r + 1])
        shard = mfx.extract_shard(d, lo, hi)
        comm = mfx.Comm(None, r, nshards, 0, local_group=7002)
        s = mfx.CcdSolver(shard, mfx.test_data_of(shard), _params(mfx, k, 0.05, 1, 1, 1, 1), comm=comm,
                          global_col_nnz=gcnt, global_test_nnz=d.nnz_test)
        s.set_factors(np.ascontiguousarray(W0[:, lo:hi]))
        assert comm.agree(0) == 0
        if r == 2:
            comm.abort()
            res[r] = "aborted"
        else:
            try:
                s.iterate(1)
                res[r] = "finished"
            except mfx.MfxError as e:
                res[r] = "error: " + str(e)
        s.close(); comm.close()

    th = [threading.Thread(target=run, args=(r,)) for r in range(nshards)]
    [x.start() for x in th]
    [x.join(timeout=120) for x in th]
    assert not any(x.is_alive() for x in th), res
    assert res[2] == "aborted" and all(isinstance(x, str) and x.startswith("error") and "aborted" in x for x in res[:2]), res


@pytest.mark.parametrize("panel_rows,tiles,T", [(64, 0, 1), (97, 2, 1), (4096, 0, 2), (300, 4, 1), (-1, 0, 1), (-1, 2, 3), (0, 4, 2)])
def test_fused_finalize_equals_the_separate_kernel(mfx, medium, monkeypatch, panel_rows, tiles, T):
    """Opt-in (MFX_FUSE_FINALIZE=1, =2 with the dispatch sorted by first segment; off by default because it measured
    slower): the finalize of a fused pass runs inside the pass (the workgroup whose arrival completes a group of segments
    computes g / (lambda |Omega| + h) for it, reading the other workgroups' partial sums past the caches):
    same lookups and order of additions as k_finalize, so factors, RMSE trace and residuals are bit-identical to
    a solve with MFX_FUSE_FINALIZE=0 -- on layouts with many panels and short spans (many chunks per group,
    segments crossing chunks) and over several outer iterations (the arrival counters reset themselves).
    (r4) Also the PLAIN layout (panel_rows -1, or 0 = chosen for this small matrix), where it is the default: 256-thread
    workgroups of four spans, the read-only sweeps of T > 1 fused as well."""
    d = medium
    k, lam, t = 6, 0.05, 4
    W0 = mfx.initial_col(k, d.rows)
    outs = []
    for fuse in ("1", "2", "0"):
        monkeypatch.setenv("MFX_FUSE_FINALIZE", fuse)
        s = mfx.CcdSolver(d, mfx.test_data_of(d), _params(mfx, k, lam, t, T, 1, 1, tiles, panel_rows))
        s.set_factors(W0.copy())
        rep = s.iterate(t)
        W, H = s.get_factors()
        csc, csr = s.get_residual(d.nnz)
        s.close()
        outs.append((W, H, csc, csr, np.array([r.rmse for r in rep])))
    for o in outs[:2]:
        for a, b in zip(o[:4], outs[2][:4]):
            assert np.array_equal(bits(a), bits(b))
        assert np.array_equal(o[4], outs[2][4])


def test_plain_layout_can_fuse_the_finalize(mfx, medium, monkeypatch):
    """(r4) The fused finalize also exists for the plain layout of small matrices (256-thread workgroups of four spans; the
    read-only sweeps of T > 1 too): with MFX_FUSE_FINALIZE=1 every pass carries its finalize -- two launches per rank instead
    of four.  Opt-in, because under hipGraph replay it measured slower than the kernel boundary it removes (ccd_solver.hpp)."""
    d, k = medium, 5

    def launches(panel_rows, T=1):
        p = _params(mfx, k, 0.05, 1, T, 1, 1, 0, panel_rows)
        p.profile = 1
        s = mfx.CcdSolver(d, mfx.test_data_of(d), p)
        s.set_factors(mfx.initial_col(k, d.rows))
        s.iterate(1)
        kt = s.kernel_times()
        info = s.layout_info()
        s.close()
        return {n: v[1] for n, v in kt.items()}, info

    monkeypatch.delenv("MFX_FUSE_FINALIZE", raising=False)
    monkeypatch.setenv("MFX_OWNER_PASSES", "0")  # (the default for this size -- segment-owner passes -- has its own test below)
    kt, info = launches(0)
    assert info["csc"]["kind"] == "plain" and info["csr"]["kind"] == "plain", info
    assert kt["ccd_finalize"] == 2 * k, kt
    monkeypatch.setenv("MFX_FUSE_FINALIZE", "1")
    kt, _ = launches(0)
    assert kt.get("ccd_finalize", 0) == 0 and kt["ccd_fused_csc_pass"] == k and kt["ccd_fused_csr_pass"] == k, kt
    kt, _ = launches(0, T=3)
    assert kt.get("ccd_finalize", 0) == 0 and kt["ccd_flat_sweep"] == 4 * k, kt


@pytest.mark.parametrize("T", [1, 3])
def test_small_matrices_take_the_segment_owner_passes(mfx, orc, medium, monkeypatch, T):
    """(r4) Small single-GPU solves (plain layout on both sides) run two launches per rank: every row / column has ONE owner
    (a wavefront, or a whole workgroup from 1024 entries on) that walks it, divides and writes the factor entry and the
    operand packs itself (k_seg_owner) -- no finalize kernel, no cross-workgroup step.  Same element arithmetic as every other
    path (the residual copies stay bit-identical to each other); the sums are added in the owner's fixed order: oracle
    tolerances, bitwise reproducible, and the flat-stream path is still there (MFX_OWNER_PASSES=0)."""
    d, k, lam, t = medium, 6, 0.05, 3
    W0 = mfx.initial_col(k, d.rows)
    Wr, Hr, rmse_ref, *_ = orc.ccdr1(d, W0, k, lam, t, T, orc.max_threads())
    outs = {}
    for mode in ("owner", "owner_again", "flat"):
        if mode == "flat":
            monkeypatch.setenv("MFX_OWNER_PASSES", "0")
        else:
            monkeypatch.delenv("MFX_OWNER_PASSES", raising=False)
        p = _params(mfx, k, lam, t, T, 1, 1)
        p.profile = 1
        s = mfx.CcdSolver(d, mfx.test_data_of(d), p)
        s.set_factors(W0.copy())
        rep = s.iterate(t)
        W, H = s.get_factors()
        csc, csr = s.get_residual(d.nnz)
        kt = {n: v[1] for n, v in s.kernel_times().items()}
        s.close()
        outs[mode] = (W, H, csc, csr, np.array([r.rmse for r in rep]), kt)
    W, H, csc, csr, rm, kt = outs["owner"]
    assert kt.get("ccd_finalize", 0) == 0 and kt["ccd_fused_csc_pass"] == k * t and kt["ccd_fused_csr_pass"] == k * t, kt
    assert outs["flat"][5]["ccd_finalize"] == 2 * k * t * T
    assert relerr(W, Wr) < 2e-3 and relerr(H, Hr) < 2e-3 and np.all(np.abs(rm - rmse_ref) < 1e-4)
    assert all(np.array_equal(bits(a), bits(b)) for a, b in zip(outs["owner"][:4], outs["owner_again"][:4]))  # reproducible
    # both residual copies hold the same values (same operands, same order of the two updates, in either kernel family)
    order = np.lexsort((np.repeat(np.arange(d.rows), np.diff(d.csr_row_ptr.astype(np.int64))), d.csr_col_idx))
    assert np.array_equal(bits(csr[order]), bits(csc))
    assert relerr(W, outs["flat"][0]) < 1e-3 and relerr(H, outs["flat"][1]) < 1e-3


# ---- the flags the reference parses and ignores (-N, -e, -p/-q), opt-in with their LIBPMF meaning ---------------------
def _ext_params(mfx, k, lam, t, T, schedule, variant, panel_rows=0, **ext):
    p = _params(mfx, k, lam, t, T, schedule, variant, panel_rows=panel_rows)
    p.libpmf_flags = 1
    p.eps = 0.0
    for a, b in ext.items():
        setattr(p, a, b)
    return p


@pytest.mark.parametrize("schedule,variant,panel_rows", [(1, 1, 0), (1, 1, 64), (0, 1, 0), (0, 0, 0), (1, 2, 0)])
def test_rank_trace_is_calrmse_r1(mfx, orc, medium, schedule, variant, panel_rows):
    """-p 1 -q 1 under libpmf_flags: the per-rank RMSE of the reference's calrmse_r1 (src/tools.cpp:261-270; its call
    site is the commented block src/CCD.cpp:141-148).  Checked against the oracle's restatement rank by rank, and --
    the property that pins it to the reference's own outputs -- the value after the LAST rank of an outer iteration is
    that iteration's test RMSE (the incrementally updated test residual equals the directly computed one)."""
    d = medium
    k, lam, t, T = 5, 0.05, 3, 2
    W0 = mfx.initial_col(k, d.rows)
    Wr, Hr, rmse_ref, trace_ref, done_ref = orc.ccdr1_ext(d, W0, k, lam, t, T, 2)
    s = mfx.CcdSolver(d, mfx.test_data_of(d), _ext_params(mfx, k, lam, t, T, schedule, variant, panel_rows, verbose=1, do_predict=1))
    s.set_factors(W0.copy())
    rep = s.iterate(t)
    trace, secs, done = s.rank_trace(t, k)
    W, H = s.get_factors()
    s.close()
    assert list(done) == [k] * t and trace.shape == (t, k) and np.all(secs > 0)
    assert np.all(np.abs(trace - trace_ref) < 1e-4), (trace, trace_ref)
    rm = np.array([r.rmse for r in rep])
    assert np.all(np.abs(trace[:, -1] - rm) < 2e-6), (trace[:, -1], rm)
    assert np.all(np.abs(rm - rmse_ref) < 1e-4)


@pytest.mark.parametrize("schedule,variant,panel_rows", [(1, 1, 0), (1, 1, 64), (0, 1, 0), (1, 2, 0)])
def test_nmf_projection_vs_oracle(mfx, orc, medium, schedule, variant, panel_rows):
    """-N 1 under libpmf_flags ("parity unpinned": no code for it in the reference tree; LIBPMF's meaning, checked
    against the oracle's restatement): every factor entry is >= 0, RMSE trace and factors as the oracle's."""
    d = medium
    k, lam, t = 5, 0.05, 3
    W0 = mfx.initial_col(k, d.rows)
    Wr, Hr, rmse_ref, _, _ = orc.ccdr1_ext(d, W0, k, lam, t, 1, 2, do_nmf=1)
    assert (Wr < 0).sum() == 0 and (Hr < 0).sum() == 0
    _, _, rmse_plain, _, _ = orc.ccdr1_ext(d, W0, k, lam, t, 1, 2)
    assert np.abs(rmse_plain - rmse_ref).max() > 1e-5  # the projection is active on this input
    s = mfx.CcdSolver(d, mfx.test_data_of(d), _ext_params(mfx, k, lam, t, 1, schedule, variant, panel_rows, do_nmf=1))
    s.set_factors(W0.copy())
    rep = s.iterate(t)
    W, H = s.get_factors()
    s.close()
    assert W.min() >= 0 and H.min() >= 0
    assert np.all(np.abs(np.array([r.rmse for r in rep]) - rmse_ref) < 1e-4)
    scale = float(max(np.abs(Wr).max(), np.abs(Hr).max()))
    assert np.abs(W - Wr).max() < 2e-3 * scale and np.abs(H - Hr).max() < 2e-3 * scale
    # and the default ignores the flag, like the reference
    p = _params(mfx, k, lam, t, 1, schedule, variant, panel_rows=panel_rows)
    p.do_nmf = 1
    s = mfx.CcdSolver(d, mfx.test_data_of(d), p)
    s.set_factors(W0.copy())
    rep = s.iterate(t)
    s.close()
    assert np.all(np.abs(np.array([r.rmse for r in rep]) - rmse_plain) < 1e-4)


@pytest.mark.parametrize("schedule,variant,panel_rows,eps", [(1, 1, 0, 0.05), (1, 1, 64, 0.05), (0, 1, 0, 0.05), (1, 2, 0, 0.05), (1, 1, 0, 0.6)])
def test_eps_stopping_rule_vs_oracle(mfx, orc, medium, schedule, variant, panel_rows, eps):
    """-e under libpmf_flags ("parity unpinned", LIBPMF's rule restated in the oracle): inner iterations stop when the
    function decrease falls below eps x its running maximum; five ranks that stop in their first inner iteration end
    the outer iteration.  Same ranks skipped, same RMSE trace and factors as the oracle."""
    d = medium
    k, lam, t, T = 8, 0.05, 4, 3
    W0 = mfx.initial_col(k, d.rows)
    Wr, Hr, rmse_ref, trace_ref, done_ref = orc.ccdr1_ext(d, W0, k, lam, t, T, 2, eps=eps)
    _, _, rmse_plain, _, _ = orc.ccdr1_ext(d, W0, k, lam, t, T, 2)
    assert np.abs(rmse_plain - rmse_ref).max() > 1e-6  # the rule fires on this input
    if eps > 0.5:
        assert done_ref.min() < k  # ... and cuts the rank loop short
    s = mfx.CcdSolver(d, mfx.test_data_of(d), _ext_params(mfx, k, lam, t, T, schedule, variant, panel_rows, eps=eps, verbose=1, do_predict=1))
    s.set_factors(W0.copy())
    rep = s.iterate(t)
    trace, secs, done = s.rank_trace(t, k)
    W, H = s.get_factors()
    s.close()
    assert list(done) == list(done_ref), (done, done_ref)
    assert np.array_equal(np.isnan(trace), np.isnan(trace_ref))
    assert np.nanmax(np.abs(trace - trace_ref)) < 1e-4
    assert np.all(np.abs(np.array([r.rmse for r in rep]) - rmse_ref) < 1e-4)
    scale = float(max(np.abs(Wr).max(), np.abs(Hr).max()))
    assert np.abs(W - Wr).max() < 2e-3 * scale and np.abs(H - Hr).max() < 2e-3 * scale
