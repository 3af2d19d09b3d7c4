"""Full BASELINE.json size (configs[2]: 480189 x 17770, Z = 99 072 112, k = 64) on the GPU, checked
through size-independent properties (the CPU oracle would need minutes here):

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
