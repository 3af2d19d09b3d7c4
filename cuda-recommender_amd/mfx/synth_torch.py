"""Device-side synthetic rating generator (torch is used here only as device-memory plumbing):
builds a Netflix-shaped dual CSR+CSC matrix directly in HBM so that bench.py and the full-size
property tests never push gigabytes through PCIe.  Same model as dataset.synth_ratings
(planted low rank + noise, clipped to [1, 5]) with log-normal user activity / item popularity.

Index tensors are int32 (bit-identical to the uint32 the C ABI expects while nnz < 2^31).
"""
from __future__ import annotations

import math

import torch


def synth_ratings_device(rows: int, cols: int, nnz: int, seed: int = 1234, device="cuda:0",
                         sigma_rows: float = 1.2, sigma_cols: float = 1.8, planted_rank: int = 8,
                         noise: float = 0.1, test_frac: float = 0.01, row_lo: int = 0, row_hi: int = -1,
                         shard=None):
    """Returns a dict of tensors on `device`: rows, cols, csr_row_ptr, csr_col_idx, csr_val,
    csc_col_ptr, csc_row_idx, csc_val, test_row, test_col, test_val, csc_of_csr (for every CSC
    position the CSR position of the same rating).  row_lo/row_hi keep only that row block
    (row ids rebased), which is how a multi-GPU shard is generated in place: every rank draws the
    same global matrix from the same seed and keeps its rows.  shard = (g, G) picks block g of the G
    nnz-balanced contiguous row blocks (the rule of mfx_partition_rows: block g starts at the first row
    whose prefix of non-zeros reaches g/G of the total) instead of an explicit row_lo/row_hi."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    dev = torch.device(device)

    def cdf(n, sigma):
        w = torch.exp(sigma * torch.randn(n, generator=g, device=dev, dtype=torch.float64))
        c = torch.cumsum(w / w.sum(), 0)
        c[-1] = 1.0
        return c

    cr, cc = cdf(rows, sigma_rows), cdf(cols, sigma_cols)
    ntest = int(round(nnz * test_frac))
    want = nnz + ntest
    keys = torch.zeros(0, dtype=torch.int64, device=dev)
    while keys.numel() < want:
        need = int((want - keys.numel()) * 1.25) + 1024
        i = torch.searchsorted(cr, torch.rand(need, generator=g, device=dev, dtype=torch.float64)).clamp_(max=rows - 1)
        j = torch.searchsorted(cc, torch.rand(need, generator=g, device=dev, dtype=torch.float64)).clamp_(max=cols - 1)
        keys = torch.unique(torch.cat([keys, i * cols + j]))
        del i, j
    perm = torch.randperm(keys.numel(), generator=g, device=dev)[:want]
    keys = keys[perm]
    del perm
    ws = torch.randn(rows, planted_rank, generator=g, device=dev)
    hs = torch.randn(cols, planted_rank, generator=g, device=dev)

    def values(k):
        i, j = k // cols, k % cols
        out = torch.empty(k.numel(), device=dev, dtype=torch.float32)
        step = 1 << 24
        for s in range(0, k.numel(), step):
            e = min(k.numel(), s + step)
            raw = (ws[i[s:e]] * hs[j[s:e]]).sum(1) / math.sqrt(planted_rank)
            out[s:e] = (3.0 + raw + noise * torch.randn(e - s, generator=g, device=dev)).clamp_(1.0, 5.0)
        return i, j, out

    test_keys, train_keys = keys[:ntest], torch.sort(keys[ntest:]).values  # CSR order = sorted keys
    del keys
    ti, tj, tv = values(test_keys)
    ri, rj, rv = values(train_keys)
    del train_keys, test_keys, ws, hs
    if shard is not None and shard[1] > 1:
        g_id, g_cnt = int(shard[0]), int(shard[1])
        ptr = torch.zeros(rows + 1, dtype=torch.int64, device=dev)
        ptr[1:] = torch.cumsum(torch.bincount(ri, minlength=rows), 0)
        total = int(ptr[-1])
        tgt = torch.tensor([(total * q + g_cnt - 1) // g_cnt for q in (g_id, g_id + 1)], dtype=torch.int64, device=dev)
        b = torch.searchsorted(ptr, tgt)  # lower bound, like std::lower_bound over csr_row_ptr
        row_lo = 0 if g_id == 0 else int(b[0])
        row_hi = rows if g_id == g_cnt - 1 else int(b[1])
        del ptr
    if row_hi < 0:
        row_hi = rows
    if row_lo != 0 or row_hi != rows:
        keep = (ri >= row_lo) & (ri < row_hi)
        ri, rj, rv = ri[keep] - row_lo, rj[keep], rv[keep]
        keep = (ti >= row_lo) & (ti < row_hi)
        ti, tj, tv = ti[keep] - row_lo, tj[keep], tv[keep]
        del keep
    lrows = row_hi - row_lo
    csr_ptr = torch.zeros(lrows + 1, dtype=torch.int64, device=dev)
    csr_ptr[1:] = torch.cumsum(torch.bincount(ri, minlength=lrows), 0)
    csc_ptr = torch.zeros(cols + 1, dtype=torch.int64, device=dev)
    csc_ptr[1:] = torch.cumsum(torch.bincount(rj, minlength=cols), 0)
    order = torch.sort(rj * lrows + ri).indices  # CSC order: by column, then row
    i32 = lambda t: t.to(torch.int32).contiguous()
    return dict(rows=lrows, cols=cols,
                csr_row_ptr=i32(csr_ptr), csr_col_idx=i32(rj), csr_val=rv.contiguous(),
                csc_col_ptr=i32(csc_ptr), csc_row_idx=i32(ri[order]), csc_val=rv[order].contiguous(),
                test_row=i32(ti), test_col=i32(tj), test_val=tv.contiguous(), csc_of_csr=order)


def leading_row_block(d, nblocks: int):
    """The first of `nblocks` nnz-balanced contiguous blocks of user rows of the device matrix `d` (the rule of
    mfx_partition_rows), as a device dict of the same form: local CSR = a prefix of the CSR arrays, local CSC = the
    entries of every column whose row falls in the block, in the column's own order."""
    ptr = d["csr_row_ptr"].long()
    total = int(ptr[-1])
    r = int(torch.searchsorted(ptr, torch.tensor([(total + nblocks - 1) // nblocks], dtype=torch.int64, device=ptr.device))[0])
    r = max(1, min(r, int(d["rows"])))
    z = int(ptr[r])
    keep = d["csc_row_idx"] < r
    cols = int(d["cols"])
    csc_ptr = torch.zeros(cols + 1, dtype=torch.int64, device=ptr.device)
    csc_ptr[1:] = torch.cumsum(torch.bincount(d["csr_col_idx"][:z].long(), minlength=cols), 0)
    tkeep = d["test_row"] < r
    i32 = lambda t: t.to(torch.int32).contiguous()
    return dict(rows=r, cols=cols, csr_row_ptr=d["csr_row_ptr"][:r + 1].contiguous(), csr_col_idx=d["csr_col_idx"][:z].contiguous(),
                csr_val=d["csr_val"][:z].contiguous(), csc_col_ptr=i32(csc_ptr), csc_row_idx=d["csc_row_idx"][keep].contiguous(),
                csc_val=d["csc_val"][keep].contiguous(), test_row=d["test_row"][tkeep].contiguous(),
                test_col=d["test_col"][tkeep].contiguous(), test_val=d["test_val"][tkeep].contiguous())


def to_rating_data(d):
    """Device dict -> host RatingData (numpy), for the CPU-baseline leg and parity checks."""
    import numpy as np
    from .dataset import RatingData
    u32 = lambda t: np.ascontiguousarray(t.cpu().numpy().view(np.uint32))
    f32 = lambda t: np.ascontiguousarray(t.cpu().numpy())
    return RatingData(int(d["rows"]), int(d["cols"]), u32(d["csr_row_ptr"]), u32(d["csr_col_idx"]), f32(d["csr_val"]),
                      u32(d["csc_col_ptr"]), u32(d["csc_row_idx"]), f32(d["csc_val"]),
                      u32(d["test_row"]), u32(d["test_col"]), f32(d["test_val"]))
