"""ctypes front end of oracle/liboracle.so -- the CPU oracle.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module
(see oracle/mf_oracle.h).  It never touches /root/reference at run time.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_u32p = C.POINTER(C.c_uint32)
_f32p = C.POINTER(C.c_float)
_f64p = C.POINTER(C.c_double)


def build() -> None:
    """(Re)builds liboracle.so and, where /root/reference exists, oracle/_ref/ref_cpu."""
    subprocess.run(["make", "-s", "-C", _HERE], check=True)


def lib() -> C.CDLL:
    global _LIB
    if _LIB is None:
        os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")  # no spinning barriers on a shared host
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
        _LIB.orc_update_rating.restype = C.c_float
        _LIB.orc_calrmse.restype = C.c_double
        _LIB.orc_max_threads.restype = C.c_int
    return _LIB


def _p(a, ty):
    return a.ctypes.data_as(ty) if a is not None else None


def _chk(a, dt):
    assert a.dtype == dt and a.flags["C_CONTIGUOUS"], (a.dtype, dt)
    return a


def _usable_cores() -> int:
    """Cores this process can really run on: affinity mask capped by the cgroup CPU quota.  A GPU
    box shows every logical CPU of the host (128-256) but grants one GPU's share of them;
    starting that many OpenMP threads with spinning barriers makes the oracle crawl."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(float(q) / float(p))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except Exception:
            pass
    return max(1, n)


def max_threads() -> int:
    """Threads to run the oracle with: min(OpenMP's idea, usable cores, 32)."""
    return max(1, min(int(lib().orc_max_threads()), _usable_cores(), 32))


def initial_col(k: int, n: int) -> np.ndarray:
    """reference: initial_col(X, k, n) -> flat [k][n]."""
    x = np.empty((k, n), np.float32)
    lib().orc_initial_col(_p(x, _f32p), C.c_long(k), C.c_long(n))
    return x


def rank_one_sweep(ptr, idx, val, vec, lam, threads=1) -> np.ndarray:
    nseg = ptr.shape[0] - 1
    out = np.empty(nseg, np.float32)
    lib().orc_rank_one_sweep(C.c_long(nseg), _p(_chk(ptr, np.uint32), _u32p), _p(_chk(idx, np.uint32), _u32p),
                             _p(_chk(val, np.float32), _f32p), _p(_chk(vec, np.float32), _f32p),
                             C.c_float(lam), _p(out, _f32p), C.c_int(threads))
    return out


def update_rating(ptr, idx, val, gathered, per_seg, add, threads=1) -> None:
    """In place on `val`."""
    nseg = ptr.shape[0] - 1
    lib().orc_update_rating(C.c_long(nseg), _p(_chk(ptr, np.uint32), _u32p), _p(_chk(idx, np.uint32), _u32p),
                            _p(_chk(val, np.float32), _f32p), _p(_chk(gathered, np.float32), _f32p),
                            _p(_chk(per_seg, np.float32), _f32p), C.c_int(1 if add else 0), C.c_int(threads))


def calrmse(d, W, H, k, als) -> float:
    if d.nnz_test == 0:
        return 0.0
    return float(lib().orc_calrmse(C.c_long(d.nnz_test), _p(d.test_row, _u32p), _p(d.test_col, _u32p),
                                   _p(d.test_val, _f32p), _p(_chk(W, np.float32), _f32p),
                                   _p(_chk(H, np.float32), _f32p), C.c_long(d.rows), C.c_long(d.cols),
                                   C.c_long(k), C.c_int(1 if als else 0)))


def ccdr1(d, W, k, lam, maxiter, maxinner, threads=1):
    """Runs the restated ccdr1_OMP on a COPY of d's values.  W: [k][rows] initial factors.
    Returns (W, H, rmse[maxiter], times[maxiter,2], csc_val_residual, csr_val_residual)."""
    W = np.array(W, np.float32, copy=True, order="C").reshape(k, d.rows)
    H = np.zeros((k, d.cols), np.float32)
    csc_val, csr_val = d.csc_val.copy(), d.csr_val.copy()
    rmse = np.zeros(maxiter, np.float64)
    times = np.zeros((maxiter, 2), np.float64)
    lib().orc_ccdr1(C.c_long(d.rows), C.c_long(d.cols), C.c_long(d.nnz),
                    _p(d.csc_col_ptr, _u32p), _p(d.csc_row_idx, _u32p), _p(csc_val, _f32p),
                    _p(d.csr_row_ptr, _u32p), _p(d.csr_col_idx, _u32p), _p(csr_val, _f32p),
                    _p(W, _f32p), _p(H, _f32p), C.c_long(k), C.c_float(lam), C.c_int(maxiter),
                    C.c_int(maxinner), C.c_int(threads), C.c_long(d.nnz_test),
                    _p(d.test_row, _u32p), _p(d.test_col, _u32p), _p(d.test_val, _f32p),
                    _p(rmse, _f64p), _p(times, _f64p))
    return W, H, rmse, times, csc_val, csr_val


def ccdr1_ext(d, W, k, lam, maxiter, maxinner, threads=1, do_nmf=0, eps=0.0):
    """ccdr1 with do_nmf / eps (LIBPMF 1.41 meaning; parity unpinned) and the per-rank calrmse_r1 trace (pinned:
    src/tools.cpp:261-270).  Returns (W, H, rmse[maxiter], rank_rmse[maxiter, k], ranks_done[maxiter])."""
    W = np.array(W, np.float32, copy=True, order="C").reshape(k, d.rows)
    H = np.zeros((k, d.cols), np.float32)
    csc_val, csr_val = d.csc_val.copy(), d.csr_val.copy()
    rmse = np.zeros(maxiter, np.float64)
    rank_rmse = np.full((maxiter, k), np.nan, np.float64)
    done = np.zeros(maxiter, np.int32)
    lib().orc_ccdr1_ext(C.c_long(d.rows), C.c_long(d.cols),
                        _p(d.csc_col_ptr, _u32p), _p(d.csc_row_idx, _u32p), _p(csc_val, _f32p),
                        _p(d.csr_row_ptr, _u32p), _p(d.csr_col_idx, _u32p), _p(csr_val, _f32p),
                        _p(W, _f32p), _p(H, _f32p), C.c_long(k), C.c_float(lam), C.c_int(maxiter),
                        C.c_int(maxinner), C.c_int(threads), C.c_long(d.nnz_test),
                        _p(d.test_row, _u32p), _p(d.test_col, _u32p), _p(d.test_val, _f32p),
                        C.c_int(do_nmf), C.c_float(eps), _p(rmse, _f64p), _p(rank_rmse, _f64p),
                        done.ctypes.data_as(C.POINTER(C.c_int)))
    return W, H, rmse, rank_rmse, done


def gramian(idx, X, k) -> np.ndarray:
    A = np.empty((k, k), np.float32)
    lib().orc_gramian(C.c_long(idx.shape[0]), _p(_chk(idx, np.uint32), _u32p), _p(_chk(X, np.float32), _f32p),
                      C.c_long(k), _p(A, _f32p))
    return A


def chol_inverse(A) -> np.ndarray:
    A = np.array(A, np.float32, copy=True, order="C")
    lib().orc_chol_inverse(C.c_long(A.shape[0]), _p(A, _f32p))
    return A


def als_half(ptr, idx, val, X, k, lam, threads=1) -> np.ndarray:
    nseg = ptr.shape[0] - 1
    Y = np.empty((nseg, k), np.float32)
    lib().orc_als_half(C.c_long(nseg), _p(_chk(ptr, np.uint32), _u32p), _p(_chk(idx, np.uint32), _u32p),
                       _p(_chk(val, np.float32), _f32p), _p(_chk(X, np.float32), _f32p), _p(Y, _f32p),
                       C.c_long(k), C.c_float(lam), C.c_int(threads))
    return Y


def als(d, H, k, lam, maxiter, threads=1):
    """Restated ALS_OMP.  H: [cols][k] initial.  Returns (W, H, rmse, times)."""
    H = np.array(H, np.float32, copy=True, order="C").reshape(d.cols, k)
    W = np.zeros((d.rows, k), np.float32)
    rmse = np.zeros(maxiter, np.float64)
    times = np.zeros(maxiter, np.float64)
    lib().orc_als(C.c_long(d.rows), C.c_long(d.cols), C.c_long(d.nnz),
                  _p(d.csc_col_ptr, _u32p), _p(d.csc_row_idx, _u32p), _p(d.csc_val, _f32p),
                  _p(d.csr_row_ptr, _u32p), _p(d.csr_col_idx, _u32p), _p(d.csr_val, _f32p),
                  _p(W, _f32p), _p(H, _f32p), C.c_long(k), C.c_float(lam), C.c_int(maxiter),
                  C.c_int(threads), C.c_long(d.nnz_test), _p(d.test_row, _u32p), _p(d.test_col, _u32p),
                  _p(d.test_val, _f32p), _p(rmse, _f64p), _p(times, _f64p))
    return W, H, rmse, times
