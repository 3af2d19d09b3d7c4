"""Host-side mirror of the reference's solver interface over libmfx.so's C ABI.

Same names, argument order and error behaviour as the reference's driver-facing API
(reference: cuda_src/CCD_CUDA.h:49, cuda_src/ALS_CUDA.h:40, src/pmf.h:8-43, src/tools.h,
src/extras.h) so that tests read like the reference's own call sites (src/main.cpp:86-141).
Everything numerical happens inside libmfx.so on the GPU; nothing here computes on the CPU
except the reporting helpers the reference also runs on the host (calculate_rmse_directly,
golden_compare).
"""
from __future__ import annotations

import ctypes as C
import sys
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np

from . import _lib as L
from .dataset import RatingData


# ---------------------------------------------------------------------------------------------
# parameter / command line (reference: src/pmf.h:8-43, src/extras.cpp:46-141)
# ---------------------------------------------------------------------------------------------
class solvertype:
    CCD = 0
    ALS = 1


class parameter:
    """Field-for-field the reference's `parameter` with its defaults (src/pmf.h:26-42), plus the
    knobs this implementation adds (schedule, kernel_variant, device, profile)."""

    def __init__(self):
        self.solver_type = solvertype.CCD
        self.k = 10
        self.threads = 4
        self.maxiter = 5
        self.maxinneriter = 1
        self.lambda_ = 0.1
        self.eps = 1e-3
        self.do_predict = 0
        self.verbose = 0
        self.do_nmf = 0
        self.enable_cuda = False
        self.enable_omp = False
        self.nBlocks = 32
        self.nThreadsPerBlock = 256
        self.src_dir = "../data/simple"
        # additions
        self.device = 0
        self.schedule = 1
        self.kernel_variant = 1
        self.profile = 0
        self.tiles_per_span = 0
        self.panel_rows = 0
        self.wg_waves = 0
        self.graph = 0
        self.layout_build = 0
        self.log = 0  # print the reference's per-iteration "[-INFO-]" line
        # opt-in: give eps / do_nmf / (verbose and do_predict) their LIBPMF meaning.  The reference parses them and
        # reads none (src/pmf.h:33-36), so by default they are ignored here too.
        self.libpmf_flags = 0

    def to_c(self) -> L.mfx_params:
        p = L.mfx_params()
        L.lib().mfx_params_default(C.byref(p))
        p.k, p.lambda_, p.maxiter, p.maxinneriter = int(self.k), float(self.lambda_), int(self.maxiter), int(self.maxinneriter)
        p.nBlocks, p.nThreadsPerBlock = int(self.nBlocks), int(self.nThreadsPerBlock)
        p.verbose, p.device, p.schedule = int(self.log), int(self.device), int(self.schedule)
        p.kernel_variant, p.profile, p.tiles_per_span = int(self.kernel_variant), int(self.profile), int(self.tiles_per_span)
        p.panel_rows, p.wg_waves, p.graph = int(self.panel_rows), int(self.wg_waves), int(self.graph)
        p.layout_build = int(self.layout_build)
        if self.libpmf_flags:
            p.do_nmf, p.eps = int(self.do_nmf), float(self.eps)
            p.rank_trace = 1 if (self.verbose and self.do_predict) else 0
        return p


HELP = """Usage: omp-pmf-train [options] data_dir [model_filename]
options:
    -k rank : set the rank (default 10)
    -n threads : set the number of threads (default 4)
    -l lambda : set the regularization parameter lambda (default 0.1)
    -t max_iter: set the number of iterations (default 5)
    -T max_inner_iter: set the number of inner iterations used in CCDR1 (default 5)
    -e epsilon : set inner termination criterion epsilon of CCDR1 (default 1e-3)
    -p do_predict: do prediction or not (default 0)
    -q verbose: show information or not (default 0)
    -N do_nmf: do nmf (default 0)
    -CUDA: Flag to enable CUDA
    -nBlocks: Number of blocks on CUDA (default 32)
    -nThreadsPerBlock: Number of threads per block on CUDA (default 256)
    -ALS: Flag to enable ALS algorithm, if not present CCD++ is used
"""


class UsageError(SystemExit):
    """Raised where the reference calls exit_with_help() (prints HELP, exit status 1)."""


def parse_command_line(argv: Sequence[str]) -> parameter:
    """argv[0] is the program name.  Reproduces the reference's scanner including its quirk:
    every token starting with '-' pre-consumes the next argv, valueless flags (-CUDA, -OMP,
    -ALS) hand it back, so a valueless flag as the LAST argv is a usage error
    (src/extras.cpp:72-91)."""
    def usage():
        sys.stdout.write(HELP)
        raise UsageError(1)

    param = parameter()
    argc = len(argv)
    i = 1
    while i < argc:
        if not argv[i].startswith("-"):
            break
        i += 1
        if i >= argc:
            usage()
        flag, val = argv[i - 1], argv[i]
        if flag == "-nBlocks":
            param.nBlocks = _atoi(val)
        elif flag == "-nThreadsPerBlock":
            param.nThreadsPerBlock = _atoi(val)
        elif flag == "-CUDA":
            param.enable_cuda = True; i -= 1
        elif flag == "-OMP":
            param.enable_omp = True; i -= 1
        elif flag == "-ALS":
            param.solver_type = solvertype.ALS; i -= 1
        else:
            c = flag[1:2]
            if c == "k": param.k = _atoi(val)
            elif c == "n": param.threads = _atoi(val)
            elif c == "l": param.lambda_ = _atof(val)
            elif c == "t": param.maxiter = _atoi(val)
            elif c == "T": param.maxinneriter = _atoi(val)
            elif c == "e": param.eps = _atof(val)
            elif c == "p": param.do_predict = _atoi(val)
            elif c == "q": param.verbose = _atoi(val)
            elif c == "N": param.do_nmf = 1 if _atoi(val) == 1 else 0
            else:
                sys.stderr.write(f"unknown option: -{c}\n")
                usage()
        i += 1
    if param.do_predict != 0:
        param.verbose = 1
    if i >= argc:
        usage()
    param.src_dir = argv[i][:1023]
    return param


def _atoi(s: str) -> int:
    import re
    m = re.match(r"\s*[+-]?\d+", s)
    return int(m.group(0)) if m else 0


def _atof(s: str) -> float:
    import re
    m = re.match(r"\s*[+-]?(\d+\.?\d*([eE][+-]?\d+)?|\.\d+([eE][+-]?\d+)?)", s)
    return float(m.group(0)) if m else 0.0


# ---------------------------------------------------------------------------------------------
# data views
# ---------------------------------------------------------------------------------------------
@dataclass
class TestData:
    """reference: TestData (src/pmf_util.h:151-211)."""
    rows: int
    cols: int
    test_row: np.ndarray
    test_col: np.ndarray
    test_val: np.ndarray

    @property
    def nnz(self) -> int:
        return int(self.test_val.shape[0])


def test_data_of(d: RatingData) -> TestData:
    return TestData(d.rows, d.cols, d.test_row, d.test_col, d.test_val)


def _vp(a: Optional[np.ndarray]):
    return a.ctypes.data_as(C.c_void_p) if a is not None and a.size else None


def _csx(R: RatingData) -> L.mfx_csx:
    R.check_types()
    return L.mfx_csx(R.rows, R.cols, R.nnz, _vp(R.csc_col_ptr), _vp(R.csc_row_idx), _vp(R.csc_val),
                     _vp(R.csr_row_ptr), _vp(R.csr_col_idx), _vp(R.csr_val))


def _coo(T) -> L.mfx_coo:
    if T is None:
        return L.mfx_coo(0, None, None, None)
    for a, dt in ((T.test_row, np.uint32), (T.test_col, np.uint32), (T.test_val, np.float32)):
        assert a.dtype == dt and a.flags["C_CONTIGUOUS"]
    return L.mfx_coo(int(T.test_val.shape[0]), _vp(T.test_row), _vp(T.test_col), _vp(T.test_val))


def _f32c(a, shape=None) -> np.ndarray:
    assert isinstance(a, np.ndarray) and a.dtype == np.float32 and a.flags["C_CONTIGUOUS"], "need C-contiguous float32"
    if shape is not None:
        assert a.shape == tuple(shape), (a.shape, shape)
    return a


# ---------------------------------------------------------------------------------------------
# host helpers on the path
# ---------------------------------------------------------------------------------------------
def device_count() -> int:
    return int(L.lib().mfx_device_count())


def initial_col(k: int, n: int) -> np.ndarray:
    """reference: initial_col(X, k, n) (src/tools.cpp:165-173) -> [k][n], glibc rand(), seed 0."""
    X = np.empty((k, n), np.float32)
    L.lib().mfx_initial_col(X.ctypes.data_as(L.f32p), k, n)
    return X


def calculate_rmse_directly(W: np.ndarray, H: np.ndarray, T, rank: int, ifALS: bool, quiet: bool = False) -> float:
    """reference: src/extras.cpp:182-216 (host-side, fp64 accumulation of fp32 products)."""
    import time
    t0 = time.time()
    if T.nnz == 0:
        raise SystemExit(1)  # the reference exits when there are no test instances (:212)
    i, j = T.test_row.astype(np.int64), T.test_col.astype(np.int64)
    acc = np.zeros(T.nnz, np.float64)
    for t in range(rank):
        a = W[i, t] if ifALS else W[t, i]
        b = H[j, t] if ifALS else H[t, j]
        acc += (a * b).astype(np.float64)
    rmse = float(np.sqrt(np.sum((acc - T.test_val.astype(np.float64)) ** 2) / T.nnz))
    if not quiet:
        print("Test RMSE = %f. Calculated in %fs" % (rmse, time.time() - t0))
    return rmse


def golden_compare(W: np.ndarray, W_ref: np.ndarray, k: int, m: int, quiet: bool = False) -> int:
    """reference: src/extras.cpp:218-238: counts entries with |a-b| > 0.1*|b|; returns the count."""
    a, b = W.reshape(k, m).astype(np.float64), W_ref.reshape(k, m).astype(np.float64)
    errors = int(np.count_nonzero(np.abs(a - b) > 0.1 * np.abs(b)))
    if not quiet:
        if errors == 0:
            print("Check... PASS!")
        else:
            print("Check... NO PASS! [%.4f%%] #Error = %u out of %u entries." % (100.0 * errors / (k * m), errors, k * m))
    return errors


# ---------------------------------------------------------------------------------------------
# the two drop-in entry points
# ---------------------------------------------------------------------------------------------
def kernel_wrapper_ccdpp_NV(R: RatingData, T, W: np.ndarray, H: np.ndarray, parameters: parameter) -> List[L.mfx_iter_report]:
    """reference: kernel_wrapper_ccdpp_NV(SparseMatrix&, TestData&, MatData& W, MatData& H,
    parameter&) (cuda_src/CCD_CUDA.cu:164-179).  W [k][rows] initialised by the caller, H [k][cols]
    (content ignored, CCD++ starts from 0); both overwritten in place.  Like the reference it
    does not raise on a device failure: it prints "CCD FAILED: ..." and returns."""
    k = int(parameters.k)
    _f32c(W, (k, R.rows)); _f32c(H, (k, R.cols))
    reports = (L.mfx_iter_report * max(1, int(parameters.maxiter)))()
    csx, coo, p = _csx(R), _coo(T), parameters.to_c()
    rc = L.lib().mfx_ccdpp_run(C.byref(csx), C.byref(coo), _vp(W), _vp(H), C.byref(p), reports)
    kernel_wrapper_ccdpp_NV.last_status = rc
    return list(reports)[: int(parameters.maxiter)]


def kernel_wrapper_als_NV(R: RatingData, T, W: np.ndarray, H: np.ndarray, parameters: parameter) -> List[L.mfx_iter_report]:
    """reference: kernel_wrapper_als_NV (cuda_src/ALS_CUDA.cu:183-198).  W [rows][k], H [cols][k]."""
    k = int(parameters.k)
    _f32c(W, (R.rows, k)); _f32c(H, (R.cols, k))
    reports = (L.mfx_iter_report * max(1, int(parameters.maxiter)))()
    csx, coo, p = _csx(R), _coo(T), parameters.to_c()
    rc = L.lib().mfx_als_run(C.byref(csx), C.byref(coo), _vp(W), _vp(H), C.byref(p), reports)
    kernel_wrapper_als_NV.last_status = rc
    return list(reports)[: int(parameters.maxiter)]


# ---------------------------------------------------------------------------------------------
# resident solvers
# ---------------------------------------------------------------------------------------------
def _kernel_times(fn, handle):
    cap = 32
    names = (C.c_char_p * cap)()
    secs = (C.c_double * cap)()
    cnt = (C.c_int64 * cap)()
    n = fn(handle, cap, names, secs, cnt)
    return {names[i].decode(): (secs[i], cnt[i]) for i in range(n)}


class Comm:
    """RCCL communicator (one process per GPU).  `uid` is the 128-byte id from Comm.unique_id()
    on rank 0, shipped to the other ranks by the caller."""

    def __init__(self, uid: Optional[bytes], rank: int, nranks: int, device: int, local_group: Optional[int] = None):
        self.handle = C.c_void_p()
        if local_group is not None:  # in-process loopback group (threads of one process), see mfx.h
            L.check(L.lib().mfx_comm_create_local(C.byref(self.handle), int(local_group), rank, nranks, device))
        else:
            buf = C.create_string_buffer(bytes(uid), L.MFX_COMM_ID_BYTES)
            L.check(L.lib().mfx_comm_create(C.byref(self.handle), buf, rank, nranks, device))
        self.rank, self.nranks = rank, nranks

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(L.MFX_COMM_ID_BYTES)
        L.check(L.lib().mfx_comm_unique_id(buf))
        return buf.raw

    def agree(self, local_status: int = 0) -> int:
        """Collective: the worst status over all ranks (mfx_comm_agree).  Call it once after every rank's
        setup, failed ranks included, before the first iterate."""
        out = C.c_int(0)
        L.check(L.lib().mfx_comm_agree(self.handle, int(local_status), C.byref(out)))
        return int(out.value)

    def abort(self):
        """Releases the ranks waiting for this one after a local failure (mfx_comm_abort)."""
        if self.handle:
            L.lib().mfx_comm_abort(self.handle)

    def close(self):
        if self.handle:
            L.lib().mfx_comm_destroy(self.handle)
            self.handle = C.c_void_p()


class CcdSolver:
    """Resident CCD++ (mfx_ccd_*).  Arrays may be numpy (host) or anything exposing
    `data_ptr()` (torch CUDA tensors, space=device)."""

    def __init__(self, R, T, parameters: parameter, comm: Optional[Comm] = None,
                 global_col_nnz=None, global_test_nnz: int = 0, device_arrays: Optional[dict] = None):
        self.p = parameters
        self.handle = C.c_void_p()
        self._keep = []
        if device_arrays is not None:
            d = device_arrays
            ptr = lambda t: C.c_void_p(int(t.data_ptr())) if t is not None and t.numel() else None
            self.rows, self.cols, nnz = int(d["rows"]), int(d["cols"]), int(d["csr_val"].numel())
            csx = L.mfx_csx(self.rows, self.cols, nnz, ptr(d["csc_col_ptr"]), ptr(d["csc_row_idx"]), ptr(d["csc_val"]),
                            ptr(d["csr_row_ptr"]), ptr(d["csr_col_idx"]), ptr(d["csr_val"]))
            tv = d.get("test_val")
            coo = L.mfx_coo(int(tv.numel()) if tv is not None else 0, ptr(d.get("test_row")), ptr(d.get("test_col")), ptr(tv))
            space = L.MFX_DEVICE
            gcn = ptr(global_col_nnz) if global_col_nnz is not None else None
        else:
            self.rows, self.cols = R.rows, R.cols
            csx, coo, space = _csx(R), _coo(T), L.MFX_HOST
            gcn = _vp(global_col_nnz) if global_col_nnz is not None else None
        shard = None
        if comm is not None:
            shard = L.mfx_shard(comm.handle, gcn, int(global_test_nnz))
        cp = parameters.to_c()
        L.check(L.lib().mfx_ccd_create(C.byref(self.handle), C.byref(csx), C.byref(coo), C.byref(cp), space,
                                       C.byref(shard) if shard is not None else None))
        self.k = int(parameters.k)

    def set_factors(self, W):
        if hasattr(W, "data_ptr"):
            L.check(L.lib().mfx_ccd_set_factors(self.handle, C.c_void_p(int(W.data_ptr())), None, L.MFX_DEVICE))
        else:
            _f32c(W, (self.k, self.rows))
            L.check(L.lib().mfx_ccd_set_factors(self.handle, _vp(W), None, L.MFX_HOST))

    def iterate(self, n_outer: int, with_rmse: bool = True) -> List[L.mfx_iter_report]:
        reports = (L.mfx_iter_report * max(1, n_outer))()
        L.check(L.lib().mfx_ccd_iterate(self.handle, n_outer, 1 if with_rmse else 0, reports))
        return list(reports)[:n_outer]

    def get_factors(self):
        W = np.empty((self.k, self.rows), np.float32)
        H = np.empty((self.k, self.cols), np.float32)
        L.check(L.lib().mfx_ccd_get_factors(self.handle, _vp(W), _vp(H), L.MFX_HOST))
        return W, H

    def get_residual(self, nnz: int):
        a, b = np.empty(nnz, np.float32), np.empty(nnz, np.float32)
        L.check(L.lib().mfx_ccd_get_residual(self.handle, _vp(a), _vp(b)))
        return a, b

    def rank_trace(self, n_outer: int, k: int):
        """(rmse[n_outer, k], seconds[n_outer, k], ranks_done[n_outer]) of the last iterate() call (parameter.libpmf_flags
        with verbose and do_predict: calrmse_r1 after every rank; NaN where the eps rule skipped a rank)."""
        rm = np.full((n_outer, k), np.nan, np.float64)
        sec = np.zeros((n_outer, k), np.float64)
        done = np.zeros(n_outer, np.int32)
        n = L.lib().mfx_ccd_rank_trace(self.handle, n_outer * k, rm.ctypes.data_as(L.f64p), sec.ctypes.data_as(L.f64p), n_outer,
                                       done.ctypes.data_as(C.POINTER(C.c_int32)))
        L.check(min(n, 0))
        return rm[:n], sec[:n], done[:n]

    def set_profile(self, on: bool):
        L.check(L.lib().mfx_ccd_set_profile(self.handle, 1 if on else 0))

    def layout_info(self):
        """{"csc": {...}, "csr": {...}}: panels, entries per panel, kind ("lds" / "cache" / "plain" / "scatter" / "scatter32": 32-bit segment ids), tiles per span."""
        out = {}
        for side, name in ((0, "csc"), (1, "csr")):
            v = (C.c_int32 * 4)()
            L.check(L.lib().mfx_ccd_layout_info(self.handle, side, v))
            out[name] = {"panels": int(v[0]), "panel_rows": int(v[1]),
                         "kind": "scatter" if v[2] == 2 else "scatter32" if v[2] == 3 else "lds" if v[2] else ("cache" if v[1] else "plain"),
                         "tiles_per_span": int(v[3])}  # kind "tile": panel_rows = slice entries, tiles_per_span = segments per block
        return out

    def kernel_times(self):
        return _kernel_times(L.lib().mfx_ccd_kernel_times, self.handle)

    def close(self):
        if self.handle:
            L.lib().mfx_ccd_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class AlsSolver:
    """Resident ALS (mfx_als_*)."""

    def __init__(self, R: RatingData, T, parameters: parameter, comm: Optional[Comm] = None,
                 row_range=None, col_range=None, device_arrays: Optional[dict] = None):
        """With `comm`: rank-local ALS shard of the GLOBAL matrix R -- this rank solves user rows
        row_range for the W-half and item columns col_range for the H-half (mfx_als_create_sharded).
        device_arrays: the matrix as a dict of device tensors (mfx.synth_torch), single GPU only."""
        self.handle = C.c_void_p()
        cp = parameters.to_c()
        if device_arrays is not None:
            assert comm is None, "device-resident inputs: single-GPU ALS only"
            d = device_arrays
            ptr = lambda t: C.c_void_p(int(t.data_ptr())) if t is not None and t.numel() else None
            self.rows, self.cols, self.k = int(d["rows"]), int(d["cols"]), int(parameters.k)
            csx = L.mfx_csx(self.rows, self.cols, int(d["csr_val"].numel()), ptr(d["csc_col_ptr"]), ptr(d["csc_row_idx"]), ptr(d["csc_val"]),
                            ptr(d["csr_row_ptr"]), ptr(d["csr_col_idx"]), ptr(d["csr_val"]))
            tv = d.get("test_val")
            coo = L.mfx_coo(int(tv.numel()) if tv is not None else 0, ptr(d.get("test_row")), ptr(d.get("test_col")), ptr(tv))
            L.check(L.lib().mfx_als_create(C.byref(self.handle), C.byref(csx), C.byref(coo), C.byref(cp), L.MFX_DEVICE))
            return
        self.rows, self.cols, self.k = R.rows, R.cols, int(parameters.k)
        if comm is None:
            csx, coo = _csx(R), _coo(T)
            L.check(L.lib().mfx_als_create(C.byref(self.handle), C.byref(csx), C.byref(coo), C.byref(cp), L.MFX_HOST))
            return
        (rl, rh), (cl, ch) = row_range, col_range
        a, b = int(R.csr_row_ptr[rl]), int(R.csr_row_ptr[rh])
        c, d = int(R.csc_col_ptr[cl]), int(R.csc_col_ptr[ch])
        keep = [np.ascontiguousarray(x) for x in (
            (R.csr_row_ptr[rl:rh + 1] - np.uint32(a)).astype(np.uint32), R.csr_col_idx[a:b], R.csr_val[a:b],
            (R.csc_col_ptr[cl:ch + 1] - np.uint32(c)).astype(np.uint32), R.csc_row_idx[c:d], R.csc_val[c:d])]
        self._keep = keep
        csx = L.mfx_csx(R.rows, R.cols, 0, _vp(keep[3]), _vp(keep[4]), _vp(keep[5]), _vp(keep[0]), _vp(keep[1]), _vp(keep[2]))
        csx.csc_col_ptr = keep[3].ctypes.data_as(C.c_void_p)  # pointer arrays are never empty
        csx.csr_row_ptr = keep[0].ctypes.data_as(C.c_void_p)
        sel = (R.test_row >= rl) & (R.test_row < rh)
        Tl = TestData(R.rows, R.cols, np.ascontiguousarray(R.test_row[sel]), np.ascontiguousarray(R.test_col[sel]),
                      np.ascontiguousarray(R.test_val[sel]))
        self._keep.append(Tl)
        coo = _coo(Tl)
        shard = L.mfx_als_shard(comm.handle, rl, rh, cl, ch, int(R.test_val.shape[0]))
        L.check(L.lib().mfx_als_create_sharded(C.byref(self.handle), C.byref(csx), C.byref(coo), C.byref(cp), C.byref(shard)))

    def set_factors(self, H, W=None):
        _f32c(H, (self.cols, self.k))
        L.check(L.lib().mfx_als_set_factors(self.handle, _vp(W) if W is not None else None, _vp(H), L.MFX_HOST))

    def iterate(self, n_iter: int, with_rmse: bool = True):
        reports = (L.mfx_iter_report * max(1, n_iter))()
        L.check(L.lib().mfx_als_iterate(self.handle, n_iter, 1 if with_rmse else 0, reports))
        return list(reports)[:n_iter]

    def get_factors(self):
        W = np.empty((self.rows, self.k), np.float32)
        H = np.empty((self.cols, self.k), np.float32)
        L.check(L.lib().mfx_als_get_factors(self.handle, _vp(W), _vp(H), L.MFX_HOST))
        return W, H

    def kernel_times(self):
        return _kernel_times(L.lib().mfx_als_kernel_times, self.handle)

    def close(self):
        if self.handle:
            L.lib().mfx_als_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---------------------------------------------------------------------------------------------
# single operators (one per reference function on the path)
# ---------------------------------------------------------------------------------------------
def _u32(a):
    assert a.dtype == np.uint32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(L.u32p)


def _f32(a):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(L.f32p)


def rank_one_sweep(ptr, idx, val, vec, lam: float, variant: int = 1, device: int = 0) -> np.ndarray:
    nseg = ptr.shape[0] - 1
    out = np.empty(nseg, np.float32)
    L.check(L.lib().mfx_rank_one_sweep(nseg, idx.shape[0], _u32(ptr), _u32(idx), _f32(val), vec.shape[0], _f32(vec),
                                       lam, _f32(out), variant, device))
    return out


def update_rating(ptr, idx, val, gathered, per_seg, add: bool, variant: int = 1, device: int = 0) -> None:
    """In place on `val`."""
    nseg = ptr.shape[0] - 1
    L.check(L.lib().mfx_update_rating(nseg, idx.shape[0], _u32(ptr), _u32(idx), _f32(val), gathered.shape[0],
                                      _f32(gathered), _f32(per_seg), 1 if add else 0, variant, device))


def test_rmse(T, W, H, rows: int, cols: int, k: int, ifALS: bool, device: int = 0) -> float:
    coo = _coo(T)
    out = C.c_double(0.0)
    L.check(L.lib().mfx_test_rmse(C.byref(coo), _f32(W), _f32(H), rows, cols, k, 1 if ifALS else 0, C.byref(out), device))
    return float(out.value)


test_rmse.__test__ = False  # not a pytest test


def als_gramian(idx, X, k: int, device: int = 0) -> np.ndarray:
    A = np.empty((k, k), np.float32)
    L.check(L.lib().mfx_als_gramian(idx.shape[0], _u32(idx), X.shape[0], _f32(X), k, _f32(A), device))
    return A


def als_half(ptr, idx, val, X, k: int, lam: float, device: int = 0, variant: int = 1) -> np.ndarray:
    """variant 1: MFMA Gramian + Cholesky solve (the product path); 0: as written, bit-identical to src/ALS.cpp."""
    nseg = ptr.shape[0] - 1
    Y = np.empty((nseg, k), np.float32)
    L.check(L.lib().mfx_als_half(nseg, idx.shape[0], _u32(ptr), _u32(idx), _f32(val), X.shape[0], _f32(X), _f32(Y),
                                 k, lam, variant, device))
    return Y


def als_inverse(A, device: int = 0) -> np.ndarray:
    """inverseMatrix_CholeskyMethod (src/ALS.cpp:41-64) on one matrix, the reference's operation order."""
    A = np.ascontiguousarray(A, np.float32)
    out = np.empty_like(A)
    L.check(L.lib().mfx_als_inverse(A.shape[0], _f32(A), _f32(out), device))
    return out


def partition_rows(R: RatingData, nshards: int) -> np.ndarray:
    bounds = np.zeros(nshards + 1, np.int64)
    L.check(L.lib().mfx_partition_rows(R.rows, _u32(R.csr_row_ptr), nshards, bounds.ctypes.data_as(L.i64p)))
    return bounds


def partition_cols(R: RatingData, nshards: int) -> np.ndarray:
    """nnz-balanced contiguous column blocks (the H-half of a sharded ALS)."""
    bounds = np.zeros(nshards + 1, np.int64)
    L.check(L.lib().mfx_partition_rows(R.cols, _u32(R.csc_col_ptr), nshards, bounds.ctypes.data_as(L.i64p)))
    return bounds


def extract_shard(R: RatingData, row_lo: int, row_hi: int) -> RatingData:
    """Local sub-matrix of rows [row_lo, row_hi): local CSR + local CSC over local row ids; the
    test set is filtered to the same rows (row ids rebased)."""
    lnnz = int(R.csr_row_ptr[row_hi]) - int(R.csr_row_ptr[row_lo])
    nr = row_hi - row_lo
    out = RatingData(nr, R.cols, np.zeros(nr + 1, np.uint32), np.zeros(lnnz, np.uint32), np.zeros(lnnz, np.float32),
                     np.zeros(R.cols + 1, np.uint32), np.zeros(lnnz, np.uint32), np.zeros(lnnz, np.float32))
    csx = _csx(R)
    L.check(L.lib().mfx_extract_shard(C.byref(csx), row_lo, row_hi, _u32(out.csr_row_ptr), _u32(out.csr_col_idx),
                                      _f32(out.csr_val), _u32(out.csc_col_ptr), _u32(out.csc_row_idx), _f32(out.csc_val)))
    keep = (R.test_row >= row_lo) & (R.test_row < row_hi)
    out.test_row = np.ascontiguousarray(R.test_row[keep] - np.uint32(row_lo), dtype=np.uint32)
    out.test_col = np.ascontiguousarray(R.test_col[keep])
    out.test_val = np.ascontiguousarray(R.test_val[keep])
    return out
