"""ctypes binding of libmfx.so (include/mfx.h).  The library is the product; this file only
declares its C ABI.  Loading fails loudly when the shared object is missing -- there is no
Python or CPU fallback for any compute entry point."""
from __future__ import annotations

import ctypes as C
import os

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.environ.get("MFX_LIB_PATH") or os.path.join(_PKG, "libmfx.so")  # (override: A/B builds of the library)

MFX_HOST, MFX_DEVICE = 0, 1
MFX_COMM_ID_BYTES = 128
MFX_VERSION = 2  # include/mfx.h

u32p = C.POINTER(C.c_uint32)
f32p = C.POINTER(C.c_float)
f64p = C.POINTER(C.c_double)
i64p = C.POINTER(C.c_int64)


class mfx_csx(C.Structure):
    _fields_ = [("rows", C.c_int64), ("cols", C.c_int64), ("nnz", C.c_int64),
                ("csc_col_ptr", C.c_void_p), ("csc_row_idx", C.c_void_p), ("csc_val", C.c_void_p),
                ("csr_row_ptr", C.c_void_p), ("csr_col_idx", C.c_void_p), ("csr_val", C.c_void_p)]


class mfx_coo(C.Structure):
    _fields_ = [("nnz", C.c_int64), ("row", C.c_void_p), ("col", C.c_void_p), ("val", C.c_void_p)]


class mfx_params(C.Structure):
    _fields_ = [("k", C.c_uint32), ("lambda_", C.c_float), ("maxiter", C.c_int32), ("maxinneriter", C.c_int32),
                ("nBlocks", C.c_uint32), ("nThreadsPerBlock", C.c_uint32), ("verbose", C.c_int32),
                ("device", C.c_int32), ("schedule", C.c_int32), ("kernel_variant", C.c_int32),
                ("profile", C.c_int32), ("tiles_per_span", C.c_int32), ("panel_rows", C.c_int32),
                ("wg_waves", C.c_int32), ("graph", C.c_int32), ("layout_build", C.c_int32),
                ("do_nmf", C.c_int32), ("eps", C.c_float), ("rank_trace", C.c_int32)]


class mfx_iter_report(C.Structure):
    _fields_ = [("rank_time", C.c_double), ("update_time", C.c_double), ("rmse", C.c_double),
                ("rmse_time", C.c_double)]


class mfx_shard(C.Structure):
    _fields_ = [("comm", C.c_void_p), ("global_col_nnz", C.c_void_p), ("global_test_nnz", C.c_int64)]


class mfx_als_shard(C.Structure):
    _fields_ = [("comm", C.c_void_p), ("row_lo", C.c_int64), ("row_hi", C.c_int64), ("col_lo", C.c_int64),
                ("col_hi", C.c_int64), ("global_test_nnz", C.c_int64)]


class MfxError(RuntimeError):
    pass


# name -> (restype, argtypes); every symbol include/mfx.h declares
SIGNATURES = {
    "mfx_last_error": (C.c_char_p, []),
    "mfx_version": (C.c_int, []),
    "mfx_device_count": (C.c_int, []),
    "mfx_params_default": (None, [C.POINTER(mfx_params)]),
    "mfx_ccdpp_run": (C.c_int, [C.POINTER(mfx_csx), C.POINTER(mfx_coo), C.c_void_p, C.c_void_p,
                                C.POINTER(mfx_params), C.POINTER(mfx_iter_report)]),
    "mfx_als_run": (C.c_int, [C.POINTER(mfx_csx), C.POINTER(mfx_coo), C.c_void_p, C.c_void_p,
                              C.POINTER(mfx_params), C.POINTER(mfx_iter_report)]),
    "mfx_ccd_create": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(mfx_csx), C.POINTER(mfx_coo),
                                 C.POINTER(mfx_params), C.c_int, C.POINTER(mfx_shard)]),
    "mfx_ccd_set_factors": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "mfx_ccd_iterate": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(mfx_iter_report)]),
    "mfx_ccd_get_factors": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "mfx_ccd_get_residual": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "mfx_ccd_kernel_times": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_char_p), f64p, i64p]),
    "mfx_ccd_set_profile": (C.c_int, [C.c_void_p, C.c_int]),
    "mfx_ccd_layout_info": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int32)]),
    "mfx_ccd_rank_trace": (C.c_int, [C.c_void_p, C.c_int, f64p, f64p, C.c_int, C.POINTER(C.c_int32)]),
    "mfx_ccd_destroy": (C.c_int, [C.c_void_p]),
    "mfx_als_create": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(mfx_csx), C.POINTER(mfx_coo),
                                 C.POINTER(mfx_params), C.c_int]),
    "mfx_als_create_sharded": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(mfx_csx), C.POINTER(mfx_coo),
                                         C.POINTER(mfx_params), C.POINTER(mfx_als_shard)]),
    "mfx_als_set_factors": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "mfx_als_iterate": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(mfx_iter_report)]),
    "mfx_als_get_factors": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "mfx_als_kernel_times": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_char_p), f64p, i64p]),
    "mfx_als_destroy": (C.c_int, [C.c_void_p]),
    "mfx_rank_one_sweep": (C.c_int, [C.c_int64, C.c_int64, u32p, u32p, f32p, C.c_int64, f32p, C.c_float, f32p,
                                     C.c_int, C.c_int]),
    "mfx_update_rating": (C.c_int, [C.c_int64, C.c_int64, u32p, u32p, f32p, C.c_int64, f32p, f32p, C.c_int,
                                    C.c_int, C.c_int]),
    "mfx_test_rmse": (C.c_int, [C.POINTER(mfx_coo), f32p, f32p, C.c_int64, C.c_int64, C.c_int64, C.c_int, f64p,
                                C.c_int]),
    "mfx_als_gramian": (C.c_int, [C.c_int64, u32p, C.c_int64, f32p, C.c_int64, f32p, C.c_int]),
    "mfx_als_inverse": (C.c_int, [C.c_int64, f32p, f32p, C.c_int]),
    "mfx_als_half": (C.c_int, [C.c_int64, C.c_int64, u32p, u32p, f32p, C.c_int64, f32p, f32p, C.c_int64,
                               C.c_float, C.c_int, C.c_int]),
    "mfx_comm_unique_id": (C.c_int, [C.c_void_p]),
    "mfx_comm_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "mfx_comm_create_local": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int]),
    "mfx_comm_agree": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int)]),
    "mfx_comm_abort": (C.c_int, [C.c_void_p]),
    "mfx_comm_rank": (C.c_int, [C.c_void_p]),
    "mfx_comm_size": (C.c_int, [C.c_void_p]),
    "mfx_comm_destroy": (C.c_int, [C.c_void_p]),
    "mfx_initial_col": (None, [f32p, C.c_int64, C.c_int64]),
    "mfx_partition_rows": (C.c_int, [C.c_int64, u32p, C.c_int, i64p]),
    "mfx_extract_shard": (C.c_int, [C.POINTER(mfx_csx), C.c_int64, C.c_int64, u32p, u32p, f32p, u32p, u32p, f32p]),
}

_LIB = None
_HIP_RUNTIME = None  # path of the HIP runtime pinned for this process (diagnostics, tests)


def mapped_hip_runtimes() -> list:
    """Distinct libamdhip64 files mapped into this process (more than one = two HIP runtimes)."""
    try:
        with open("/proc/self/maps") as f:
            return sorted({ln.split()[-1] for ln in f if "libamdhip64" in ln})
    except OSError:
        return []


def _pin_one_hip_runtime() -> None:
    """One HIP runtime per process, whatever the import order.

    PyTorch-ROCm wheels bundle their own libamdhip64 / libhsa-runtime64 under the SONAME of
    /opt/rocm's (libamdhip64.so.7).  The dynamic loader reuses whichever copy is mapped first for
    every later NEEDED entry with that SONAME, but torch's own libraries ask for the un-versioned
    file name and so always get torch's copy: `import mfx; import torch` used to end with /opt/rocm's
    runtime serving libmfx and torch's serving torch -- two HSA clients in one process, device
    pointers of one unknown to the other, heap corruption at exit (seen on the GPU box, round 1).
    So when torch is installed but not imported yet, its copy is mapped first (RTLD_GLOBAL, exactly
    what torch's own _load_global_deps does); libmfx's NEEDED libamdhip64.so.7 then resolves to it,
    and a later `import torch` finds the file already loaded.  MFX_HIP_RUNTIME=system opts out (for
    processes that will never import torch)."""
    global _HIP_RUNTIME
    import sys
    if os.environ.get("MFX_HIP_RUNTIME", "") == "system":
        return
    already = mapped_hip_runtimes()
    if already:  # torch (or the host program) got there first: libmfx will share that copy
        _HIP_RUNTIME = already[0]
        return
    if "torch" in sys.modules:
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
    cand = os.path.join(libdir, "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)
        _HIP_RUNTIME = cand
        # the same for RCCL, lazily: libmfx dlopens it on first use and takes this path first (csrc/comm.cpp)
        rccl = os.path.join(libdir, "librccl.so")
        if os.path.exists(rccl):
            os.environ.setdefault("MFX_RCCL_PATH", rccl)


def lib() -> C.CDLL:
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise MfxError(f"{LIB_PATH} is missing: build it with `make -C {_PKG} lib` "
                           "(or __graft_entry__.build()); there is no fallback path")
        _pin_one_hip_runtime()
        _LIB = C.CDLL(LIB_PATH)
        two = mapped_hip_runtimes()
        if len(two) > 1:
            raise MfxError("two HIP runtimes are mapped into this process (" + ", ".join(two) + "): libmfx "
                           "would not see the other runtime's device memory; import torch before mfx, or "
                           "set MFX_HIP_RUNTIME=system in processes that never load torch")
        _LIB.mfx_version.restype = C.c_int
        if _LIB.mfx_version() != MFX_VERSION:  # struct layouts / argument lists of another revision: never call into it
            got = _LIB.mfx_version()
            _LIB = None
            raise MfxError(f"{LIB_PATH} speaks ABI revision {got}, this binding revision {MFX_VERSION} "
                           f"(include/mfx.h MFX_VERSION): rebuild with `make -C {_PKG} lib`")
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(_LIB, name)  # AttributeError here == the .so does not export the ABI
            fn.restype = res
            fn.argtypes = args
    return _LIB


def check(rc: int) -> None:
    if rc != 0:
        raise MfxError(f"libmfx error {rc}: {lib().mfx_last_error().decode(errors='replace')}")
