"""mfx -- Python host mirror of the MI355X-native CCD++/ALS solver (libmfx.so, include/mfx.h)."""
from . import dataset  # noqa: F401
from .api import (AlsSolver, CcdSolver, Comm, TestData, UsageError, als_gramian, als_half, als_inverse,  # noqa: F401
                  calculate_rmse_directly, device_count, extract_shard, golden_compare, initial_col,
                  kernel_wrapper_als_NV, kernel_wrapper_ccdpp_NV, parameter, parse_command_line,
                  partition_cols, partition_rows, rank_one_sweep, solvertype, test_data_of, test_rmse, update_rating)
from ._lib import LIB_PATH, MfxError, lib  # noqa: F401
