"""Shared test plumbing: marker registration, import paths, fixture loaders."""
import os
import sys

# OpenMP worker threads of the CPU oracle (and of anything else in the process) must sleep between parallel
# regions, not spin: the GPU box grants 16 cores, and spinning pools made the full-size oracle runs 5x slower
# inside the suite than on their own.  Must be in the environment before the first OpenMP runtime loads.
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
os.environ.setdefault("GOMP_SPINCOUNT", "0")

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "cuda-recommender_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")
CASES = ("tiny", "small", "edge")


# Some libraries narrow the CPU affinity of the thread that initialises them (RCCL pins around its GPU's NUMA
# node) and every thread created afterwards inherits the narrow mask -- the oracle's OpenMP workers among them:
# the full-size ALS oracle run took 240 s inside the suite against 41 s on its own.  Put the mask of every thread
# of the process back before each test.
_AFFINITY0 = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else None


@pytest.fixture(autouse=True)
def _restore_cpu_affinity():
    if _AFFINITY0 is not None:
        try:
            for tid in os.listdir("/proc/self/task"):
                try:
                    if os.sched_getaffinity(int(tid)) != _AFFINITY0:
                        os.sched_setaffinity(int(tid), _AFFINITY0)
                except OSError:
                    pass
        except OSError:
            pass
    yield


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    """Returns (fixture dict, RatingData) for tests/golden/<name>.npz (plain arrays only)."""
    from mfx import dataset as ds
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    g = {k: z[k] for k in z.files}
    d = ds.RatingData(int(g["rows"][0]), int(g["cols"][0]), g["csr_row_ptr"], g["csr_col_idx"], g["csr_val"],
                      g["csc_col_ptr"], g["csc_row_idx"], g["csc_val"], g["test_row"], g["test_col"], g["test_val"])
    d.validate()
    return g, d


@pytest.fixture(params=CASES)
def golden(request):
    return (request.param,) + load_golden(request.param)


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)
