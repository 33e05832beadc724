import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

# one hardware queue per stream for the tests that run streams side by side (HIP's default of 4 lets two chains overlap, not more); read
# by the runtime at its first HIP call, so it has to be in the environment before any test touches the GPU
os.environ.setdefault("GPU_MAX_HW_QUEUES", "40")

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


@pytest.fixture(scope="session")
def oracle_knn_lib():
    """oracle/liboracle_knn.so -- the plain-C restatement (test infrastructure)."""
    path = os.path.join(REPO, "oracle", "liboracle_knn.so")
    if not os.path.exists(path):
        subprocess.check_call(["make", "-C", os.path.join(REPO, "oracle"), "all"])
    lib = ctypes.CDLL(path)
    lib.oracle_knn.restype = ctypes.c_int
    lib.oracle_knn.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_long] * 5
    return lib


def run_oracle_knn(lib, ref, query, k):
    ref = np.ascontiguousarray(ref, np.float32)
    query = np.ascontiguousarray(query, np.float32)
    b, d, nr = ref.shape
    nq = query.shape[2]
    idx = np.zeros((b, k, nq), np.int64)
    rc = lib.oracle_knn(ref.ctypes.data, query.ctypes.data, idx.ctypes.data, b, d, nr, nq, k)
    assert rc == 1
    return idx
