"""Pins oracle/knn_oracle.c against (a) golden indices produced by the reference's own knn_cpu.cpp and
(b) when present (build container only) the reference object oracle/_ref/libknn_ref.so itself."""
import ctypes
import os

import numpy as np
import pytest

from conftest import REPO, golden, run_oracle_knn


def test_oracle_knn_matches_reference_golden(oracle_knn_lib):
    g = golden("knn")
    for i in range(int(g["n_cases"])):
        ref, qry, want = g["ref_%d" % i], g["query_%d" % i], g["idx_%d" % i]
        got = run_oracle_knn(oracle_knn_lib, ref, qry, want.shape[1])
        assert np.array_equal(got, want), "case %d" % i


def test_oracle_knn_matches_reference_object(oracle_knn_lib):
    path = os.path.join(REPO, "oracle", "_ref", "libknn_ref.so")
    if not os.path.exists(path):
        pytest.skip("oracle/_ref not built (reference absent on this machine)")
    import torch  # noqa: F401  (resolves the torch symbols the reference object links against)
    lib = ctypes.CDLL(path)
    lib.ref_knn.restype = ctypes.c_int
    lib.ref_knn.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_long] * 5
    rng = np.random.default_rng(123)
    for (b, d, nr, nq, k, qz) in [(1, 3, 300, 400, 1, 0), (1, 3, 128, 256, 1, 0.5), (2, 5, 40, 33, 3, 0.5)]:
        ref = rng.standard_normal((b, d, nr)).astype(np.float32)
        qry = rng.standard_normal((b, d, nq)).astype(np.float32)
        if qz:
            ref, qry = (np.round(ref / qz) * qz).astype(np.float32), (np.round(qry / qz) * qz).astype(np.float32)
        want = np.zeros((b, k, nq), np.int64)
        assert lib.ref_knn(ref.ctypes.data, qry.ctypes.data, want.ctypes.data, b, d, nr, nq, k) == 1
        assert np.array_equal(run_oracle_knn(oracle_knn_lib, ref, qry, k), want)
