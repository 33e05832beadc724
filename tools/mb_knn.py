"""k-NN (k = 1, dim 3) throughput: pairs/s and the share of the fp32 vector rate at 11 lane-operations per pair, every form.
Needs the ablation build (make -C autoposeestimation_amd/csrc ablations; APE_HIP_LIB=autoposeestimation_amd/libape_hip_abl.so): the
product library has no ape_knn_debug and only the form it launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from autoposeestimation_amd import _lib
from autoposeestimation_amd.DenseFusion.lib.knn import KNearestNeighbor
PEAK_LANE_OPS = 256 * 4 * 32 * 2.4e9       # lanes per clock x clock: one fp32 operation per lane and cycle (an FMA counts once here)
knn = KNearestNeighbor(1)
import ctypes
_lib.lib().ape_knn_debug.argtypes = [ctypes.c_int]      # (exported by the ablation build only; not part of include/ape_hip.h)
def t(f, n=5, rounds=5):
    f(); torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): f()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n)
    return sorted(ts)[len(ts) // 2]
for b, nr, nq in ((1, 1000, 1_000_000), (1, 1000, 4_000_000), (64, 1000, 1000), (32, 1000, 1000), (1, 500, 1_000_000)):
    ref = torch.randn(b, 3, nr, device="cuda"); qry = torch.randn(b, 3, nq, device="cuda")
    row = []
    for dbg in (0, 4, 8, 2, 1):
        _lib.lib().ape_knn_debug(dbg)
        ms = t(lambda: knn(ref, qry))
        pairs = b * nr * nq / (ms * 1e-3)
        row.append("%s %.3f ms %.2f" % ({0: "auto", 1: "one query per lane", 2: "no group minima", 4: "four per lane", 8: "two per lane"}[dbg], ms, pairs * 11 / PEAK_LANE_OPS))
    _lib.lib().ape_knn_debug(0)
    print("%d x %d refs x %d queries:  %s" % (b, nr, nq, "  |  ".join(row)))
