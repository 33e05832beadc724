"""Drop-in for DenseFusion/lib/knn/__init__.py (reference :9-23) and its pybind module `knn_pytorch`
(src/vision.cpp:3-5), backed by the gfx950 kernel `ape_knn_f32` (include/ape_hip.h).

    knn = KNearestNeighbor(k)
    inds = knn(ref[B,D,Nr], query[B,D,Nq])      # -> int64 [B,k,Nq], 1-based, on the input's device

The reference wrapper is a legacy autograd Function that .float().cuda()s its inputs (:16-17); this one
does the same conversion and is a plain callable (it never had a backward)."""
import torch

from autoposeestimation_amd import _lib


class _KnnPytorch:
    """Stands in for the pybind11 module `knn_pytorch`: knn(ref, query, idx) fills idx in place, returns 1."""

    @staticmethod
    def knn(ref, query, idx):
        if ref.dim() != 3 or query.dim() != 3 or idx.dim() != 3:
            raise ValueError("knn expects ref[B,D,Nr], query[B,D,Nq], idx[B,k,Nq]")
        if ref.size(0) != query.size(0) or ref.size(1) != query.size(1) or idx.size(0) != ref.size(0) \
                or idx.size(2) != query.size(2):
            raise ValueError("knn: inconsistent shapes %s %s %s" % (tuple(ref.shape), tuple(query.shape), tuple(idx.shape)))
        rc = _lib.lib().ape_knn_f32(_lib.dptr(ref, torch.float32), _lib.dptr(query, torch.float32),
                                    _lib.dptr(idx, torch.int64), ref.size(0), ref.size(1), ref.size(2),
                                    query.size(2), idx.size(1), _lib.stream_ptr())
        _lib.check(rc, "ape_knn_f32")
        return 1  # knn.h:63


def load_compiled():
    """The COMPILED binding (src/knn_binding.cpp over src/knn_rocm.h, built by build_ext.py / __graft_entry__.build()): the pybind11
    module the reference's `from lib.knn import knn_pytorch` resolves to, calling ape_knn_f32 through the C ABI from C++.  None when it
    has not been built."""
    import importlib.util
    import os
    import sysconfig
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "knn_pytorch" + sysconfig.get_config_var("EXT_SUFFIX"))
    if not os.path.exists(path):
        return None
    spec = importlib.util.spec_from_file_location("knn_pytorch", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


# the ctypes form is the default (no build step beyond libape_hip.so); both call the same exported symbol
knn_pytorch = _KnnPytorch()


class KNearestNeighbor:
    """Compute k nearest neighbours for each query point (reference knn/__init__.py:9-23)."""

    def __init__(self, k):
        self.k = k

    def forward(self, ref, query):
        ref = ref.float().cuda().contiguous()
        query = query.float().cuda().contiguous()
        inds = torch.empty(query.shape[0], self.k, query.shape[2], dtype=torch.int64, device=query.device)
        knn_pytorch.knn(ref, query, inds)
        return inds

    __call__ = forward
