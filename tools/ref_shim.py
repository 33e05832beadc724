"""Import shim that makes the reference's Python importable IN THE BUILD CONTAINER ONLY.

Used exclusively by tools/gen_golden.py to produce tests/golden/*.npz.  Nothing under tests/,
bench.py or the package imports this module: /root/reference does not exist on the GPU box.

Third-party modules the reference imports but that are not installed here are replaced with empty
stand-ins *for import purposes only* -- no arithmetic of theirs is emulated, so no golden vector
depends on them.
"""
import ctypes
import os
import sys
import types

import numpy as np
import torch

REFERENCE_ROOT = os.environ.get("APE_REFERENCE_ROOT", "/root/reference")
_REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _Anything(types.ModuleType):
    """Module whose every attribute is another stand-in (so `from x import y` succeeds)."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        sub = _Anything(self.__name__ + "." + name)
        setattr(self, name, sub)
        return sub

    __path__ = []  # lets `import a.b.c` treat every stand-in as a package

    def __call__(self, *a, **k):
        return _Anything(self.__name__ + "()")


_STUBBED = [
    "torchvision", "torchvision.transforms", "torchvision.transforms.functional", "torchvision.utils",
    "torchvision.models",
    "open3d", "cv2", "segmentation_models_pytorch", "transforms3d", "mathutils",
    "mathutils.geometry", "pyrealsense2", "matplotlib", "matplotlib.pyplot", "matplotlib.animation",
    "imageio", "scipy.io",
]


def _load_ref_knn():
    """ctypes handle on oracle/_ref/libknn_ref.so (the reference's knn_cpu.cpp compiled in place)."""
    path = os.path.join(_REPO, "oracle", "_ref", "libknn_ref.so")
    if not os.path.exists(path):
        raise RuntimeError("run `make -C oracle ref` first")
    lib = ctypes.CDLL(path)
    lib.ref_knn.restype = ctypes.c_int
    lib.ref_knn.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_long] * 5
    return lib


def ref_knn(ref: torch.Tensor, query: torch.Tensor, k: int) -> torch.Tensor:
    lib = _load_ref_knn()
    ref = ref.float().contiguous()
    query = query.float().contiguous()
    inds = torch.empty(query.shape[0], k, query.shape[2], dtype=torch.int64)
    rc = lib.ref_knn(ref.data_ptr(), query.data_ptr(), inds.data_ptr(),
                     ref.shape[0], ref.shape[1], ref.shape[2], query.shape[2], k)
    assert rc == 1
    return inds


class _RefKNearestNeighbor:
    """CPU caller of the reference's compiled knn (the reference wrapper, knn/__init__.py:9-23,
    is a legacy autograd Function that also forces .cuda(); it cannot run on torch 2.x)."""

    def __init__(self, k):
        self.k = k

    def __call__(self, ref, query):
        return ref_knn(ref, query, self.k)


def install():
    if not os.path.isdir(REFERENCE_ROOT):
        raise RuntimeError("reference not present: %s" % REFERENCE_ROOT)
    for name in _STUBBED:
        if name not in sys.modules:
            sys.modules[name] = _Anything(name)
    if not hasattr(np, "float"):
        np.float = float  # reference uses the removed alias in non-golden code paths
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    # knn_pytorch pybind module is absent (.MISSING_LARGE_BLOBS); give the package a placeholder so
    # `from DenseFusion.lib.knn import knn_pytorch` resolves, then patch the wrapper class.
    sys.modules.setdefault("DenseFusion.lib.knn.knn_pytorch", _Anything("knn_pytorch"))
    import DenseFusion.lib.knn as knn_pkg
    knn_pkg.knn_pytorch = sys.modules["DenseFusion.lib.knn.knn_pytorch"]
    import DenseFusion.lib.loss as loss
    import DenseFusion.lib.loss_refiner as loss_refiner
    loss.KNearestNeighbor = _RefKNearestNeighbor
    loss_refiner.KNearestNeighbor = _RefKNearestNeighbor
    return True
