"""Builds the pybind11 module `knn_pytorch` (src/knn_binding.cpp over src/knn_rocm.h) in-tree, next to this file: the compiled form of the
reference's one native binding (DenseFusion/lib/knn/src/knn.h:12, vision.cpp:3-5) against libape_hip.so.

    python -m autoposeestimation_amd.DenseFusion.lib.knn.build_ext

Host C++ only (the kernel lives in libape_hip.so): g++ with torch's headers and libraries, ROCm's HIP headers for c10/hip/HIPStream.h, an
rpath to both library directories.  No GPU is needed to build."""
import os
import subprocess
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(HERE))))
PKG = os.path.join(REPO, "autoposeestimation_amd")
TARGET = os.path.join(HERE, "knn_pytorch" + sysconfig.get_config_var("EXT_SUFFIX"))


def build(force=False):
    import torch
    from torch.utils import cpp_extension as ce
    srcs = [os.path.join(HERE, "src", "knn_binding.cpp"), os.path.join(HERE, "src", "knn_rocm.h"), os.path.join(REPO, "include", "ape_hip.h")]
    if not force and os.path.exists(TARGET) and all(os.path.getmtime(TARGET) >= os.path.getmtime(s) for s in srcs):
        return TARGET
    tlib = ce.library_paths()[0]
    cmd = ["g++", "-shared", "-fPIC", "-O2", "-std=c++17", "-DTORCH_EXTENSION_NAME=knn_pytorch", "-DTORCH_API_INCLUDE_EXTENSION_H",
           "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI),
           "-I" + os.path.join(REPO, "include"), "-I/opt/rocm/include", "-I" + sysconfig.get_paths()["include"]]
    cmd += ["-I" + p for p in ce.include_paths()]
    cmd += [srcs[0], "-o", TARGET, "-L" + tlib, "-L" + PKG, "-lc10", "-lc10_hip", "-ltorch", "-ltorch_cpu", "-ltorch_hip", "-ltorch_python", "-lape_hip",
            "-Wl,-rpath," + tlib, "-Wl,-rpath," + PKG, "-Wl,-rpath,/opt/rocm/lib", "-Wno-attributes"]
    subprocess.check_call(cmd)
    return TARGET


if __name__ == "__main__":
    print(build(force=True))
