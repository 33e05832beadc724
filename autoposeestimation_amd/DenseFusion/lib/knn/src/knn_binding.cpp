// pybind11 module `knn_pytorch` over knn_rocm.h -- what DenseFusion/lib/knn/src/vision.cpp:3-5 binds in the reference, built against
// libape_hip.so instead of knn.cu / knn_cpu.cpp (autoposeestimation_amd/DenseFusion/lib/knn/build_ext.py).
#include "knn_rocm.h"

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m)
{
    m.def("knn", &knn, "k-nearest neighbours on the GPU: fills idx[B,k,Nq] (1-based) for query[B,D,Nq] against ref[B,D,Nr]; returns 1");
}
