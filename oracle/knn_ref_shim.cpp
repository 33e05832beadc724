// TEST INFRASTRUCTURE: C-linkage door onto the reference's own knn_cpu(), compiled in place from
// /root/reference/DenseFusion/lib/knn/src/cpu/knn_cpu.cpp by oracle/Makefile (target `ref`).
// This file contains no reference code: it declares the prototype published in
// DenseFusion/lib/knn/src/cpu/vision.h:4-6 and performs the batch loop + scratch allocation the
// reference dispatcher performs in DenseFusion/lib/knn/src/knn.h:54-63.
#include <cstdlib>

void knn_cpu(float* ref_dev, int ref_width, float* query_dev, int query_width,
             int height, int k, float* dist_dev, long* ind_dev, long* ind_buf);

extern "C" int ref_knn(const float* ref, const float* query, long* idx,
                       long batch, long dim, long ref_nb, long query_nb, long k)
{
    float* dist = static_cast<float*>(std::malloc(sizeof(float) * ref_nb * query_nb));
    long*  buf  = static_cast<long*>(std::malloc(sizeof(long) * ref_nb));
    if (!dist || !buf) { std::free(dist); std::free(buf); return -1; }
    for (long b = 0; b < batch; ++b)
        knn_cpu(const_cast<float*>(ref) + b * dim * ref_nb, (int)ref_nb,
                const_cast<float*>(query) + b * dim * query_nb, (int)query_nb,
                (int)dim, (int)k, dist, idx + b * k * query_nb, buf);
    std::free(dist);
    std::free(buf);
    return 1;   // knn.h:63
}
