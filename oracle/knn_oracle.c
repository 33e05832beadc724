/* TEST INFRASTRUCTURE -- CPU oracle for the native k-NN of KochPJ/AutoPoseEstimation.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product path (autoposeestimation_amd/csrc) never calls it.
 *
 * Restates DenseFusion/lib/knn/src/cpu/knn_cpu.cpp:4-55 + the dispatcher knn.h:12-66:
 *   - squared L2 distance accumulated in float32, dimension by dimension, starting from 0,
 *     one rounding per multiply and per add (no FMA)                       (knn_cpu.cpp:8-16)
 *   - the reference bubble-sorts ALL refs per query, swapping only on strict '>' (knn_cpu.cpp:21-40),
 *     i.e. a stable ascending sort: the k results are the k smallest distances, ties in
 *     ascending ref index.  We obtain the same k entries by stable selection, O(Nq*Nr*k).
 *   - output is 1-based and laid out ind[b][i][q]                          (knn_cpu.cpp:42-43)
 *   - NaN distances: bubble sort never swaps on NaN; inputs with NaN are outside the contract.
 *
 * Parity: pinned against the reference object itself (oracle/_ref/libknn_ref.so, built from the
 * reference source in place) in tests/test_oracle_knn.py and against tests/golden/knn_*.npz.
 */
#include <stdint.h>
#include <stdlib.h>

static float sqdist(const float* ref, long ref_nb, long r, const float* query, long query_nb, long q, long dim)
{
    float d = 0.0f;
    for (long h = 0; h < dim; ++h) {
        float diff = ref[h * ref_nb + r] - query[h * query_nb + q];
        float sq = diff * diff;
        d = d + sq;
    }
    return d;
}

/* returns 1 like the reference dispatcher (knn.h:63); -1 on allocation failure */
int oracle_knn(const float* ref, const float* query, int64_t* idx,
               long batch, long dim, long ref_nb, long query_nb, long k)
{
    float* dist = (float*)malloc(sizeof(float) * (size_t)(ref_nb > 0 ? ref_nb : 1));
    char* taken = (char*)malloc((size_t)(ref_nb > 0 ? ref_nb : 1));
    if (!dist || !taken) { free(dist); free(taken); return -1; }
    for (long b = 0; b < batch; ++b) {
        const float* rb = ref + b * dim * ref_nb;
        const float* qb = query + b * dim * query_nb;
        int64_t* ib = idx + b * k * query_nb;
        for (long q = 0; q < query_nb; ++q) {
            for (long r = 0; r < ref_nb; ++r) { dist[r] = sqdist(rb, ref_nb, r, qb, query_nb, q, dim); taken[r] = 0; }
            for (long i = 0; i < k && i < ref_nb; ++i) {
                long best = -1;
                for (long r = 0; r < ref_nb; ++r) {
                    if (taken[r]) continue;
                    if (best < 0 || dist[r] < dist[best]) best = r;   /* strict '<': lowest index wins ties */
                }
                taken[best] = 1;
                ib[q + i * query_nb] = best + 1;
            }
        }
    }
    free(dist);
    free(taken);
    return 1;
}
