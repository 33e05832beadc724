// Probe: operand layout of v_mfma_scale_f32_16x16x128_f8f6f4 with e2m3 (FP6) operands -- which lane / register bits hold A[m][k] and
// B[k][n], how the E8M0 scales apply.  D = sum_k A[m][k] B[k][n] against a host evaluation of the same codes.
// Hypothesis under test: lane l holds row (l & 15) and the 32 k-values 32 (l >> 4) .. + 31, value i of them in bits [6 i, 6 i + 5] of
// the lane's 8-register operand (little endian, registers 6 and 7 unused); the scale register's byte 0 is the E8M0 exponent of that
// lane's 32-value block; C/D: lane l, register e = D[4 (l >> 4) + e][l & 15] (row = A's m, column = B's n).
// hipcc --offload-arch=gfx950 -O2 tools/probes/mx6_layout_probe.hip -o /tmp/mx6_probe && /tmp/mx6_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

__global__ void probe(const unsigned* a_regs, const unsigned* b_regs, const unsigned* sa, const unsigned* sb, float* d)
{
    const int l = threadIdx.x;
    i32x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (int)a_regs[l * 8 + i]; b[i] = (int)b_regs[l * 8 + i]; }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 2, 2, 0, (int)sa[l], 0, (int)sb[l]);
#pragma unroll
    for (int e = 0; e < 4; ++e) d[l * 4 + e] = c[e];
}

static float e2m3(unsigned code)
{
    const int s = (code >> 5) & 1, e = (code >> 3) & 3, m = code & 7;
    const float v = e == 0 ? m / 8.f : ldexpf(1.f + m / 8.f, e - 1);
    return s ? -v : v;
}

int main()
{
    std::vector<unsigned> acode(16 * 128), bcode(128 * 16), areg(64 * 8, 0), breg(64 * 8, 0), sa(64), sb(64);
    unsigned s = 99;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s >> 8; };
    for (auto& v : acode) v = rnd() & 63;
    for (auto& v : bcode) v = rnd() & 63;
    for (int l = 0; l < 64; ++l) { sa[l] = 120 + rnd() % 12 + (0xAB00u << 8); sb[l] = 122 + rnd() % 9 + (0xCD00u << 8); }    // junk in the upper bytes: must be ignored
    for (int l = 0; l < 64; ++l) {
        const int m = l & 15, g = l >> 4;
        for (int i = 0; i < 32; ++i) {
            const int k = 32 * g + i, bit = 6 * i;
            const unsigned ca = acode[m * 128 + k], cb = bcode[k * 16 + m];
            for (int t = 0; t < 6; ++t) {
                if ((ca >> t) & 1) areg[l * 8 + (bit + t) / 32] |= 1u << ((bit + t) % 32);
                if ((cb >> t) & 1) breg[l * 8 + (bit + t) / 32] |= 1u << ((bit + t) % 32);
            }
        }
        areg[l * 8 + 6] = 0xDEADBEEF; areg[l * 8 + 7] = 0x12345678;      // must be ignored
        breg[l * 8 + 6] = 0xFFFFFFFF; breg[l * 8 + 7] = 0x0BADF00D;
    }
    unsigned *da, *db, *dsa, *dsb;
    float* dd;
    hipMalloc(&da, 64 * 8 * 4); hipMalloc(&db, 64 * 8 * 4); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256); hipMalloc(&dd, 64 * 4 * 4);
    hipMemcpy(da, areg.data(), 64 * 8 * 4, hipMemcpyHostToDevice); hipMemcpy(db, breg.data(), 64 * 8 * 4, hipMemcpyHostToDevice);
    hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dd);
    std::vector<float> d(256);
    hipMemcpy(d.data(), dd, 1024, hipMemcpyDeviceToHost);
    double worst = 0;
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int e = 0; e < 4; ++e) {
            const int row = 4 * (l >> 4) + e, col = l & 15;
            double want = 0;
            for (int k = 0; k < 128; ++k) {
                const int g = k / 32;
                const double xa = e2m3(acode[row * 128 + k]) * ldexp(1.0, (int)(sa[(g << 4) | row] & 255) - 127);
                const double xb = e2m3(bcode[k * 16 + col]) * ldexp(1.0, (int)(sb[(g << 4) | col] & 255) - 127);
                want += xa * xb;
            }
            const double err = fabs(d[l * 4 + e] - want) / (fabs(want) + 1e-30);
            if (err > 1e-5) { if (bad < 6) printf("lane %d reg %d: got %g want %g\n", l, e, d[l * 4 + e], want); ++bad; }
            if (err > worst) worst = err;
        }
    printf("%s: %d of 256 outputs off, worst relative error %.3g\n", bad ? "LAYOUT HYPOTHESIS WRONG" : "layout confirmed", bad, worst);
    return 0;
}
