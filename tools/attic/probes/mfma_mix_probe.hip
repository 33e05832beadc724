// Probe: matrix-pipe time of one "128-k unit" of a 16x16 output block under three operand splits, with the
// real kernels' register shape (16 independent accumulator blocks per wave, 2 waves per SIMD):
//   bf16x3   : 12 x v_mfma_f32_16x16x32_bf16                                   (today: xh wh + xl wh + xh wl per 32-k chunk)
//   f16+mx8  : 4 x v_mfma_f32_16x16x32_f16 + 2 x v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3 cross terms)
//   f16+mx6  : 4 x v_mfma_f32_16x16x32_f16 + 2 x v_mfma_scale_f32_16x16x128_f8f6f4 (e2m3 cross terms)
//   f16x2    : 8 x v_mfma_f32_16x16x32_f16                                      (two of three products: the timing ablation's mix)
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O2 tools/probes/mfma_mix_probe.hip -o /tmp/mfma_mix && /tmp/mfma_mix
#include <hip/hip_runtime.h>
#include <cstdio>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(512) void mix_kernel(float* out, const int* in, int iters)
{
    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    i32x8 a8, b8;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a8[i] = in[threadIdx.x * 8 + i]; b8[i] = in[4096 + threadIdx.x * 8 + i]; }
    const int sa = in[8192 + threadIdx.x], sb = in[8192 + 512 + threadIdx.x];
    const bf16x8 ah = __builtin_bit_cast(bf16x8, __builtin_shufflevector(a8, a8, 0, 1, 2, 3)), al = __builtin_bit_cast(bf16x8, __builtin_shufflevector(a8, a8, 4, 5, 6, 7));
    const bf16x8 bh = __builtin_bit_cast(bf16x8, __builtin_shufflevector(b8, b8, 0, 1, 2, 3)), bl = __builtin_bit_cast(bf16x8, __builtin_shufflevector(b8, b8, 4, 5, 6, 7));
    const f16x8 fa = __builtin_bit_cast(f16x8, ah), fb = __builtin_bit_cast(f16x8, bh), fbl = __builtin_bit_cast(f16x8, bl);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (MODE == 0) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, al, acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl, ah, acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, ah, acc[i], 0, 0, 0);
                }
            } else if (MODE == 3) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb, fa, acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fbl, fa, acc[i], 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb, fa, acc[i], 0, 0, 0);
                constexpr int FMT = MODE == 1 ? 0 : 2;          // cbsz / blgp: 0 = e4m3, 2 = e2m3
                acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(b8, a8, acc[i], FMT, FMT, 0, sb, 0, sa);
                acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8, acc[i], FMT, FMT, 0, sa, 0, sb);
            }
        }
    }
    f32x4 s = acc[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) s += acc[i];
    out[blockIdx.x * 512 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

template <int MODE>
void run(const char* name, float* out, const int* in, int passes_per_unit)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int iters = 2000;
    hipLaunchKernelGGL(mix_kernel<MODE>, dim3(256), dim3(512), 0, 0, out, in, 10);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(mix_kernel<MODE>, dim3(256), dim3(512), 0, 0, out, in, iters);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: 2 waves x iters x 16 units
    const double units = 2.0 * iters * 16;
    const double ns_per_unit = ms * 1e6 / units;
    printf("%-8s %8.3f ms   %7.1f ns per 128-k unit per SIMD   (%d passes nominal -> %.2f ns per pass; chip-wide %.0f TFLOP/s of algorithmic products)\n", name, ms, ns_per_unit,
           passes_per_unit, ns_per_unit / passes_per_unit, 1024.0 * 2 * 16 * 16 * 128 / ns_per_unit * 1e-3);
}

int main()
{
    int* in;
    float* out;
    (void)hipMalloc(&in, 16384 * sizeof(int));
    (void)hipMalloc(&out, 256 * 512 * sizeof(float));
    int h[16384];
    unsigned s = 12345;
    for (int i = 0; i < 16384; ++i) { s = s * 1664525u + 1013904223u; h[i] = (int)((s >> 4) & 0x3B3B3B3Bu); }        // small finite values in every format
    for (int i = 8192; i < 9216; ++i) h[i] = 0x7F7F7F7F;       // E8M0 scale 2^0 in every byte
    (void)hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    run<0>("bf16x3", out, in, 48);
    run<3>("f16x2", out, in, 32);
    run<1>("f16+mx8", out, in, 32);
    run<2>("f16+mx6", out, in, 24);
    run<0>("bf16x3", out, in, 48);
    return 0;
}
