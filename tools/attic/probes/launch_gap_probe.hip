// Probe: what does a back-to-back launch cost as a function of the workgroup's LDS allocation on gfx950?  The step's launch list
// shows ~11 us between two 160 KB-LDS kernels (halo_s32, gemm_s32<256>), ~6 us entering or leaving one, 0 between 144 KB ones.
// A trivial kernel (every workgroup touches its LDS once and leaves), 256 workgroups of `threads`, launched N times in a row on
// one stream between two events; then the same alternating with a small-LDS kernel.
//   hipcc --offload-arch=gfx950 -O2 launch_gap_probe.hip -o launch_gap_probe && ./launch_gap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
extern __shared__ char lds[];
__global__ void touch(float* out, int lds_bytes, int spin)
{
    float* f = (float*)lds;
    f[threadIdx.x] = (float)threadIdx.x;
    if (lds_bytes >= 4096) f[lds_bytes / 4 - 1 - threadIdx.x] = 1.f;
    __syncthreads();
    float s = f[(threadIdx.x + 1) % blockDim.x];
    for (int i = 0; i < spin; ++i) s = s * 1.0001f + 0.5f;
    if (s == -1.f) out[blockIdx.x] = s;
}
struct Big { float* out; int lds_bytes; int spin; int pad[90]; };      // 376 bytes of kernel arguments, as HaloS32Args
template <bool REGS>
__global__ void touch_big(Big a)
{
    float* f = (float*)lds;
    f[threadIdx.x] = (float)threadIdx.x;
    if (a.lds_bytes >= 4096) f[a.lds_bytes / 4 - 1 - threadIdx.x] = 1.f;
    __syncthreads();
    float s = f[(threadIdx.x + 1) % blockDim.x];
    if (REGS) asm volatile("v_mov_b32 v250, 0" ::: "v250");       // forces the full register file (2 waves / SIMD)
    for (int i = 0; i < a.spin; ++i) s = s * 1.0001f + 0.5f;
    if (s == -1.f) a.out[blockIdx.x] = s;
}
template <bool REGS>
static float run_big(int n, int lds_a, int thr, int spin, float* out)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute(reinterpret_cast<const void*>(touch_big<REGS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    Big a{out, lds_a, spin, {}};
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0, 0);
        for (int i = 0; i < n; ++i) hipLaunchKernelGGL(touch_big<REGS>, dim3(256), dim3(thr), lds_a, 0, a);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
    }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1000.f / n;
}
static float run(int n, int lds_a, int thr_a, int lds_b, int thr_b, int spin, float* out)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {          // (first pass warms up)
        hipEventRecord(e0, 0);
        for (int i = 0; i < n; ++i) {
            const bool b = (lds_b >= 0) && (i & 1);
            hipLaunchKernelGGL(touch, dim3(256), dim3(b ? thr_b : thr_a), b ? lds_b : lds_a, 0, out, b ? lds_b : lds_a, spin);
        }
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
    }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1000.f / n;
}
int main()
{
    float* out;
    hipMalloc(&out, 1 << 20);
    hipFuncSetAttribute(reinterpret_cast<const void*>(touch), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int sizes[] = {0, 64 << 10, 144 << 10, 160 << 10};
    for (int spin : {0, 20000}) {
        printf("spin %d: back-to-back launches of 256 workgroups, us per launch\n", spin);
        for (int thr : {256, 512}) {
            for (int s : sizes) printf("  threads %d lds %6d : %7.2f\n", thr, s, run(400, s, thr, -1, 0, spin, out));
        }
        printf("  376-byte kernarg, 160K/512          : %7.2f\n", run_big<false>(400, 160 << 10, 512, spin, out));
        printf("  376-byte kernarg + 251 registers     : %7.2f\n", run_big<true>(400, 160 << 10, 512, spin, out));
        printf("  376-byte kernarg + regs, lds 0       : %7.2f\n", run_big<true>(400, 0, 512, spin, out));
        printf("  alternating 160K/512 with 0/256     : %7.2f\n", run(400, 160 << 10, 512, 0, 256, spin, out));
        printf("  alternating 160K/512 with 144K/512  : %7.2f\n", run(400, 160 << 10, 512, 144 << 10, 512, spin, out));
        printf("  alternating 144K/512 with 0/256     : %7.2f\n", run(400, 144 << 10, 512, 0, 256, spin, out));
    }
    return 0;
}
