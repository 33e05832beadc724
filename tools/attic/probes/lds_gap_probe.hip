// Probe: cost of back-to-back launches of a one-workgroup-per-CU kernel as a function of its dynamic LDS size.
// (profiles/r05_step_launch_list.txt: 5-6 us of idle time on either side of every 160 KB kernel, none around the 144 KB one.)
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O2 tools/probes/lds_gap_probe.hip -o /tmp/lds_gap && /tmp/lds_gap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

extern __shared__ char smem[];

__global__ __launch_bounds__(512) void touch_kernel(float* out, int lds_bytes, int spin)
{
    volatile char* s = smem;
    s[(threadIdx.x * 64) % lds_bytes] = (char)threadIdx.x;
    __syncthreads();
    float v = s[(threadIdx.x * 128 + 7) % lds_bytes];
    for (int i = 0; i < spin; ++i) v = v * 1.0001f + 0.5f;
    if (threadIdx.x == 0) out[blockIdx.x] = v;
}

int main()
{
    float* out;
    hipMalloc(&out, 4096 * sizeof(float));
    hipStream_t st;
    hipStreamCreate(&st);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipFuncSetAttribute(reinterpret_cast<const void*>(touch_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int sizes[] = {0 + 1024, 32 * 1024, 64 * 1024, 65 * 1024, 96 * 1024, 128 * 1024, 144 * 1024, 152 * 1024, 156 * 1024, 158 * 1024, 159 * 1024, 160 * 1024 - 512, 160 * 1024};
    for (int spin : {0, 20000})
        for (int grid : {256, 2048})
            for (int lds : sizes) {
                for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(touch_kernel, dim3(grid), dim3(512), lds, st, out, lds, spin);
                hipStreamSynchronize(st);
                const int n = 200;
                hipEventRecord(e0, st);
                for (int i = 0; i < n; ++i) hipLaunchKernelGGL(touch_kernel, dim3(grid), dim3(512), lds, st, out, lds, spin);
                hipEventRecord(e1, st);
                hipEventSynchronize(e1);
                float ms = 0;
                hipEventElapsedTime(&ms, e0, e1);
                printf("spin %6d grid %5d lds %7d B : %8.2f us per launch\n", spin, grid, lds, ms * 1000.f / n);
            }
    // alternating: a big-LDS kernel followed by a small one
    for (int lds : {144 * 1024, 160 * 1024}) {
        const int n = 200;
        hipEventRecord(e0, st);
        for (int i = 0; i < n; ++i) {
            hipLaunchKernelGGL(touch_kernel, dim3(256), dim3(512), lds, st, out, lds, 20000);
            hipLaunchKernelGGL(touch_kernel, dim3(2048), dim3(512), 1024, st, out, 1024, 2000);
        }
        hipEventRecord(e1, st);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        printf("alternating big %7d B + small: %8.2f us per pair\n", lds, ms * 1000.f / n);
    }
    return 0;
}
