// Probe (development aid): do the DS read forms agree on LDS addresses above 64 KB?  Round 1's fused up-sampling table lived at
// byte 68224 of a 73408-byte workgroup allocation; a second read of an entry, which hipcc lowered to
// `ds_read2_b32 v, vaddr offset0:1 offset1:2` + `ds_read_b32 v, vaddr offset:12`, returned wrong data, while ds_read_b64 /
// ds_read_b128 of the same address did not.  Every LDS dword holds its own dword index, so each form's result is checkable.
//   hipcc --offload-arch=gfx950 -O2 -o lds_read2_probe tools/probes/lds_read2_probe.hip && ./lds_read2_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int LDS_BYTES = 73408;
constexpr int NRES = 12;

__global__ __launch_bounds__(256, 2) void probe(unsigned* out, int base_bytes)
{
    extern __shared__ __attribute__((aligned(16))) unsigned smem[];
    for (int i = threadIdx.x; i < LDS_BYTES / 4; i += blockDim.x) smem[i] = (unsigned)i;
    __syncthreads();
    const unsigned addr = (unsigned)(base_bytes + threadIdx.x * 16);    // byte address inside the allocation
    unsigned r[NRES];
    asm volatile("ds_read_b32 %0, %1 offset:12\n\ts_waitcnt lgkmcnt(0)" : "=v"(r[0]) : "v"(addr) : "memory");
    unsigned long long p;
    asm volatile("ds_read2_b32 %0, %1 offset0:1 offset1:2\n\ts_waitcnt lgkmcnt(0)" : "=v"(p) : "v"(addr) : "memory");
    r[1] = (unsigned)p; r[2] = (unsigned)(p >> 32);
    uint4 q;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(q) : "v"(addr) : "memory");
    r[3] = q.x; r[4] = q.y; r[5] = q.z; r[6] = q.w;
    asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(p) : "v"(addr) : "memory");
    r[7] = (unsigned)p; r[8] = (unsigned)(p >> 32);
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r[9]) : "v"(addr) : "memory");
    asm volatile("ds_read2_b32 %0, %1 offset0:0 offset1:3\n\ts_waitcnt lgkmcnt(0)" : "=v"(p) : "v"(addr) : "memory");
    r[10] = (unsigned)p; r[11] = (unsigned)(p >> 32);
    for (int k = 0; k < NRES; ++k) out[((size_t)blockIdx.x * blockDim.x + threadIdx.x) * NRES + k] = r[k];
}

int main()
{
    const char* names[NRES] = {"ds_read_b32 offset:12", "ds_read2_b32 offset0:1 .lo", "ds_read2_b32 offset1:2 .hi", "ds_read_b128 .x", "ds_read_b128 .y",
                               "ds_read_b128 .z", "ds_read_b128 .w", "ds_read_b64 .lo", "ds_read_b64 .hi", "ds_read_b32", "ds_read2_b32 offset0:0 .lo",
                               "ds_read2_b32 offset1:3 .hi"};
    const int expect_off[NRES] = {3, 1, 2, 0, 1, 2, 3, 0, 1, 0, 0, 3};
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    const int grids[] = {16, 256, 512, 4096};
    const int bases[] = {1024, 61440, 65536 - 4096, 65536, 68224};
    for (int g : grids)
        for (int base : bases) {
            unsigned* d;
            const size_t n = (size_t)g * 256 * NRES;
            hipMalloc(&d, n * 4);
            hipMemset(d, 0xff, n * 4);
            hipLaunchKernelGGL(probe, dim3(g), dim3(256), LDS_BYTES, 0, d, base);
            if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
            std::vector<unsigned> h(n);
            hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
            hipFree(d);
            long bad[NRES] = {0};
            unsigned sample[NRES] = {0}, sample_want[NRES] = {0};
            for (int b = 0; b < g; ++b)
                for (int t = 0; t < 256; ++t)
                    for (int k = 0; k < NRES; ++k) {
                        const unsigned want = (unsigned)((base + t * 16) / 4 + expect_off[k]);
                        const unsigned got = h[((size_t)b * 256 + t) * NRES + k];
                        if (got != want) { if (!bad[k]) { sample[k] = got; sample_want[k] = want; } ++bad[k]; }
                    }
            printf("grid %5d  base %6d (last byte %6d):", g, base, base + 255 * 16 + 15);
            bool any = false;
            for (int k = 0; k < NRES; ++k)
                if (bad[k]) { any = true; printf("\n    %-28s wrong in %ld of %d reads (e.g. got dword %u, want %u)", names[k], bad[k], g * 256, sample[k], sample_want[k]); }
            printf(any ? "\n" : " all forms agree\n");
        }
    return 0;
}
