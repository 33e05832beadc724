// Probe: is the SCALAR offset of a raw buffer load part of the range check on gfx950?  A 4096-byte buffer descriptor over a larger
// allocation filled with 1.0f; lanes load a dword at voffset = 16 * lane with soffset in {0, 2048, 3072 + 1024 = 4096, 8192}.
// Prints what lane 0 and lane 63 get for every soffset, for the register form and for the LDS-DMA form (buffer_load ... lds).
//   hipcc --offload-arch=gfx950 soffset_range_probe.hip -o /tmp/soffset_probe && /tmp/soffset_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) void lds_void;
__global__ void probe(const float* buf, float* out, int soff)
{
    __shared__ float lds[64];
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)buf, 0, 4096, 0x00020000);
    const int lane = threadIdx.x;
    lds[lane] = -7.f;
    __syncthreads();
    const float v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, lane * 16, soff, 0));
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)lds, 4, lane * 16, soff, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    out[lane] = v;
    out[64 + lane] = lds[lane];
}
int main()
{
    float *buf, *out;
    hipMalloc(&buf, 1 << 20);
    hipMalloc(&out, 128 * 4);
    float* h = (float*)malloc(1 << 20);
    for (int i = 0; i < (1 << 18); ++i) h[i] = 1.f + i;
    hipMemcpy(buf, h, 1 << 20, hipMemcpyHostToDevice);
    for (int soff : {0, 2048, 3072, 3088, 4096, 8192}) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, buf, out, soff);
        float r[128];
        hipMemcpy(r, out, sizeof r, hipMemcpyDeviceToHost);
        // lane l reads byte soff + 16 l: in range iff soff + 16 l + 4 <= 4096 (when the scalar offset counts) / 16 l + 4 <= 4096 (when it does not)
        printf("soffset %5d: register form lane 0 %.0f lane 63 %.0f | lds form lane 0 %.0f lane 63 %.0f   (in-buffer values would be %d and %d)\n", soff, r[0], r[63],
               r[64], r[127], 1 + soff / 4, 1 + (soff + 63 * 16) / 4);
    }
    return 0;
}
