// Probe (development aid): the instruction pattern of round 1's failing fused-upsampling blend (commit aa70900), isolated.
//   ds_write_b128 x2 (weight tile)  ->  ds_read2_b32 v[a:a+1], addr offset0:1 offset1:2 ; ds_read_b32 v[c], addr offset:12   (two items)
//   -> s_waitcnt lgkmcnt(2) -> v_pk_mul_f32 d[0:1], x[0:1], v[a:a+1] op_sel:[0,1]      (both halves must be x * v[a+1])
// In the failing kernel only the LOW half of that product was wrong, only in lanes 48..63, only with the chip loaded.
//   hipcc --offload-arch=gfx950 -O2 -o ds_read2_pk_probe tools/probes/ds_read2_pk_probe.hip && ./ds_read2_pk_probe
#include <hip/hip_runtime.h>
#include <cstdio>

typedef unsigned long long u64;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256, 2) void probe(unsigned* out, int iters)
{
    extern __shared__ __attribute__((aligned(16))) unsigned smem[];
    const int tid = threadIdx.x;
    unsigned* tbl = smem + 68224 / 4;            // 324 entries x 16 B at byte 68224, as in the failing kernel: {x, flags, lx, ly}
    for (int px = tid; px < 324; px += 256) {
        tbl[px * 4 + 0] = px; tbl[px * 4 + 1] = 7u;
        tbl[px * 4 + 2] = __float_as_uint(0.25f + 0.001f * px); tbl[px * 4 + 3] = __float_as_uint(0.5f + 0.001f * px);
    }
    __syncthreads();
    unsigned nbad_lo = 0, nbad_hi = 0, lanes = 0;
    const unsigned waddr = 51840 + (tid >> 2) * 64 + (tid & 3) * 16;
    for (int it = 0; it < iters; ++it) {
        const int pxa = (tid + 256 * ((it * 2) % 10)) >> 3, pxb = (tid + 256 * ((it * 2 + 1) % 10)) >> 3;
        const unsigned aa = 68224 + pxa * 16, ab = 68224 + pxb * 16;
        const u32x4 w0 = {(unsigned)it, (unsigned)tid, 3u, 4u}, w1 = {5u, 6u, (unsigned)it, (unsigned)tid};
        const float xl = 1.0f + (tid & 7), xh = 2.0f + (tid & 7);
        const u64 x = ((u64)__float_as_uint(xh) << 32) | __float_as_uint(xl);
        u64 pa, pb, da, db;
        float la, lb;
        asm volatile(
            "ds_write_b128 %6, %7\n\t"
            "ds_write_b128 %6, %8 offset:4096\n\t"
            "ds_read2_b32 %0, %9 offset0:1 offset1:2\n\t"
            "ds_read_b32 %2, %9 offset:12\n\t"
            "ds_read2_b32 %1, %10 offset0:1 offset1:2\n\t"
            "ds_read_b32 %3, %10 offset:12\n\t"
            "s_waitcnt lgkmcnt(2)\n\t"
            "v_pk_mul_f32 %4, %11, %0 op_sel:[0,1]\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_pk_mul_f32 %5, %11, %1 op_sel:[0,1]\n\t"
            : "=&v"(pa), "=&v"(pb), "=&v"(la), "=&v"(lb), "=&v"(da), "=&v"(db)
            : "v"(waddr), "v"(w0), "v"(w1), "v"(aa), "v"(ab), "v"(x)
            : "memory");
        const float lxa = __uint_as_float(tbl[pxa * 4 + 2]), lxb = __uint_as_float(tbl[pxb * 4 + 2]);   // plain reads: the expected weights
        const bool blo = __uint_as_float((unsigned)da) != xl * lxa || __uint_as_float((unsigned)db) != xl * lxb;
        const bool bhi = __uint_as_float((unsigned)(da >> 32)) != xh * lxa || __uint_as_float((unsigned)(db >> 32)) != xh * lxb;
        nbad_lo += blo; nbad_hi += bhi;
        if (blo || bhi) lanes |= 1u << ((tid & 63) >> 4);
        (void)la; (void)lb; (void)pa; (void)pb;
        __syncthreads();
    }
    if (nbad_lo) atomicAdd(&out[0], nbad_lo);
    if (nbad_hi) atomicAdd(&out[1], nbad_hi);
    if (lanes) atomicOr(&out[2], lanes);
}

int main()
{
    const int lds = 73408;
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    unsigned* d;
    hipMalloc(&d, 16);
    for (int grid : {64, 512, 4096}) {
        hipMemset(d, 0, 16);
        hipLaunchKernelGGL(probe, dim3(grid), dim3(256), lds, 0, d, 2000);
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
        unsigned h[4];
        hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("grid %5d x 2000 iterations: wrong low halves %u, wrong high halves %u, lane-group mask %x\n", grid, h[0], h[1], h[2]);
    }
    return 0;
}
