// ranstest.hip -- cycles per dependent rANS encoder step, registers only (no memory in the loop)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__device__ __forceinline__ uint32_t rans_step(uint32_t x, uint32_t lf, uint32_t rcp, uint32_t &e)
{
    const uint32_t lo = lf & 0xffffu, fr = lf >> 16;
    const uint32_t xmax = fr << 15;
    const bool b1 = x >= xmax;
    uint32_t em = b1 ? ((x & 0xffu) | (1u << 16)) : 0u;
    x = b1 ? (x >> 8) : x;
    const bool b2 = x >= xmax;
    em = b2 ? ((em & 0xffu) | ((x & 0xffu) << 8) | (2u << 16)) : em;
    x = b2 ? (x >> 8) : x;
    const uint32_t qm = __umulhi(x, rcp) >> ((31 - __clz((int)((fr - 1) | 1u))) & 31);
    const uint32_t q = (fr >= 2) ? qm : x;
    e = em;
    return x + lo + q * (65536u - fr);
}
__global__ void k(uint32_t *out, const uint2 *recs, int iters, uint64_t *clk)
{
    uint2 r[8];
    for (int i = 0; i < 8; i++) r[i] = recs[threadIdx.x * 8 + i];
    uint32_t x = 1u << 23, acc = 0;
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++) { uint32_t e; x = rans_step(x, r[j].x, r[j].y, e); acc += e; }
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = x + acc;
    if (threadIdx.x == 0) clk[0] = t1 - t0;
}
int main()
{
    uint2 h[512];
    for (int i = 0; i < 512; i++) {
        uint32_t fr = 1000 + (i * 7919) % 30000, lo = (i * 31) % 20000;
        int sh = 32 - __builtin_clz(fr - 1);
        uint32_t rcp = (uint32_t)((((uint64_t)1 << (sh + 31)) + fr - 1) / fr);
        h[i].x = lo | (fr << 16); h[i].y = rcp;
    }
    uint2 *d; uint32_t *o; uint64_t *c, hc;
    hipMalloc(&d, sizeof h); hipMalloc(&o, 1024); hipMalloc(&c, 8);
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    const int iters = 200000;
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, d, iters, c);
        hipDeviceSynchronize();
        hipMemcpy(&hc, c, 8, hipMemcpyDeviceToHost);
        printf("cycles per step: %.1f\n", (double)hc / (iters * 8.0));
    }
    return 0;
}
