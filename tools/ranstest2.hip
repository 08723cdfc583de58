// ranstest2.hip -- cycles per rANS encoder step with per-step records read from LDS (as k_rans_lanes does)
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
__global__ void k(uint32_t *out, const uint2 *recs, int iters, uint64_t *clk, int active)
{
    __shared__ uint2 rb[4][128];
    __shared__ uint32_t eb[4][128];
    for (int i = threadIdx.x; i < 512; i += 64) (&rb[0][0])[i] = recs[i];
    __syncthreads();
    uint32_t x = 1u << 23;
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x < active) {
        const uint2 *r = &rb[threadIdx.x & 3][0];
        uint32_t *ebp = &eb[threadIdx.x & 3][0];
        for (int it = 0; it < iters; it++) {
            for (int k = 127; k >= 3; k -= 4) {
                const uint2 r0 = r[k], r1 = r[k - 1], r2 = r[k - 2], r3 = r[k - 3];
                uint32_t e0, e1, e2, e3;
                x = rans_step(x, r0.x, r0.y, e0); x = rans_step(x, r1.x, r1.y, e1);
                x = rans_step(x, r2.x, r2.y, e2); x = rans_step(x, r3.x, r3.y, e3);
                ebp[k] = e0; ebp[k - 1] = e1; ebp[k - 2] = e2; ebp[k - 3] = e3;
            }
        }
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = x + eb[0][threadIdx.x];
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
    const int iters = 2000;
    for (int active : {4, 64}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, d, iters, c, active);
        hipDeviceSynchronize();
        hipMemcpy(&hc, c, 8, hipMemcpyDeviceToHost);
        printf("active %d: cycles per step: %.1f\n", active, (double)hc / (iters * 128.0));
    }
    return 0;
}
