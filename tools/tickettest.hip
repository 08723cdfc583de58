// tickettest.hip -- what a per-tile ticket costs: 16384 workgroups of 256 threads that do nothing but take a number from ONE device-scope
// counter (the one-pass radix scatter's tile ticket, radix.hip), against the same grid without the atomic and with the atomic spread over 8 / 64
// counters; and the same with a dependent global load + store behind the ticket (the tile's first real use of its number).
//   hipcc --offload-arch=gfx950 -O3 tools/tickettest.hip -o tools/_bin/tickettest && tools/_bin/tickettest
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int NCOUNTERS, bool LOADS>
__global__ __launch_bounds__(256) void k_ticket(uint32_t *tickets, const uint64_t *in, uint64_t *out, uint32_t ntiles)
{
    __shared__ uint32_t s_tile;
    if (threadIdx.x == 0) {
        if (NCOUNTERS == 0) s_tile = blockIdx.x;
        else {
            const uint32_t c = blockIdx.x % (uint32_t)NCOUNTERS;
            s_tile = atomicAdd(&tickets[c * 64], 1u) * (uint32_t)NCOUNTERS + c;
        }
    }
    __syncthreads();
    const uint32_t tile = s_tile;
    if (LOADS) {
        uint64_t acc = 0;
        const uint64_t *p = in + (size_t)(tile % ntiles) * 4096;
#pragma unroll
        for (int k = 0; k < 16; k++) acc += p[k * 256 + threadIdx.x];
        out[(size_t)(tile % ntiles) * 256 + threadIdx.x] = acc;
    } else if (threadIdx.x == 0) out[tile % ntiles] = tile;
}

template <int NC, bool LOADS>
int run(const char *what, uint32_t *d_t, const uint64_t *d_in, uint64_t *d_out, uint32_t ntiles)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e9f, sum = 0;
    for (int rep = 0; rep < 12; rep++) {
        CK(hipMemsetAsync(d_t, 0, 64 * 64 * 4, 0));
        CK(hipEventRecord(a, 0));
        hipLaunchKernelGGL((k_ticket<NC, LOADS>), dim3(ntiles), dim3(256), 0, 0, d_t, d_in, d_out, ntiles);
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (rep >= 2) { sum += ms; if (ms < best) best = ms; }
    }
    printf("%-64s %8.1f us (best %7.1f)  = %6.1f ns per workgroup\n", what, sum / 10 * 1e3, best * 1e3, sum / 10 * 1e6 / ntiles);
    return 0;
}

int main()
{
    const uint32_t ntiles = 16384;
    uint32_t *d_t; uint64_t *d_in, *d_out;
    CK(hipMalloc(&d_t, 64 * 64 * 4));
    CK(hipMalloc(&d_in, (size_t)ntiles * 4096 * 8));
    CK(hipMalloc(&d_out, (size_t)ntiles * 256 * 8));
    CK(hipMemset(d_in, 1, (size_t)ntiles * 4096 * 8));
    printf("# %u workgroups of 256 threads per launch (one 64 MiB block = 16384 tiles per radix pass)\n", ntiles);
    if (run<0, false>("no ticket (tile = blockIdx.x), one word stored", d_t, d_in, d_out, ntiles)) return 1;
    if (run<1, false>("ONE ticket counter (atomicAdd, device scope)", d_t, d_in, d_out, ntiles)) return 1;
    if (run<8, false>("8 counters (blockIdx mod 8)", d_t, d_in, d_out, ntiles)) return 1;
    if (run<64, false>("64 counters", d_t, d_in, d_out, ntiles)) return 1;
    if (run<0, true>("no ticket, 32 KB loaded + 2 KB stored per workgroup", d_t, d_in, d_out, ntiles)) return 1;
    if (run<1, true>("ONE ticket counter, then the 32 KB load (address from the ticket)", d_t, d_in, d_out, ntiles)) return 1;
    if (run<8, true>("8 counters, then the load", d_t, d_in, d_out, ntiles)) return 1;
    return 0;
}
