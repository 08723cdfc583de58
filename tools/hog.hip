// hog.hip -- co-running "resource hogs" for the interference matrix (tools/interfere.py): what does the compress loop compete for?
// A hog is a persistent kernel on a stream of its own that keeps `wgs_per_cu` workgroups of 256 threads per CU busy with ONE kind
// of work until the host raises a flag:
//   0 valu    dependent + independent integer multiply-adds (issue slots of the SIMDs, no memory)
//   1 lds     ds_read / ds_write of a private 16 KB region, conflict-free (LDS bandwidth)
//   2 stream  16-byte coalesced reads of a 2 GiB buffer (HBM / fabric streaming bandwidth)
//   3 gather  random 4-byte reads from a 256 MiB table (line-granular random accesses)
//   4 sleep   s_sleep in a loop (holds the wave slots and nothing else)
// build:  hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/hog.hip -o tools/_bin/libhog.so
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

namespace {
// The stop flag lives in DEVICE memory (the host raises it with a 4-byte copy on a second stream) and is polled by one lane per
// workgroup: 131 072 threads polling a pinned HOST word over PCIe starved the command processor's own reads of the AQL packets and
// kernel arguments of every other stream -- the first version of this tool measured that, not interference (profiles/r04_interference.txt).
__device__ __forceinline__ bool stopped(const volatile uint32_t *flag) { return __hip_atomic_load(const_cast<const uint32_t *>(flag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u; }

__global__ __launch_bounds__(256) void k_hog(int kind, const volatile uint32_t *flag, const uint4 *big, size_t big_n16, const uint32_t *table, uint32_t table_mask,
                                            unsigned long long *work)
{
    __shared__ uint32_t lds[4096];
    __shared__ uint32_t s_stop;
    uint32_t x = threadIdx.x * 2654435761u + blockIdx.x, y = x ^ 0x9E3779B9u, z = 0, w = 1;
    unsigned long long done = 0;
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = i;
    __syncthreads();
    size_t pos = ((size_t)blockIdx.x * 256 + threadIdx.x) % big_n16;
    const uint64_t t_start = __builtin_amdgcn_s_memrealtime();     // 100 MHz: a hog never outlives 6 s, whatever happens to the host
    for (;;) {
        if (kind == 0) {
#pragma unroll
            for (int k = 0; k < 256; k++) { x = x * 3u + y; y = y * 5u + 1u; z = z * 7u + w; w = w * 9u + 3u; }
            done += 1024;
        } else if (kind == 1) {
#pragma unroll
            for (int k = 0; k < 64; k++) { const uint32_t a = (threadIdx.x + 256 * (k & 15)) & 4095; x += lds[a]; lds[a] = x; }
            done += 128;
        } else if (kind == 2) {
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const uint4 v = big[pos];
                x += v.x ^ v.y ^ v.z ^ v.w;
                pos += (size_t)gridDim.x * 256;
                if (pos >= big_n16) pos -= big_n16;
            }
            done += 16 * 16;
        } else if (kind == 3) {
#pragma unroll
            for (int k = 0; k < 16; k++) { x = x * 1664525u + 1013904223u; y += table[(x >> 4) & table_mask]; }
            done += 16;
        } else {
            __builtin_amdgcn_s_sleep(64);
            done += 1;
        }
        if (threadIdx.x == 0) s_stop = (stopped(flag) || __builtin_amdgcn_s_memrealtime() - t_start > 600000000ull) ? 1u : 0u;
        __syncthreads();
        const uint32_t stop = s_stop;
        __syncthreads();
        if (stop) break;
    }
    if ((x ^ y ^ z ^ w) == 0x12345u) lds[0] = x;                  // keep the work alive
    atomicAdd(work, done);
    if (lds[0] == 0xFFFFFFFFu) atomicAdd(work, 1ull);
}

struct Hog {
    hipStream_t stream = nullptr, ctl = nullptr;
    uint32_t *flag = nullptr;           // device memory
    uint4 *big = nullptr;
    uint32_t *table = nullptr;
    unsigned long long *work = nullptr;
    size_t big_n16 = 0;
    int cus = 0;
} H;
}  // namespace

extern "C" int hog_init(void)
{
    if (H.stream) return 0;
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, 0) != hipSuccess) return -1;
    H.cus = p.multiProcessorCount;
    if (hipStreamCreateWithFlags(&H.stream, hipStreamNonBlocking) != hipSuccess) return -2;
    if (hipStreamCreateWithFlags(&H.ctl, hipStreamNonBlocking) != hipSuccess) return -2;
    if (hipMalloc((void **)&H.flag, 64) != hipSuccess) return -3;
    H.big_n16 = ((size_t)2 << 30) / 16;
    if (hipMalloc((void **)&H.big, H.big_n16 * 16) != hipSuccess) return -4;
    if (hipMalloc((void **)&H.table, (size_t)256 << 20) != hipSuccess) return -5;
    if (hipMalloc((void **)&H.work, 8) != hipSuccess) return -6;
    hipMemset(H.big, 1, H.big_n16 * 16);
    hipMemset(H.table, 2, (size_t)256 << 20);
    hipDeviceSynchronize();
    return H.cus;
}

extern "C" int hog_start(int kind, int wgs_per_cu)
{
    (void)hipMemsetAsync(H.flag, 0, 4, H.stream);
    (void)hipMemsetAsync(H.work, 0, 8, H.stream);
    uint32_t *dflag = H.flag;
    hipLaunchKernelGGL(k_hog, dim3(H.cus * wgs_per_cu), dim3(256), 0, H.stream, kind, dflag, H.big, H.big_n16, H.table, (uint32_t)((64u << 20) - 1), H.work);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// stops the hog; returns its work units (valu: multiply-adds per lane / lds: accesses per lane / stream: bytes per lane / gather: accesses per lane)
extern "C" double hog_stop(void)
{
    static const uint32_t one = 1u;
    (void)hipMemcpyAsync(H.flag, &one, 4, hipMemcpyHostToDevice, H.ctl);
    (void)hipStreamSynchronize(H.ctl);
    (void)hipStreamSynchronize(H.stream);
    unsigned long long w = 0;
    hipMemcpy(&w, H.work, 8, hipMemcpyDeviceToHost);
    return (double)w;
}
