// pcietest.hip -- host<->device copy rates that bound the drop-in (host-buffer) path: pageable vs pinned, and host memcpy
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    const size_t n = 64u << 20;
    unsigned char *pg = (unsigned char *)malloc(n), *pg2 = (unsigned char *)malloc(n), *pin, *dev;
    memset(pg, 1, n); memset(pg2, 2, n);
    hipHostMalloc((void **)&pin, n, hipHostMallocDefault);
    hipMalloc((void **)&dev, n);
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    memset(pin, 3, n);
    auto run = [&](const char *name, auto fn) { fn(); double t0 = now(); for (int i = 0; i < 5; i++) fn(); double dt = (now() - t0) / 5; printf("%-44s %7.2f ms  %6.2f GB/s\n", name, dt * 1e3, n / dt / 1e9); };
    run("H2D pageable hipMemcpyAsync + sync", [&] { hipMemcpyAsync(dev, pg, n, hipMemcpyHostToDevice, st); hipStreamSynchronize(st); });
    run("D2H pageable hipMemcpyAsync + sync", [&] { hipMemcpyAsync(pg, dev, n, hipMemcpyDeviceToHost, st); hipStreamSynchronize(st); });
    run("H2D pinned", [&] { hipMemcpyAsync(dev, pin, n, hipMemcpyHostToDevice, st); hipStreamSynchronize(st); });
    run("D2H pinned", [&] { hipMemcpyAsync(pin, dev, n, hipMemcpyDeviceToHost, st); hipStreamSynchronize(st); });
    run("host memcpy pageable -> pinned", [&] { memcpy(pin, pg, n); });
    run("host memcpy pinned -> pageable", [&] { memcpy(pg2, pin, n); });
    run("H2D pageable via pinned, 4 MiB chunks piped", [&] {
        const size_t c = 4u << 20; hipEvent_t ev[2]; hipEventCreate(&ev[0]); hipEventCreate(&ev[1]);
        unsigned char *p2[2] = {pin, pin + (32u << 20)};
        for (size_t o = 0, k = 0; o < n; o += c, k++) { if (k >= 2) hipEventSynchronize(ev[k & 1]); memcpy(p2[k & 1], pg + o, c); hipMemcpyAsync(dev + o, p2[k & 1], c, hipMemcpyHostToDevice, st); hipEventRecord(ev[k & 1], st); }
        hipStreamSynchronize(st); hipEventDestroy(ev[0]); hipEventDestroy(ev[1]); });
    run("hipHostRegister + H2D + unregister", [&] { hipHostRegister(pg, n, hipHostRegisterDefault); hipMemcpyAsync(dev, pg, n, hipMemcpyHostToDevice, st); hipStreamSynchronize(st); hipHostUnregister(pg); });
    return 0;
}
