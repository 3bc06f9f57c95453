// How long does the HOST spend inside hipMemcpyAsync (page-locked -> device, non-blocking stream) for different source
// offsets and sizes?  A call that returns only when the copy is done cannot overlap with the copy down of another stream
// (hutk_api.cpp, encode_batch_pipelined).   hipcc -O2 tools/memcpy_async_probe.cpp -o /tmp/memcpy_async_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t N = 600u << 20;
    uint8_t *h, *d;
    hipHostMalloc((void**)&h, N, hipHostMallocDefault);
    hipMalloc((void**)&d, N);
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    struct { size_t off, size; const char* what; } cases[] = {
        {0, 63397960, "offset 0, 63397960 B"}, {0, 67108864, "offset 0, 64 MiB"}, {0, 67000001, "offset 0, 67000001 B (odd)"},
        {1, 63397960, "offset 1"}, {3, 63397959, "offset 3, odd size"}, {16, 63397952, "offset 16, size % 64 == 0"},
        {64, 33554432, "offset 64, 32 MiB"}, {7, 33554433, "offset 7, 32 MiB + 1"}, {4096, 8388608, "offset 4096, 8 MiB"}, {5, 8388611, "offset 5, 8 MiB + 3"},
    };
    for (auto& c : cases) {
        hipDeviceSynchronize();
        double best_call = 1e9, best_total = 1e9;
        for (int r = 0; r < 3; r++) {
            const double t0 = now();
            hipMemcpyAsync(d + 64, h + c.off, c.size, hipMemcpyHostToDevice, s);
            const double t1 = now();
            hipStreamSynchronize(s);
            const double t2 = now();
            if (t1 - t0 < best_call) best_call = t1 - t0;
            if (t2 - t0 < best_total) best_total = t2 - t0;
        }
        printf("%-34s call returns after %7.3f ms, copy done after %7.3f ms (%.1f GB/s)\n", c.what, best_call, best_total, c.size / best_total / 1e6);
    }
    // the same towards the host
    for (auto& c : cases) {
        hipDeviceSynchronize();
        double best_call = 1e9, best_total = 1e9;
        for (int r = 0; r < 3; r++) {
            const double t0 = now();
            hipMemcpyAsync(h + c.off, d + 64, c.size, hipMemcpyDeviceToHost, s);
            const double t1 = now();
            hipStreamSynchronize(s);
            const double t2 = now();
            if (t1 - t0 < best_call) best_call = t1 - t0;
            if (t2 - t0 < best_total) best_total = t2 - t0;
        }
        printf("D2H %-30s call returns after %7.3f ms, copy done after %7.3f ms (%.1f GB/s)\n", c.what, best_call, best_total, c.size / best_total / 1e6);
    }
    return 0;
}
