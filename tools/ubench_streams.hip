// How many HIP streams of one process really run kernels side by side on this device?  Each stream gets one small
// busy-wait kernel (1 workgroup, ~200 us); if the streams map to distinct hardware queues the whole batch takes ~200 us.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/ubench_streams tools/ubench_streams.hip && /tmp/ubench_streams
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void spin(long long cycles, unsigned *sink) {
    const long long t0 = wall_clock64();
    unsigned v = 0;
    while (wall_clock64() - t0 < cycles) v++;
    if (v == 0xFFFFFFFFu) *sink = v;
}
static double run(const std::vector<hipStream_t> &st, long long cyc, unsigned *d) {
    for (auto s : st) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, 1000, d); // warm
    hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    for (auto s : st) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, cyc, d);
    hipDeviceSynchronize();
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
}
int main() {
    unsigned *d; hipMalloc((void **)&d, 4);
    int lo, hi; hipDeviceGetStreamPriorityRange(&lo, &hi);
    printf("priority range: least %d greatest %d\n", lo, hi);
    const long long cyc = 20000; // wall_clock64 ticks at 100 MHz -> 200 us
    for (int n : {1, 2, 3, 4, 5, 6, 8, 10, 12}) {
        std::vector<hipStream_t> st(n);
        for (auto &s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        printf("%2d streams, one priority : %8.1f us\n", n, run(st, cyc, d));
        for (auto s : st) hipStreamDestroy(s);
    }
    for (int n : {3, 6, 9, 12}) {
        std::vector<hipStream_t> st(n);
        for (int i = 0; i < n; i++) hipStreamCreateWithPriority(&st[i], hipStreamNonBlocking, hi + (i % (lo - hi + 1)));
        printf("%2d streams, %d priorities: %8.1f us\n", n, lo - hi + 1, run(st, cyc, d));
        for (auto s : st) hipStreamDestroy(s);
    }
    return 0;
}
