// Does hipExtAnyOrderLaunch let two kernels of ONE stream run side by side on this device/runtime?
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
__global__ void spin(long long cycles, unsigned *sink) {
    const long long t0 = wall_clock64();
    unsigned v = 0;
    while (wall_clock64() - t0 < cycles) v++;
    if (v == 0xFFFFFFFFu) *sink = v;
}
int main() {
    unsigned *d; (void)hipMalloc((void **)&d, 4);
    hipStream_t s; (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    for (int flags = 0; flags < 2; flags++) {
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, 1000LL, d);
        (void)hipStreamSynchronize(s);
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < 4; i++) hipExtLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, nullptr, nullptr, flags, 20000LL, d);
        hipError_t e = hipStreamSynchronize(s);
        printf("flags=%d: 4 x 200 us kernels on one stream: %.1f us (%s)\n", flags,
               std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(), hipGetErrorString(e));
    }
    return 0;
}
