// dev tool: which compute units does a stream created with hipExtStreamCreateWithCUMask use?  Every workgroup records (XCC_ID, HW_ID); the host prints, per mask, how many
// distinct (xcc, se, cu) ran workgroups and which mask bits correspond to which (xcc, se, cu).    hipcc --offload-arch=gfx950 -O2 tools/ubench_cumask.hip -o tools/ubench_cumask
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <set>
#include <map>
#include <vector>
__global__ void who(unsigned *out) {
    if (threadIdx.x == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc;
        for (int i = 0; i < 2000; i++) __builtin_amdgcn_s_sleep(20); // long enough for the whole grid to spread
    }
}
static void run(const char *what, const uint32_t *mask, int words, unsigned *d, std::vector<unsigned> &h, int n) {
    hipStream_t s;
    if (mask) { if (hipExtStreamCreateWithCUMask(&s, words, mask) != hipSuccess) { printf("%s: create failed\n", what); return; } }
    else hipStreamCreate(&s);
    hipMemsetAsync(d, 0, n * 8, s);
    hipLaunchKernelGGL(who, dim3(n), dim3(64), 0, s, d);
    hipStreamSynchronize(s);
    hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost);
    std::map<unsigned, std::set<unsigned>> per;
    for (int i = 0; i < n; i++) { const unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 15; per[xcc].insert(((hw >> 13) & 7) << 8 | ((hw >> 12) & 1) << 4 | ((hw >> 8) & 15)); }
    int tot = 0; printf("%-34s", what);
    for (auto &p : per) { printf(" xcc%u:%zu", p.first, p.second.size()); tot += (int)p.second.size(); }
    printf("  total %d\n", tot);
    hipStreamDestroy(s);
}
int main() {
    const int n = 4096;
    unsigned *d; hipMalloc(&d, n * 8);
    std::vector<unsigned> h(2 * n);
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    const int ncu = pr.multiProcessorCount, words = (ncu + 31) / 32;
    printf("%d compute units\n", ncu);
    uint32_t full[16]; memset(full, 0, sizeof full);
    for (int i = 0; i < ncu; i++) full[i >> 5] |= 1u << (i & 31);
    run("no mask", nullptr, 0, d, h, n);
    run("full mask", full, words, d, h, n);
    uint32_t m[16];
    memcpy(m, full, sizeof m); for (int k = 0; k < 40; k++) m[k >> 5] &= ~(1u << (k & 31));
    run("first 40 bits cleared", m, words, d, h, n);
    memcpy(m, full, sizeof m); for (int k = 0; k < 40; k++) { int i = k * ncu / 40; m[i >> 5] &= ~(1u << (i & 31)); }
    run("every 6.4th bit cleared (40)", m, words, d, h, n);
    memset(m, 0, sizeof m); for (int k = 0; k < 8; k++) m[0] |= 1u << k;
    run("only bits 0..7", m, words, d, h, n);
    memset(m, 0, sizeof m); m[0] = 0xFFFFFFFFu;
    run("only bits 0..31", m, words, d, h, n);
    memset(m, 0, sizeof m); m[0] = 1u;
    run("only bit 0", m, words, d, h, n);
    memset(m, 0, sizeof m); m[0] = 1u << 8;
    run("only bit 8", m, words, d, h, n);
    return 0;
}
