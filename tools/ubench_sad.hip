// Micro-benchmark: issue rate of the byte-SAD instructions on gfx950 (one wave per SIMD and 2/4 waves per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define N 4096
template <int MODE>
__global__ void k(const uint32_t *in, uint64_t *out) {
    uint32_t a = in[threadIdx.x], b = in[threadIdx.x + 64], c = in[threadIdx.x + 128];
    uint64_t acc[8];
    uint32_t acc32[8];
    for (int i = 0; i < 8; i++) { acc[i] = 0; acc32[i] = 0; }
    for (int it = 0; it < N; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (MODE == 0) acc[i] = __builtin_amdgcn_qsad_pk_u16_u8(((uint64_t)b << 32) | (a + i), c, acc[i]);
            if (MODE == 1) acc32[i] = __builtin_amdgcn_sad_u8(a + i, c, acc32[i]);
            if (MODE == 2) { uint32_t s = __builtin_amdgcn_alignbyte(b, a + i, 1); acc32[i] = __builtin_amdgcn_sad_u8(s, c, acc32[i]); }
            if (MODE == 3) acc32[i] = __builtin_amdgcn_sad_u16(a + i, c, acc32[i]);
            if (MODE == 4) acc[i] = __builtin_amdgcn_mqsad_pk_u16_u8(((uint64_t)b << 32) | (a + i), c, acc[i]);
            if (MODE == 5) acc32[i] = __builtin_amdgcn_msad_u8(a + i, c, acc32[i]);
        }
    }
    uint64_t r = 0;
    for (int i = 0; i < 8; i++) r += acc[i] + acc32[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int MODE>
void run(const char *name, int absdiff_per_lane, uint32_t *in, uint64_t *out) {
    for (int wps = 1; wps <= 4; wps *= 2) {
        int blocks = 256 * 4, threads = 64 * wps; // 4 blocks per CU, wps waves each -> wps waves per SIMD
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, in, out);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, in, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double instr_per_simd = (double)N * 8 * wps; // wave-instructions per SIMD (1 block per SIMD x wps waves)
        double ns_per_instr = ms * 1e6 / instr_per_simd;
        double total_absdiff = (double)blocks * threads * N * 8 * absdiff_per_lane;
        printf("%-28s waves/SIMD=%d  %.3f ms  %.2f ns per wave-instr per SIMD (~%.1f cyc @2.4GHz)  %.1f Tabsdiff/s\n", name, wps, ms,
               ns_per_instr, ns_per_instr * 2.4, total_absdiff / (ms * 1e-3) / 1e12);
    }
}
int main() {
    uint32_t *in; uint64_t *out;
    hipMalloc(&in, 4096); hipMalloc(&out, 256 * 4 * 256 * 8);
    hipMemset(in, 0x5a, 4096);
    run<0>("v_qsad_pk_u16_u8", 16, in, out);
    run<1>("v_sad_u8", 4, in, out);
    run<2>("v_alignbyte+v_sad_u8", 4, in, out);
    run<3>("v_sad_u16", 2, in, out);
    run<4>("v_mqsad_pk_u16_u8", 16, in, out);
    run<5>("v_msad_u8", 4, in, out);
    return 0;
}
