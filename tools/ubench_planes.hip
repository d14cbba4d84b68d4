// Self-check of sp_planes() (k_motion.hip): the packed / dot-product forms of the three 6-tap planes against the
// sample-by-sample definitions of 8.4.2.2.1, on random 23x23 neighbourhoods.
//   hipcc --offload-arch=gfx950 -O2 -Iinclude -o tools/ubench_planes tools/ubench_planes.hip && tools/ubench_planes
#include "../ceracoder_amd/csrc/k_motion.hip"
#include <cstdio>
#include <cstdlib>
__global__ void planes_kernel(const uint8_t *g, uint8_t *ob, uint8_t *oh, uint8_t *oj) {
    __shared__ __attribute__((aligned(16))) sp_lds L;
    const int lane = threadIdx.x;
    for (int i = lane; i < 23 * SP_GS; i += 64) L.G[i] = g[blockIdx.x * 23 * SP_GS + i];
    WAVE_SYNC();
    sp_planes(&L, lane);
    for (int i = lane; i < 19 * 18; i += 64) ob[blockIdx.x * 19 * 18 + i] = L.b[(i / 18) * SP_PS + i % 18];
    for (int i = lane; i < 18 * 19; i += 64) oh[blockIdx.x * 18 * 19 + i] = L.h[(i / 19) * SP_GS + i % 19 + 2];
    for (int i = lane; i < 18 * 18; i += 64) oj[blockIdx.x * 18 * 18 + i] = L.j[(i / 18) * SP_PS + i % 18];
}
static int t6(int a, int b, int c, int d, int e, int f) { return a - 5 * b + 20 * c + 20 * d - 5 * e + f; }
static int c255(int v) { return v < 0 ? 0 : v > 255 ? 255 : v; }
int main() {
    const int N = 64;
    uint8_t *g, *b, *h, *j;
    hipMallocManaged((void **)&g, N * 23 * SP_GS); hipMallocManaged((void **)&b, N * 19 * 18); hipMallocManaged((void **)&h, N * 18 * 19); hipMallocManaged((void **)&j, N * 18 * 18);
    srand(5);
    for (int i = 0; i < N * 23 * SP_GS; i++) g[i] = (i / (23 * SP_GS)) % 3 == 0 ? (rand() & 1 ? 255 : 0) : rand() & 255;
    hipLaunchKernelGGL(planes_kernel, dim3(N), dim3(64), 0, 0, g, b, h, j);
    hipDeviceSynchronize();
    int bad[3] = {0, 0, 0};
    for (int n = 0; n < N; n++) {
        const uint8_t *G = g + n * 23 * SP_GS;
#define GG(r, c) G[(r) * SP_GS + (c)]
        for (int R = 0; R < 19; R++) for (int C = 0; C < 18; C++) {
            int v = c255((t6(GG(R + 2, C), GG(R + 2, C + 1), GG(R + 2, C + 2), GG(R + 2, C + 3), GG(R + 2, C + 4), GG(R + 2, C + 5)) + 16) >> 5);
            if (v != b[n * 19 * 18 + R * 18 + C] && bad[0]++ < 5) printf("b[%d][%d][%d] = %d, want %d\n", n, R, C, b[n * 19 * 18 + R * 18 + C], v);
        }
        for (int R = 0; R < 18; R++) for (int C = 0; C < 19; C++) {
            int v = c255((t6(GG(R, C + 2), GG(R + 1, C + 2), GG(R + 2, C + 2), GG(R + 3, C + 2), GG(R + 4, C + 2), GG(R + 5, C + 2)) + 16) >> 5);
            if (v != h[n * 18 * 19 + R * 19 + C] && bad[1]++ < 5) printf("h[%d][%d][%d] = %d, want %d\n", n, R, C, h[n * 18 * 19 + R * 19 + C], v);
        }
        for (int R = 0; R < 18; R++) for (int C = 0; C < 18; C++) {
            int b1[6];
            for (int k = 0; k < 6; k++) b1[k] = t6(GG(R + k, C), GG(R + k, C + 1), GG(R + k, C + 2), GG(R + k, C + 3), GG(R + k, C + 4), GG(R + k, C + 5));
            int v = c255((t6(b1[0], b1[1], b1[2], b1[3], b1[4], b1[5]) + 512) >> 10);
            if (v != j[n * 18 * 18 + R * 18 + C] && bad[2]++ < 5) printf("j[%d][%d][%d] = %d, want %d\n", n, R, C, j[n * 18 * 18 + R * 18 + C], v);
        }
    }
    printf("mismatches: b %d, h %d, j %d\n", bad[0], bad[1], bad[2]);
    return bad[0] + bad[1] + bad[2] != 0;
}
