// mi355enc_kernels.hip -- hand-written HIP kernels for gfx950 (CDNA4, wave64).
//
// The per-macroblock hot path that the reference delegates to libx264 behind its
// `x264enc` pipeline token (/root/reference/pipeline/generic/x264_superfast_camlink:5):
//   me_kernel        full-search SAD motion estimation            (encoder choice)
//   inter_kernel     MC + residual + 4x4 T/Q + dequant/IDCT + recon (H.264 8.4.2.2, 8.5)
//   intra_kernel     Intra16x16 + chroma prediction, T/Q, recon     (H.264 8.3.3, 8.3.4, 8.5)
//   deblock_kernel   in-loop deblocking filter                      (H.264 8.7)
// Integer arithmetic throughout (u8 samples, 16-bit levels, 32-bit accumulators): results
// must equal oracle/h264_enc_oracle.c byte for byte.  No MFMA: nothing here is a dense
// contraction; the SAD inner loop is v_qsad_pk_u16_u8 on an LDS-staged search window.
#include "mi355enc_dev.h"

#define DEV __device__ __forceinline__

// ------------------------------------------------------------------ constant tables
// One blob so that the latency-critical wavefront kernels can stage it in LDS with a single
// round of loads (a table lookup through global memory costs a full L2 round trip each).
struct dev_tables {
    uint8_t alpha[52], beta[52], tc0[52][3], qpc[52]; // Tables 8-16, 8-17, 8-15
    uint16_t mf[6][3];                                // encoder quantiser multipliers
    uint8_t v[6][3];                                  // 8.5.9 normAdjust4x4
    uint8_t pad[2];
    uint16_t mf8[6][6];                               // 8x8 quantiser multipliers
    uint8_t v8[6][6];                                 // 8.5.9 normAdjust8x8
    uint8_t izz8[64];                                 // 8x8 zig-zag, raster position -> scan index
};
static_assert(sizeof(dev_tables) % 4 == 0, "dev_tables is copied as dwords");
#define TAB_DWORDS ((int)(sizeof(dev_tables) / 4))
__device__ const dev_tables g_tab = {
    {0,  0,  0,  0,  0,  0,  0,  0,  0,   0,   0,   0,   0,   0,   0,   0,   4,   4,
     5,  6,  7,  8,  9,  10, 12, 13, 15,  17,  20,  22,  25,  28,  32,  36,  40,  45,
     50, 56, 63, 71, 80, 90, 101, 113, 127, 144, 162, 182, 203, 226, 255, 255},
    {0, 0, 0, 0, 0, 0, 0, 0, 0,  0,  0,  0,  0,  0,  0,  0,  2,  2,
     2, 3, 3, 3, 3, 4, 4, 4, 6,  6,  7,  7,  8,  8,  9,  9,  10, 10,
     11, 11, 12, 12, 13, 13, 14, 14, 15, 15, 16, 16, 17, 17, 18, 18},
    {{0, 0, 0},   {0, 0, 0},   {0, 0, 0},   {0, 0, 0},   {0, 0, 0},   {0, 0, 0},   {0, 0, 0},   {0, 0, 0},   {0, 0, 0},
     {0, 0, 0},   {0, 0, 0},   {0, 0, 0},   {0, 0, 0},   {0, 0, 0},   {0, 0, 0},   {0, 0, 0},   {0, 0, 0},   {0, 0, 1},
     {0, 0, 1},   {0, 0, 1},   {0, 0, 1},   {0, 1, 1},   {0, 1, 1},   {1, 1, 1},   {1, 1, 1},   {1, 1, 1},   {1, 1, 1},
     {1, 1, 2},   {1, 1, 2},   {1, 1, 2},   {1, 1, 2},   {1, 2, 3},   {1, 2, 3},   {2, 2, 3},   {2, 2, 4},   {2, 3, 4},
     {2, 3, 4},   {3, 3, 5},   {3, 4, 6},   {3, 4, 6},   {4, 5, 7},   {4, 5, 8},   {4, 6, 9},   {5, 7, 10},  {6, 8, 11},
     {6, 8, 13},  {7, 10, 14}, {8, 11, 16}, {9, 12, 18}, {10, 13, 20}, {11, 15, 23}, {13, 17, 25}},
    {0,  1,  2,  3,  4,  5,  6,  7,  8,  9,  10, 11, 12, 13, 14, 15, 16, 17,
     18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 29, 30, 31, 32, 32, 33,
     34, 34, 35, 35, 36, 36, 37, 37, 37, 38, 38, 38, 39, 39, 39, 39},
    {{13107, 5243, 8066}, {11916, 4660, 7490}, {10082, 4194, 6554}, {9362, 3647, 5825}, {8192, 3355, 5243}, {7282, 2893, 4559}},
    {{10, 16, 13}, {11, 18, 14}, {13, 20, 16}, {14, 23, 18}, {16, 25, 20}, {18, 29, 23}},
    {0, 0},
    {{13107, 11428, 20972, 12222, 16777, 15481}, {11916, 10826, 19174, 11058, 14980, 14290}, {10082, 8943, 15978, 9675, 12710, 11985},
     {9362, 8228, 14913, 8931, 11984, 11259},    {8192, 7346, 13159, 7740, 10486, 9777},     {7282, 6428, 11570, 6830, 9118, 8640}},
    {{20, 18, 32, 19, 25, 24}, {22, 19, 35, 21, 28, 26}, {26, 23, 42, 24, 33, 31}, {28, 25, 45, 26, 35, 33}, {32, 28, 51, 30, 40, 38}, {36, 32, 58, 34, 46, 43}},
    {0, 1, 5, 6, 14, 15, 27, 28, 2, 4, 7, 13, 16, 26, 29, 42, 3, 8, 12, 17, 25, 30, 41, 43, 9, 11, 18, 24, 31, 40, 44, 53, 10, 19, 23, 32, 39, 45, 52, 54, 20, 22, 33, 38, 46, 51, 55, 60, 21, 34, 37, 47, 50, 56, 59, 61, 35, 36, 48, 49, 57, 58, 62, 63}};


// ------------------------------------------------------------------ global-memory accessors
// Pointers read out of frame_ctx_t are generic; plain dereferences would become flat_load/
// flat_store, which count against BOTH vmcnt and lgkmcnt -- every LDS wait would then also
// wait for the outstanding HBM load.  These force global_* instructions.
#define GAS __attribute__((address_space(1)))
DEV unsigned ldg8(const void *p) { return *(const GAS uint8_t *)p; }
DEV unsigned ldg32(const void *p) { return *(const GAS unsigned *)p; }
typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));
DEV uint2 ldg64(const void *p) { const v2u v = *(const GAS v2u *)p; return make_uint2(v.x, v.y); }
DEV uint2 ldg64x(const void *p) { return make_uint2(*(const GAS unsigned *)p, *((const GAS unsigned *)p + 1)); } // 4-byte aligned pair
DEV uint4 ldg128(const void *p) { const v4u v = *(const GAS v4u *)p; return make_uint4(v.x, v.y, v.z, v.w); }
DEV int ldg16(const void *p) { return *(const GAS int16_t *)p; }
DEV void stg8(void *p, unsigned v) { *(GAS uint8_t *)p = (uint8_t)v; }
DEV void stg16(void *p, int v) { *(GAS int16_t *)p = (int16_t)v; }
DEV void stg32(void *p, unsigned v) { *(GAS unsigned *)p = v; }
DEV void stg128(void *p, uint4 v) { v4u t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w; *(GAS v4u *)p = t; }
DEV mb_info_t unpack_mbinfo(const uint4 r) {
    mb_info_t m;
    m.mvx = (int16_t)(r.x & 0xFFFF); m.mvy = (int16_t)(r.x >> 16);
    m.mb_type = (uint8_t)(r.y & 255); m.i16_mode = (uint8_t)((r.y >> 8) & 255); m.chroma_mode = (uint8_t)((r.y >> 16) & 255); m.qp = (uint8_t)(r.y >> 24);
    m.nzmask = r.z; m.cost = r.w;
    return m;
}
DEV mb_info_t ld_mbinfo(const mb_info_t *p) { return unpack_mbinfo(ldg128(p)); } // one 16-byte load instead of four partial ones
DEV void st_mbinfo(mb_info_t *p, const mb_info_t &m) {
    uint4 r;
    r.x = ((unsigned)(uint16_t)m.mvx) | ((unsigned)(uint16_t)m.mvy << 16);
    r.y = (unsigned)m.mb_type | ((unsigned)m.i16_mode << 8) | ((unsigned)m.chroma_mode << 16) | ((unsigned)m.qp << 24);
    r.z = m.nzmask; r.w = m.cost;
    stg128(p, r);
}

// ------------------------------------------------------------------ cross-workgroup hand-off inside a persistent launch
// Agent-scope (sc1, L1-bypassing) accesses for data one workgroup produces and another consumes while both run, and a bounded
// wait on a monotonic progress counter (MI355X_MICROARCH.md, "Valid forms": producer stores the data sc1, s_waitcnt vmcnt(0),
// then stores the counter sc1; consumer polls the counter and reads the data with sc1 loads).
#define DB_SPIN_MAX (1 << 20)
DEV unsigned ld_sc1(const unsigned *p) { return __hip_atomic_load((const GAS unsigned *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
DEV void st_sc1(unsigned *p, unsigned v) { __hip_atomic_store((GAS unsigned *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
DEV int db_wait_get(unsigned *progress, unsigned *err, int need) {
    int spins = 0, v;
    while ((v = (int)ld_sc1(progress)) < need) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > DB_SPIN_MAX || ((spins & 1023) == 0 && ld_sc1(err))) { st_sc1(err, 1u); return 0x7FFFFFFF; } // bounded; once tripped, nobody waits again
    }
    return v;
}

// workgroup barrier that drains LDS traffic only: global loads (prefetch) and stores stay in flight
#define BAND_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)

DEV int iabs(int v) { return v < 0 ? -v : v; }
DEV int clip3(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }
DEV int clip255(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }
DEV int blkx(int b) { return ((b & 1) << 2) | ((b & 4) << 1); }        // luma4x4BlkIdx -> x offset (6.4.3)
DEV int blky(int b) { return ((b & 2) << 1) | ((b & 8)); }             // luma4x4BlkIdx -> y offset

// XCD-aware block remap: consecutive logical tiles land on the same XCD (blocks are dealt
// round-robin over the 8 XCDs), so neighbouring strips share one L2.  Bijective for any n.
DEV int xcd_remap(int wg, int n) {
    int q = n >> 3, r = n & 7, k = wg & 7;
    return k * q + (k < r ? k : r) + (wg >> 3);
}

// =================================================================== motion search
// One workgroup = ME_MBS horizontally adjacent macroblocks of one macroblock row, one wave per
// macroblock.  The 48 x (16*ME_MBS+32) luma search window (+-16 around the strip) is staged once in
// LDS.  Lane l < 63 of a wave owns the candidates
//   dy in [-16 + 5*(l/9), +5)   x   dx in [-16 + 4*(l%9), +4)
// (7 x 9 tiles cover 35 x 36 >= 33 x 33; 85 % of the computed SADs are real candidates) and
// accumulates them with v_qsad_pk_u16_u8 -- four 4-pixel SADs per instruction -- re-using each
// window row for the 5 dy it serves.  The (cost, dy, dx) minimum is reduced over the wave with
// cross-lane shuffles.
#ifndef ME_MBS
#define ME_MBS 4
#endif
#define ME_WQ (ME_MBS + 2)      /* uint4 per window row: 16*ME_MBS + 32 bytes */
#define ME_ROWS 50   /* 48 real rows + 2 that only masked candidates (dy = 17, 18) ever touch */
#define ME_STRIDE 53 /* words; 5*53 mod 32 = 9 -> consecutive dy-groups start 9 banks apart */
#define ME_K 5       /* dy per lane */

DEV unsigned long long qsad(unsigned lo, unsigned hi, unsigned cur, unsigned long long acc) {
    unsigned long long src = ((unsigned long long)hi << 32) | lo;
    return __builtin_amdgcn_qsad_pk_u16_u8(src, cur, acc);
}
DEV int mv_bits(int v) { // bits of se(4v): 1 for 0, else 7 + 2*floor(log2|v|)
    int a = iabs(v);
    return a == 0 ? 1 : 7 + 2 * (31 - __clz(a));
}

__global__ __launch_bounds__(64 * ME_MBS) void me_kernel(const frame_ctx_t cv, int row0) { // context by value: lives in the kernarg segment, no per-picture upload
    const frame_ctx_t *__restrict__ ctx = &cv;
    __shared__ unsigned win[ME_ROWS * ME_STRIDE];
    const int stride = ctx->stride, mbw = ctx->mbw, mbh = ctx->mbh;
    const int W = mbw * 16, H = mbh * 16;
    const int strips = (mbw + ME_MBS - 1) / ME_MBS;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int my = row0 + tile / strips, sx = tile % strips; // the launch covers macroblock rows row0 .. row0 + gridDim.x / strips - 1
    const int t = threadIdx.x;
    const uint8_t *__restrict__ ref = ctx->ref_y;

    // ---- stage the window: 48 rows x 10 uint4 (coalesced 16 B per lane)
    for (int i = t; i < 48 * ME_WQ; i += 64 * ME_MBS) {
        int row = i / ME_WQ, q = i - row * ME_WQ;
        int gy = my * 16 - 16 + row, gx = sx * (ME_MBS * 16) - 16 + 16 * q;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = ldg128(ref + (size_t)gy * stride + gx);
        unsigned *d = &win[row * ME_STRIDE + 4 * q];
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    if (t < ME_ROWS) win[t * ME_STRIDE + 4 * ME_WQ] = 0;
    if (t < 2 * (4 * ME_WQ + 1)) win[(48 + t / (4 * ME_WQ + 1)) * ME_STRIDE + t % (4 * ME_WQ + 1)] = 0;

    const int lane = t & 63, m = t >> 6;
    const int mx = sx * ME_MBS + m;
    const bool active = lane < 63 && mx < mbw;
    const int g = lane < 63 ? lane / 9 : 0, dxg = lane < 63 ? lane % 9 : 0;
    const int mxc = mx < mbw ? mx : mbw - 1;

    // ---- current macroblock: 16 rows x 4 words, identical in every lane of the wave
    unsigned c[16][4];
    {
        const uint8_t *__restrict__ src = ctx->src_y;
        const int ss = ctx->src_stride, vh = ctx->vis_h;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            int sy = my * 16 + r;
            sy = sy < vh ? sy : vh - 1;
            uint4 v = ldg128(src + (size_t)sy * ss + mxc * 16);
            c[r][0] = v.x; c[r][1] = v.y; c[r][2] = v.z; c[r][3] = v.w;
        }
    }
    __syncthreads();

    unsigned long long acc[ME_K];
#pragma unroll
    for (int d = 0; d < ME_K; d++) acc[d] = 0;
    const unsigned *wp = &win[(ME_K * g) * ME_STRIDE + 4 * m + dxg];
    // one window row ahead in registers; sched_barrier keeps the compiler from hoisting all 100 LDS
    // reads to the top (which costs > 200 VGPRs and halves the occupancy)
    unsigned w0 = wp[0], w1 = wp[1], w2 = wp[2], w3 = wp[3], w4 = wp[4];
#pragma unroll
    for (int j = 0; j < 16 + ME_K - 1; j++) {
        unsigned n0 = 0, n1 = 0, n2 = 0, n3 = 0, n4 = 0;
        if (j + 1 < 16 + ME_K - 1) {
            n0 = wp[(j + 1) * ME_STRIDE + 0]; n1 = wp[(j + 1) * ME_STRIDE + 1]; n2 = wp[(j + 1) * ME_STRIDE + 2];
            n3 = wp[(j + 1) * ME_STRIDE + 3]; n4 = wp[(j + 1) * ME_STRIDE + 4];
        }
#pragma unroll
        for (int d = 0; d < ME_K; d++) {
            const int r = j - d;
            if (r >= 0 && r < 16) {
                acc[d] = qsad(w0, w1, c[r][0], acc[d]);
                acc[d] = qsad(w1, w2, c[r][1], acc[d]);
                acc[d] = qsad(w2, w3, c[r][2], acc[d]);
                acc[d] = qsad(w3, w4, c[r][3], acc[d]);
            }
        }
        // pin this row's SADs here (pure intrinsics would otherwise sink below all the LDS reads)
        asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]));
        __builtin_amdgcn_sched_barrier(0);
        w0 = n0; w1 = n1; w2 = n2; w3 = n3; w4 = n4;
    }

    // ---- cost = SAD + lambda*(bits(dx)+bits(dy)); key = cost<<12 | (dy+16)<<6 | (dx+16)
    const int R = ctx->me_range, lambda = ctx->lambda;
    const int x0 = mxc * 16, y0 = my * 16;
    const int dx_lo = -R < -x0 ? -x0 : -R, dx_hi = R > W - 16 - x0 ? W - 16 - x0 : R;
    const int dy_lo = -R < -y0 ? -y0 : -R, dy_hi = R > H - 16 - y0 ? H - 16 - y0 : R;
    const unsigned INVALID = 0x40000000u;
    unsigned bo[4];
#pragma unroll
    for (int o = 0; o < 4; o++) {
        int dx = -16 + 4 * dxg + o;
        bo[o] = (dx >= dx_lo && dx <= dx_hi && active) ? (((unsigned)(lambda * mv_bits(dx)) << 12) | (unsigned)(dx + 16)) : INVALID;
    }
    unsigned best = 0xFFFFFFFFu;
#pragma unroll
    for (int d = 0; d < ME_K; d++) {
        int dy = -16 + ME_K * g + d;
        unsigned bd = (dy >= dy_lo && dy <= dy_hi) ? (((unsigned)(lambda * mv_bits(dy)) << 12) | ((unsigned)(dy + 16) << 6)) : INVALID;
        unsigned lo = (unsigned)acc[d], hi = (unsigned)(acc[d] >> 32);
        unsigned k0 = ((lo << 16) >> 4) + bd + bo[0];
        unsigned k1 = ((lo & 0xFFFF0000u) >> 4) + bd + bo[1];
        unsigned k2 = ((hi << 16) >> 4) + bd + bo[2];
        unsigned k3 = ((hi & 0xFFFF0000u) >> 4) + bd + bo[3];
        unsigned ka = k0 < k1 ? k0 : k1, kb = k2 < k3 ? k2 : k3;
        ka = ka < kb ? ka : kb;
        best = best < ka ? best : ka;
    }
    // ---- wave-wide minimum
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1) {
        unsigned o = (unsigned)__shfl_xor((int)best, sft, 64);
        best = best < o ? best : o;
    }
    if (lane == 0 && mx < mbw) {
        mb_info_t *mb = &ctx->mbi[my * mbw + mx];
        const int bx_ = (int)(best & 63) - 16, by_ = (int)((best >> 6) & 63) - 16;
        stg32(&mb->mvx, ((unsigned)(uint16_t)(4 * bx_)) | ((unsigned)(uint16_t)(4 * by_) << 16)); // quarter-sample units
        stg32(&mb->cost, best >> 12);
    }
}

// =================================================================== 4x4 transform helpers
// All operate on int x[16] in raster order (index y*4+x); loops are fully unrolled so the
// arrays stay in registers.
DEV void fdct4(int *x) { // Y = Cf X Cf^T
#pragma unroll
    for (int i = 0; i < 4; i++) {
        int a = x[i * 4], b = x[i * 4 + 1], c = x[i * 4 + 2], d = x[i * 4 + 3];
        int s03 = a + d, d03 = a - d, s12 = b + c, d12 = b - c;
        x[i * 4] = s03 + s12; x[i * 4 + 1] = 2 * d03 + d12; x[i * 4 + 2] = s03 - s12; x[i * 4 + 3] = d03 - 2 * d12;
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
        int a = x[j], b = x[4 + j], c = x[8 + j], d = x[12 + j];
        int s03 = a + d, d03 = a - d, s12 = b + c, d12 = b - c;
        x[j] = s03 + s12; x[4 + j] = 2 * d03 + d12; x[8 + j] = s03 - s12; x[12 + j] = d03 - 2 * d12;
    }
}
DEV void idct4(int *d) { // 8.5.12.2: rows then columns, (x+32)>>6; result = residual
#pragma unroll
    for (int i = 0; i < 4; i++) {
        int e0 = d[i * 4] + d[i * 4 + 2], e1 = d[i * 4] - d[i * 4 + 2];
        int e2 = (d[i * 4 + 1] >> 1) - d[i * 4 + 3], e3 = d[i * 4 + 1] + (d[i * 4 + 3] >> 1);
        d[i * 4] = e0 + e3; d[i * 4 + 1] = e1 + e2; d[i * 4 + 2] = e1 - e2; d[i * 4 + 3] = e0 - e3;
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
        int g0 = d[j] + d[8 + j], g1 = d[j] - d[8 + j];
        int g2 = (d[4 + j] >> 1) - d[12 + j], g3 = d[4 + j] + (d[12 + j] >> 1);
        d[j] = (g0 + g3 + 32) >> 6; d[4 + j] = (g1 + g2 + 32) >> 6; d[8 + j] = (g1 - g2 + 32) >> 6; d[12 + j] = (g0 - g3 + 32) >> 6;
    }
}
// 8.5.6 zig-zag: scan position -> raster index, packed one nibble per entry
DEV constexpr int zz(int k) { return (int)((0xFEB7ADC963258410ull >> (4 * k)) & 15); }
DEV constexpr int pos_class(int p) { // 0: (even,even)  1: (odd,odd)  2: mixed
    return ((p & 1) == 0 && (p & 4) == 0) ? 0 : (((p & 1) && (p & 4)) ? 1 : 2);
}
DEV int quant1(int coef, int mf, int f, int qbits) { // dead-zone quantiser, |level| <= 2047
    int a = iabs(coef);
    int l = (a * mf + f) >> qbits;
    l = l > 2047 ? 2047 : l;
    return coef < 0 ? -l : l;
}
struct qparams { int mf[3], v[3], qbits, f, shift; };
DEV qparams make_q(const dev_tables *T, int qp, bool intra) {
    qparams q;
    int m = qp % 6;
    q.mf[0] = T->mf[m][0]; q.mf[1] = T->mf[m][1]; q.mf[2] = T->mf[m][2];
    q.v[0] = T->v[m][0]; q.v[1] = T->v[m][1]; q.v[2] = T->v[m][2];
    q.qbits = 15 + qp / 6;
    q.f = (1 << q.qbits) / (intra ? 3 : 6);
    q.shift = qp / 6;
    return q;
}
// coef[] (raster, after fdct4) -> lev[] (zig-zag order) and coef[] := dequantised (raster).
// Scan positions below `first` are forced to zero.  Returns true if any level != 0.
template <int FIRST>
DEV bool quant_dequant(int *coef, int *lev, const qparams &q) {
    bool nz = false;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const int p = zz(k), cl = pos_class(p);
        int l = k < FIRST ? 0 : quant1(coef[p], q.mf[cl], q.f, q.qbits);
        lev[k] = l;
        nz |= l != 0;
        coef[p] = (l * q.v[cl]) << q.shift;
    }
    return nz;
}
DEV void store_levels(int16_t *dst, const int *lev) { // 16 int16 = two 16-byte stores
    uint4 a, b;
    a.x = (lev[0] & 0xFFFF) | (lev[1] << 16); a.y = (lev[2] & 0xFFFF) | (lev[3] << 16);
    a.z = (lev[4] & 0xFFFF) | (lev[5] << 16); a.w = (lev[6] & 0xFFFF) | (lev[7] << 16);
    b.x = (lev[8] & 0xFFFF) | (lev[9] << 16); b.y = (lev[10] & 0xFFFF) | (lev[11] << 16);
    b.z = (lev[12] & 0xFFFF) | (lev[13] << 16); b.w = (lev[14] & 0xFFFF) | (lev[15] << 16);
    stg128(dst, a); stg128(dst + 8, b);
}
DEV unsigned pack4(int a, int b, int c, int d) { return (unsigned)a | ((unsigned)b << 8) | ((unsigned)c << 16) | ((unsigned)d << 24); }
DEV int byte_of(unsigned w, int i) { return (int)((w >> (8 * i)) & 255); }

// Chroma of one macroblock, run by 8 consecutive lanes (cl = 0..7: plane c = cl>>2, block b = cl&3).
// value of lane (l ^ K) of this lane's quad, K = 1..3 (DPP quad_perm)
template <int K> DEV int quad_xor(int v) { return __builtin_amdgcn_update_dpp(0, v, K == 1 ? 0xB1 : K == 2 ? 0x4E : 0x1B, 0xF, 0xF, false); }
// pred[16]: prediction of this lane's 4x4 block.  Handles the 2x2 DC Hadamard across the four
// lanes of a plane with shuffles (8.5.11), writes levels + reconstruction, returns the AC flag
// in bit 0 and the plane's DC flag in bit 1.
DEV int chroma_block(const frame_ctx_t *ctx, const dev_tables *T, int mbn, int cx0, int cy0, int cl, const int *pred, int qp, bool intra, uint8_t *lrec = nullptr, const uint2 *presrc = nullptr) {
    const int c = cl >> 2, b = cl & 3, bx = (b & 1) * 4, by = (b >> 1) * 4;
    const int qpc = T->qpc[qp];
    const qparams q = make_q(T, qpc, intra);
    int x[16], lev[16];
    {
        const uint8_t *__restrict__ s = ctx->src_uv;
        const int ss = ctx->src_stride, vh2 = ctx->vis_h >> 1;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            int sy = cy0 + by + r;
            sy = sy < vh2 ? sy : vh2 - 1;
            uint2 w = presrc ? presrc[r] : ldg64(s + (size_t)sy * ss + 2 * (cx0 + bx));
            unsigned lo = c ? (w.x >> 8) : w.x, hi = c ? (w.y >> 8) : w.y;
            x[r * 4 + 0] = (int)(lo & 255) - pred[r * 4 + 0];
            x[r * 4 + 1] = (int)((lo >> 16) & 255) - pred[r * 4 + 1];
            x[r * 4 + 2] = (int)(hi & 255) - pred[r * 4 + 2];
            x[r * 4 + 3] = (int)((hi >> 16) & 255) - pred[r * 4 + 3];
        }
    }
    fdct4(x);
    const int dc = x[0];
    bool nz_ac = quant_dequant<1>(x, lev, q);
    // forward 2x2 Hadamard over the plane's four lanes; lane b keeps element b
    int d1 = quad_xor<1>(dc), d2 = quad_xor<2>(dc), d3 = quad_xor<3>(dc); // DPP quad_perm: the four blocks of a plane sit in one quad
    // with e0..e3 the values of blocks 0..3: this lane holds e_b = dc, e_{b^1} = d1, e_{b^2} = d2, e_{b^3} = d3
    int fb;
    {
        int e[4]; // e_k = DC of block k (this lane holds e_b; partners arrive by shuffle)
        e[b] = dc; e[b ^ 1] = d1; e[b ^ 2] = d2; e[b ^ 3] = d3;
        int f0 = e[0] + e[1] + e[2] + e[3], f1 = e[0] - e[1] + e[2] - e[3];
        int f2 = e[0] + e[1] - e[2] - e[3], f3 = e[0] - e[1] - e[2] + e[3];
        fb = b == 0 ? f0 : b == 1 ? f1 : b == 2 ? f2 : f3;
    }
    const int ldc = quant1(fb, q.mf[0], 2 * q.f, q.qbits + 1);
    // inverse: g = H l H over the four DC levels, dcC = ((g*LevelScale(0,0)) << (qP/6)) >> 5
    int l1 = quad_xor<1>(ldc), l2 = quad_xor<2>(ldc), l3 = quad_xor<3>(ldc);
    int gl[4];
    gl[b] = ldc; gl[b ^ 1] = l1; gl[b ^ 2] = l2; gl[b ^ 3] = l3;
    int g0 = gl[0] + gl[1] + gl[2] + gl[3], g1 = gl[0] - gl[1] + gl[2] - gl[3];
    int g2 = gl[0] + gl[1] - gl[2] - gl[3], g3 = gl[0] - gl[1] - gl[2] + gl[3];
    int gb = b == 0 ? g0 : b == 1 ? g1 : b == 2 ? g2 : g3;
    x[0] = ((gb * 16 * q.v[0]) << q.shift) >> 5;
    const bool nz_dc = (gl[0] | gl[1] | gl[2] | gl[3]) != 0;
    idct4(x);
    // levels
    int16_t *lv = ctx->levels + (size_t)mbn * MB_LEVELS;
    store_levels(lv + L_CAC + (4 * c + b) * 16, lev);
    stg16(&lv[L_CDC + 4 * c + b], ldc);
    // reconstruction: this lane owns every other byte of 8-byte row segments
    uint8_t *__restrict__ rec = ctx->rec_uv;
    const int st = ctx->stride;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        uint8_t *p = rec + (size_t)(cy0 + by + r) * st + 2 * (cx0 + bx) + c;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const unsigned v = (unsigned)clip255(pred[r * 4 + i] + x[r * 4 + i]);
            stg8(p + 2 * i, v);
            if (lrec) lrec[(by + r) * 16 + 2 * (bx + i) + c] = (uint8_t)v; // 8 x 16 interleaved tile in LDS for the persistent intra kernel
        }
    }
    return (nz_ac ? 1 : 0) | (nz_dc ? 2 : 0);
}

// =================================================================== sub-sample refinement
// One wave per macroblock.  Around the integer winner (ix, iy) the wave builds, in LDS, the
// integer samples G and the three half-sample planes of 8.4.2.2.1 (b: horizontal 6-tap,
// h: vertical 6-tap, j: centre, 6-tap over the unrounded horizontal intermediates) on an
// 18 x 18 (+1) grid; every quarter-sample candidate is then the rounded average of two plane
// entries (Table 8-12).  Two rounds (step 2, then step 1) of the 8 neighbours, visited in
// (dy, dx) raster order, strictly-lower cost wins -- the oracle's orc_subpel_frame.
// sum over the 16 lanes of a DPP row (every lane receives it)
DEV int wave16_sum(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x128, 0xF, 0xF, false); // row_ror:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x124, 0xF, 0xF, false); // row_ror:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false);  // quad_perm:[2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false);  // quad_perm:[1,0,3,2]
    return v;
}
#define SP_GS 24 /* G row stride (23 used) */
#define SP_PS 20 /* plane row stride (18/19 used) */
struct sp_lds {
    uint8_t G[23 * SP_GS];     // rows iy-3 .. iy+19, cols ix-3 .. ix+19
    int16_t B1[23 * 18];       // unrounded horizontal half samples: rows iy-3 .. iy+19, cols ix-1 .. ix+16
    uint8_t b[19 * SP_PS];     // rows iy-1 .. iy+17, cols ix-1 .. ix+16
    uint8_t h[18 * SP_PS];     // rows iy-1 .. iy+16, cols ix-1 .. ix+17
    uint8_t j[18 * SP_PS];     // rows iy-1 .. iy+16, cols ix-1 .. ix+16
    uint8_t pad[16];           // lds4() may read one word past the last sample of a plane
};
DEV int tap6(int a, int b, int c, int d, int e, int f) { return a - 5 * b + 20 * c + 20 * d - 5 * e + f; }
DEV int mvq_bits(int q) { // bits of se(q)
    unsigned k = q > 0 ? (unsigned)(2 * q - 1) : (unsigned)(-2 * q);
    return 2 * (31 - __clz((int)(k + 1))) + 1;
}
// four horizontally adjacent bytes of an LDS plane starting at byte offset `o` (any alignment): two aligned words + one v_alignbyte
DEV unsigned lds4(const uint8_t *plane, int o) {
    const unsigned *w = (const unsigned *)(plane + (o & ~3));
    return __builtin_amdgcn_alignbyte(w[1], w[0], (unsigned)(o & 3));
}
DEV unsigned avg4(unsigned a, unsigned b) { return (a | b) - (((a ^ b) >> 1) & 0x7F7F7F7Fu); } // per byte (a + b + 1) >> 1
// the four luma samples at plane positions (X..X+3, Y) (plane coordinates: 0 = ix-1 / iy-1) and fraction (fx, fy), one per byte
// (8.4.2.2.1, Table 8-12).  fx, fy are wave-uniform, so the case analysis costs no divergence.
DEV unsigned sp_sample4(const sp_lds *L, int X, int Y, int fx, int fy) {
#define SG(x, y) lds4(L->G, ((y) + 2) * SP_GS + (x) + 2)
#define SB(x, y) lds4(L->b, (y) * SP_PS + (x))
#define SH(x, y) lds4(L->h, (y) * SP_PS + (x))
#define SJ(x, y) lds4(L->j, (y) * SP_PS + (x))
    if (fy == 0) {
        if (fx == 0) return SG(X, Y);
        return fx == 2 ? SB(X, Y) : fx == 1 ? avg4(SG(X, Y), SB(X, Y)) : avg4(SG(X + 1, Y), SB(X, Y));
    }
    if (fx == 0) return fy == 2 ? SH(X, Y) : fy == 1 ? avg4(SG(X, Y), SH(X, Y)) : avg4(SG(X, Y + 1), SH(X, Y));
    if ((fx & 1) && (fy & 1)) return avg4(fy == 1 ? SB(X, Y) : SB(X, Y + 1), fx == 1 ? SH(X, Y) : SH(X + 1, Y));
    if (fx == 2 && fy == 2) return SJ(X, Y);
    if (fx == 2) return avg4(fy == 1 ? SB(X, Y) : SB(X, Y + 1), SJ(X, Y));
    return avg4(fx == 1 ? SH(X, Y) : SH(X + 1, Y), SJ(X, Y));
#undef SG
#undef SB
#undef SH
#undef SJ
}
__global__ __launch_bounds__(256) void subpel_kernel(const frame_ctx_t cv, int mb0, int mb1) {
    const frame_ctx_t *__restrict__ ctx = &cv;
    __shared__ __attribute__((aligned(16))) sp_lds LD[4];
    const int mbw = ctx->mbw, mbh = ctx->mbh, stride = ctx->stride, W = mbw * 16, H = mbh * 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int mbn = mb0 + blockIdx.x * 4 + wave; // the launch covers macroblocks mb0 .. mb1-1
    const bool ok = mbn < mb1;
    if (!ok) mbn = mb1 - 1;
    const int my = mbn / mbw, mx = mbn - my * mbw, x0 = mx * 16, y0 = my * 16;
    sp_lds *L = &LD[wave];
    const mb_info_t info = ld_mbinfo(&ctx->mbi[mbn]);
    const int ix = x0 + (info.mvx >> 2), iy = y0 + (info.mvy >> 2); // integer winner (vector is a multiple of 4 here)
    const uint8_t *__restrict__ ref = ctx->ref_y;
    // ---- G with the picture extended by coordinate clamping (8.4.2.2.1).  Window fully inside the picture (the usual
    // case, wave-uniform): 23 rows x 7 aligned words, shifted into place with v_alignbyte; otherwise byte by byte.
    if (ix - 3 >= 0 && iy - 3 >= 0 && ((ix - 3) & ~3) + 28 <= W && iy + 19 < H) {
        const int a = (ix - 3) & 3;
        const uint8_t *base = ref + (size_t)(iy - 3) * stride + ((ix - 3) & ~3);
        for (int i = lane; i < 23 * 6; i += 64) {
            const int r = i / 6, d = i - r * 6;
            const unsigned w0 = ldg32(base + (size_t)r * stride + 4 * d), w1 = ldg32(base + (size_t)r * stride + 4 * d + 4);
            *(unsigned *)&L->G[r * SP_GS + 4 * d] = __builtin_amdgcn_alignbyte(w1, w0, (unsigned)a);
        }
    } else
        for (int i = lane; i < 23 * 23; i += 64) {
            int r = i / 23, c = i - r * 23;
            int yy = clip3(0, H - 1, iy - 3 + r), xx = clip3(0, W - 1, ix - 3 + c);
            L->G[r * SP_GS + c] = (uint8_t)ldg8(ref + (size_t)yy * stride + xx);
        }
    // current macroblock: lane owns row lane>>2, columns 4*(lane&3) .. +3
    const int pr = lane >> 2, pc = (lane & 3) * 4;
    unsigned curw;
    {
        int sy = y0 + pr;
        sy = sy < ctx->vis_h ? sy : ctx->vis_h - 1;
        curw = ldg32(ctx->src_y + (size_t)sy * ctx->src_stride + x0 + pc);
    }
    WAVE_SYNC();
    // ---- horizontal half samples (unrounded B1, rounded b)
    for (int i = lane; i < 23 * 18; i += 64) {
        int r = i / 18, c = i - r * 18; // position x = ix-1+c -> G column c+2; taps at G columns c .. c+5
        const uint8_t *g = &L->G[r * SP_GS + c];
        int v = tap6(g[0], g[1], g[2], g[3], g[4], g[5]);
        L->B1[r * 18 + c] = (int16_t)v;
        if (r >= 2 && r < 21) L->b[(r - 2) * SP_PS + c] = (uint8_t)clip255((v + 16) >> 5);
    }
    // ---- vertical half samples h: rows iy-1 .. iy+16 (G rows 2..19), cols ix-1 .. ix+17 (G cols 2..20)
    for (int i = lane; i < 18 * 19; i += 64) {
        int r = i / 19, c = i - r * 19;
        const uint8_t *g = &L->G[r * SP_GS + c + 2]; // taps at G rows r .. r+5
        int v = tap6(g[0], g[SP_GS], g[2 * SP_GS], g[3 * SP_GS], g[4 * SP_GS], g[5 * SP_GS]);
        L->h[r * SP_PS + c] = (uint8_t)clip255((v + 16) >> 5);
    }
    WAVE_SYNC();
    // ---- centre samples j: vertical 6-tap over B1
    for (int i = lane; i < 18 * 18; i += 64) {
        int r = i / 18, c = i - r * 18;
        const int16_t *q = &L->B1[r * 18 + c];
        int v = tap6(q[0], q[18], q[36], q[54], q[72], q[90]);
        L->j[r * SP_PS + c] = (uint8_t)clip255((v + 512) >> 10);
    }
    WAVE_SYNC();
    // ---- two refinement rounds (half, then quarter).  The 8 candidates of a round are scored together: per lane one
    // v_sad_u8 over its 4 pixels each, two 16-bit partial sums per register (64 lanes x 1020 < 65536), one wave reduction
    // for all of them; then the candidates are compared in scan order with a strict `<`, as the oracle does.
    const int lambda = ctx->lambda;
    int bqx = info.mvx, bqy = info.mvy;
    unsigned best = info.cost;
#pragma unroll 1
    for (int step = 2; step >= 1; step--) {
        const int cqx = bqx, cqy = bqy;
        unsigned acc[4] = {0, 0, 0, 0};
#pragma unroll
        for (int c8 = 0; c8 < 8; c8++) {
            const int k = c8 < 4 ? c8 : c8 + 1;
            const int qx = cqx + (k % 3 - 1) * step, qy = cqy + (k / 3 - 1) * step;
            const int ox = qx - info.mvx, oy = qy - info.mvy;               // -3 .. 3 relative to the integer winner
            const int X = 1 + (ox >> 2) + pc, Y = 1 + (oy >> 2) + pr;       // plane coordinates of this lane's first pixel
            const unsigned sad = __builtin_amdgcn_sad_u8(curw, sp_sample4(L, X, Y, ox & 3, oy & 3), 0u);
            acc[c8 >> 1] |= sad << (16 * (c8 & 1));
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            int v = wave16_sum((int)acc[q]);
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            acc[q] = (unsigned)v;
        }
#pragma unroll
        for (int c8 = 0; c8 < 8; c8++) {
            const int k = c8 < 4 ? c8 : c8 + 1;
            const int qx = cqx + (k % 3 - 1) * step, qy = cqy + (k / 3 - 1) * step;
            const unsigned sad = (acc[c8 >> 1] >> (16 * (c8 & 1))) & 0xFFFFu;
            const unsigned cost = sad + (unsigned)(lambda * (mvq_bits(qx) + mvq_bits(qy)));
            if (cost < best) { best = cost; bqx = qx; bqy = qy; }
        }
    }
    if (lane == 0 && ok) {
        mb_info_t *mb = &ctx->mbi[mbn];
        stg32(&mb->mvx, ((unsigned)(uint16_t)bqx) | ((unsigned)(uint16_t)bqy << 16));
        stg32(&mb->cost, best);
    }
}

// 8.4.2.2.1 for one sample out of a clamped 9 x 9 neighbourhood held in registers:
// n[r][c] is the integer sample at (X - 2 + c, Y - 2 + r) of the block's first pixel; (i, jj) selects the pixel.
DEV int qpel_from9(const int (*n)[9], int i, int jj, int fx, int fy) {
#define N(dx, dy) n[jj + 2 + (dy)][i + 2 + (dx)]
#define HB1(dx, dy) tap6(N((dx) - 2, dy), N((dx) - 1, dy), N(dx, dy), N((dx) + 1, dy), N((dx) + 2, dy), N((dx) + 3, dy))
#define VH1(dx, dy) tap6(N(dx, (dy) - 2), N(dx, (dy) - 1), N(dx, dy), N(dx, (dy) + 1), N(dx, (dy) + 2), N(dx, (dy) + 3))
    const int G = N(0, 0);
    if (!fx && !fy) return G;
    const int b = clip255((HB1(0, 0) + 16) >> 5), h = clip255((VH1(0, 0) + 16) >> 5);
    if (!fy) return fx == 2 ? b : fx == 1 ? (G + b + 1) >> 1 : (N(1, 0) + b + 1) >> 1;
    if (!fx) return fy == 2 ? h : fy == 1 ? (G + h + 1) >> 1 : (N(0, 1) + h + 1) >> 1;
    const int m = clip255((VH1(1, 0) + 16) >> 5), s = clip255((HB1(0, 1) + 16) >> 5);
    if ((fx & 1) && (fy & 1)) return ((fy == 1 ? b : s) + (fx == 1 ? h : m) + 1) >> 1;
    const int j = clip255((tap6(HB1(0, -2), HB1(0, -1), HB1(0, 0), HB1(0, 1), HB1(0, 2), HB1(0, 3)) + 512) >> 10);
    if (fx == 2 && fy == 2) return j;
    if (fx == 2) return ((fy == 1 ? b : s) + j + 1) >> 1;
    return ((fx == 1 ? h : m) + j + 1) >> 1;
#undef N
#undef HB1
#undef VH1
}

// 8-point transforms of the High-profile 8x8 residual path: forward (encoder side) and 8.5.13 inverse
DEV void fdct8_1d(int *v) {
    const int s07 = v[0] + v[7], s16 = v[1] + v[6], s25 = v[2] + v[5], s34 = v[3] + v[4];
    const int a0 = s07 + s34, a1 = s16 + s25, a2 = s07 - s34, a3 = s16 - s25;
    const int d07 = v[0] - v[7], d16 = v[1] - v[6], d25 = v[2] - v[5], d34 = v[3] - v[4];
    const int a4 = d16 + d25 + (d07 + (d07 >> 1)), a5 = d07 - d34 - (d25 + (d25 >> 1));
    const int a6 = d07 + d34 - (d16 + (d16 >> 1)), a7 = d16 - d25 + (d34 + (d34 >> 1));
    v[0] = a0 + a1; v[1] = a4 + (a7 >> 2); v[2] = a2 + (a3 >> 1); v[3] = a5 + (a6 >> 2);
    v[4] = a0 - a1; v[5] = a6 - (a5 >> 2); v[6] = (a2 >> 1) - a3; v[7] = (a4 >> 2) - a7;
}
DEV void idct8_1d(int *v) {
    const int a0 = v[0] + v[4], a2 = v[0] - v[4], a4 = (v[2] >> 1) - v[6], a6 = (v[6] >> 1) + v[2];
    const int b0 = a0 + a6, b2 = a2 + a4, b4 = a2 - a4, b6 = a0 - a6;
    const int a1 = -v[3] + v[5] - v[7] - (v[7] >> 1), a3 = v[1] + v[7] - v[3] - (v[3] >> 1);
    const int a5 = -v[1] + v[7] + v[5] + (v[5] >> 1), a7 = v[3] + v[5] + v[1] + (v[1] >> 1);
    const int b1 = (a7 >> 2) + a1, b3 = a3 + (a5 >> 2), b5 = (a3 >> 2) - a5, b7 = a7 - (a1 >> 2);
    v[0] = b0 + b7; v[1] = b2 + b5; v[2] = b4 + b3; v[3] = b6 + b1;
    v[4] = b6 - b1; v[5] = b4 - b3; v[6] = b2 - b5; v[7] = b0 - b7;
}
DEV int pos_class8(int y, int x) { // 8.5.9
    if (!(y & 3) && !(x & 3)) return 0;
    if ((y & 1) && (x & 1)) return 1;
    if ((y & 3) == 2 && (x & 3) == 2) return 2;
    if ((!(y & 3) && (x & 1)) || ((y & 1) && !(x & 3))) return 3;
    if ((!(y & 3) && (x & 3) == 2) || ((y & 3) == 2 && !(x & 3))) return 4;
    return 5;
}
// 8.4.2.2.1 for one sample out of a clamped neighbourhood with NC columns held in registers:
// n[r][c] is the integer sample at (X - 2 + c, Y - 2 + r) of the region's first pixel; (i, jj) selects the pixel.
template <int NC>
DEV int qpel_nb(const int (*n)[NC], int i, int jj, int fx, int fy) {
#define N(dx, dy) n[jj + 2 + (dy)][i + 2 + (dx)]
#define HB1(dx, dy) tap6(N((dx) - 2, dy), N((dx) - 1, dy), N(dx, dy), N((dx) + 1, dy), N((dx) + 2, dy), N((dx) + 3, dy))
#define VH1(dx, dy) tap6(N(dx, (dy) - 2), N(dx, (dy) - 1), N(dx, dy), N(dx, (dy) + 1), N(dx, (dy) + 2), N(dx, (dy) + 3))
    const int G = N(0, 0);
    if (!fx && !fy) return G;
    const int b = clip255((HB1(0, 0) + 16) >> 5), h = clip255((VH1(0, 0) + 16) >> 5);
    if (!fy) return fx == 2 ? b : fx == 1 ? (G + b + 1) >> 1 : (N(1, 0) + b + 1) >> 1;
    if (!fx) return fy == 2 ? h : fy == 1 ? (G + h + 1) >> 1 : (N(0, 1) + h + 1) >> 1;
    const int m = clip255((VH1(1, 0) + 16) >> 5), s = clip255((HB1(0, 1) + 16) >> 5);
    if ((fx & 1) && (fy & 1)) return ((fy == 1 ? b : s) + (fx == 1 ? h : m) + 1) >> 1;
    const int j = clip255((tap6(HB1(0, -2), HB1(0, -1), HB1(0, 0), HB1(0, 1), HB1(0, 2), HB1(0, 3)) + 512) >> 10);
    if (fx == 2 && fy == 2) return j;
    if (fx == 2) return ((fy == 1 ? b : s) + j + 1) >> 1;
    return ((fx == 1 ? h : m) + j + 1) >> 1;
#undef N
#undef HB1
#undef VH1
}

// =================================================================== inter (P) macroblocks
// One wave = two macroblocks.  Lanes 0-31: one 4x4 luma block each (MB = lane>>4);
// lanes 32-47: one 4x4 chroma block each (MB = (lane-32)>>3); lanes 48-63 idle.
__global__ __launch_bounds__(256) void inter_kernel(const frame_ctx_t cv, int mb0, int mb1) {
    const frame_ctx_t *__restrict__ ctx = &cv;
    const int mbw = ctx->mbw, stride = ctx->stride, qp = ctx->qp;
    const int W = mbw * 16, H = ctx->mbh * 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pair = blockIdx.x * 4 + wave;
    const bool is_luma = lane < 32, is_chroma = lane >= 32 && lane < 48;
    const int sel = is_luma ? lane >> 4 : (is_chroma ? (lane - 32) >> 3 : 0);
    int mbn = mb0 + pair * 2 + sel; // the launch covers macroblocks mb0 .. mb1-1
    const bool mb_ok = mbn < mb1;
    if (!mb_ok) mbn = mb1 - 1;
    const int my = mbn / mbw, mx = mbn - my * mbw, x0 = mx * 16, y0 = my * 16;
    const mb_info_t info = ld_mbinfo(&ctx->mbi[mbn]);
    // quarter-sample vector; the clamp only guards against garbage records (real vectors are far inside it)
    const int mvx = clip3(-4 * (x0 + 24), 4 * (W - x0 + 8), info.mvx), mvy = clip3(-4 * (y0 + 24), 4 * (H - y0 + 8), info.mvy);
    int flags = 0; // bit0: AC/any nonzero, bit1: chroma DC nonzero
    __shared__ int t8tile[4][8][64]; // [wave][8x8 block of the wave's two macroblocks][8x8], used only by the 8x8 transform path
    const bool t8 = ctx->t8 != 0;
    if (t8) { // High profile: every P macroblock through the 8x8 transform.  Four lanes per 8x8 block, two rows each;
              // the separable passes alternate rows/columns through a per-block LDS tile (same-wave traffic only).
        int pr8[2][8], rs[2][8], cw[2][8];
        unsigned submask = 0; // non-zero 4x4 "sub-blocks" (scan positions 4k+j) this lane has seen
        int *tile = t8tile[wave][(lane >> 2) & 7];
        const int i8 = (lane >> 2) & 3, j = lane & 3;
        const int bx8 = x0 + (i8 & 1) * 8, by8 = y0 + (i8 >> 1) * 8;
        const int m6 = qp % 6, k6 = qp / 6;
        if (is_luma && mb_ok) {
            const uint8_t *__restrict__ s = ctx->src_y;
            const uint8_t *__restrict__ rf = ctx->ref_y;
            const int ss = ctx->src_stride, vh = ctx->vis_h;
            const int fx = mvx & 3, fy = mvy & 3, X = bx8 + (mvx >> 2), Y = by8 + 2 * j + (mvy >> 2);
            if (fx == 0 && fy == 0 && X >= 0 && Y >= 0 && X + 8 <= W && Y + 2 <= H) {
#pragma unroll
                for (int r = 0; r < 2; r++) {
                    size_t a = (size_t)(Y + r) * stride + X;
                    const unsigned *ap = (const unsigned *)(rf + (a & ~(size_t)3));
                    const unsigned w0 = ldg32(ap), w1 = ldg32(ap + 1), w2 = ldg32(ap + 2);
                    const unsigned lo = __builtin_amdgcn_alignbyte(w1, w0, (unsigned)(a & 3)), hi = __builtin_amdgcn_alignbyte(w2, w1, (unsigned)(a & 3));
#pragma unroll
                    for (int i = 0; i < 4; i++) { pr8[r][i] = byte_of(lo, i); pr8[r][4 + i] = byte_of(hi, i); }
                }
            } else {
                int n[7][13];
#pragma unroll
                for (int r = 0; r < 7; r++) {
                    const int yy = clip3(0, H - 1, Y - 2 + r);
#pragma unroll
                    for (int c2 = 0; c2 < 13; c2++) n[r][c2] = (int)ldg8(rf + (size_t)yy * stride + clip3(0, W - 1, X - 2 + c2));
                }
#pragma unroll
                for (int r = 0; r < 2; r++)
#pragma unroll
                    for (int i = 0; i < 8; i++) pr8[r][i] = qpel_nb<13>(n, i, r, fx, fy);
            }
#pragma unroll
            for (int r = 0; r < 2; r++) {
                int sy = by8 + 2 * j + r;
                sy = sy < vh ? sy : vh - 1;
                const uint2 sw = ldg64(s + (size_t)sy * ss + bx8);
#pragma unroll
                for (int i = 0; i < 4; i++) { rs[r][i] = byte_of(sw.x, i) - pr8[r][i]; rs[r][4 + i] = byte_of(sw.y, i) - pr8[r][4 + i]; }
                fdct8_1d(rs[r]);
#pragma unroll
                for (int i = 0; i < 8; i++) tile[(2 * j + r) * 8 + i] = rs[r][i];
            }
        }
        WAVE_SYNC();
        if (is_luma && mb_ok) { // columns 2j, 2j+1: second forward pass, quantise, scale
#pragma unroll
            for (int c2 = 0; c2 < 2; c2++) {
#pragma unroll
                for (int r = 0; r < 8; r++) cw[c2][r] = tile[r * 8 + 2 * j + c2];
                fdct8_1d(cw[c2]);
#pragma unroll
                for (int r = 0; r < 8; r++) {
                    const int xx = 2 * j + c2, cl = pos_class8(r, xx);
                    const int qbits = 16 + k6, f = (1 << qbits) / 6;
                    const int a = iabs(cw[c2][r]);
                    int l = (int)(((long long)a * g_tab.mf8[m6][cl] + f) >> qbits);
                    l = l > 2047 ? 2047 : l;
                    l = cw[c2][r] < 0 ? -l : l;
                    const int kk = g_tab.izz8[r * 8 + xx];
                    stg16(&ctx->levels[(size_t)mbn * MB_LEVELS + L_LUMA + (4 * i8 + (kk & 3)) * 16 + (kk >> 2)], l);
                    if (l) submask |= 1u << (kk & 3);
                    const int ls = 16 * g_tab.v8[m6][cl];
                    cw[c2][r] = qp >= 36 ? (l * ls) << (k6 - 6) : (l * ls + (1 << (5 - k6))) >> (6 - k6);
                }
            }
        }
        WAVE_SYNC();
        if (is_luma && mb_ok) {
#pragma unroll
            for (int c2 = 0; c2 < 2; c2++)
#pragma unroll
                for (int r = 0; r < 8; r++) tile[r * 8 + 2 * j + c2] = cw[c2][r];
        }
        // the 4-bit sub-block mask of the 8x8 block: OR over its four lanes
        submask |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)submask, 0xB1, 0xF, 0xF, false);
        submask |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)submask, 0x4E, 0xF, 0xF, false);
        WAVE_SYNC();
        if (is_luma && mb_ok) { // 8.5.13: rows first ...
#pragma unroll
            for (int r = 0; r < 2; r++) {
#pragma unroll
                for (int i = 0; i < 8; i++) rs[r][i] = tile[(2 * j + r) * 8 + i];
                idct8_1d(rs[r]);
            }
        }
        WAVE_SYNC();
        if (is_luma && mb_ok) {
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int i = 0; i < 8; i++) tile[(2 * j + r) * 8 + i] = rs[r][i];
        }
        WAVE_SYNC();
        if (is_luma && mb_ok) { // ... then columns, rounding
#pragma unroll
            for (int c2 = 0; c2 < 2; c2++) {
#pragma unroll
                for (int r = 0; r < 8; r++) cw[c2][r] = tile[r * 8 + 2 * j + c2];
                idct8_1d(cw[c2]);
            }
        }
        WAVE_SYNC();
        if (is_luma && mb_ok) {
#pragma unroll
            for (int c2 = 0; c2 < 2; c2++)
#pragma unroll
                for (int r = 0; r < 8; r++) tile[r * 8 + 2 * j + c2] = (cw[c2][r] + 32) >> 6;
        }
        WAVE_SYNC();
        if (is_luma && mb_ok) {
            uint8_t *__restrict__ rec = ctx->rec_y;
#pragma unroll
            for (int r = 0; r < 2; r++) {
                int o[8];
#pragma unroll
                for (int i = 0; i < 8; i++) o[i] = clip255(pr8[r][i] + tile[(2 * j + r) * 8 + i]);
                uint8_t *dst = rec + (size_t)(by8 + 2 * j + r) * stride + bx8;
                stg32(dst, pack4(o[0], o[1], o[2], o[3]));
                stg32(dst + 4, pack4(o[4], o[5], o[6], o[7]));
            }
            flags = (submask >> j) & 1; // lane 4*i8 + j reports sub-block j, which is blkIdx 4*i8 + j
        }
    } else if (is_luma && mb_ok) {
        const int b = lane & 15, bx = blkx(b), by = blky(b);
        const qparams q = make_q(&g_tab, qp, false);
        int x[16], pr[16], lev[16];
        const uint8_t *__restrict__ s = ctx->src_y;
        const uint8_t *__restrict__ rf = ctx->ref_y;
        const int ss = ctx->src_stride, vh = ctx->vis_h;
        const int fx = mvx & 3, fy = mvy & 3, X = x0 + bx + (mvx >> 2), Y = y0 + by + (mvy >> 2);
        if (fx == 0 && fy == 0 && X >= 0 && Y >= 0 && X + 4 <= W && Y + 4 <= H) { // whole-sample vector, block inside the picture
#pragma unroll
            for (int r = 0; r < 4; r++) {
                size_t a = (size_t)(Y + r) * stride + X;
                const uint2 apw = ldg64x(rf + (a & ~(size_t)3));
                unsigned pw = __builtin_amdgcn_alignbyte(apw.y, apw.x, (unsigned)(a & 3));
#pragma unroll
                for (int i = 0; i < 4; i++) pr[r * 4 + i] = byte_of(pw, i);
            }
        } else { // 8.4.2.2.1: 6-tap / averaged samples from a 9 x 9 neighbourhood, picture extended by clamping
            int n[9][9];
#pragma unroll
            for (int r = 0; r < 9; r++) {
                const int yy = clip3(0, H - 1, Y - 2 + r);
#pragma unroll
                for (int c2 = 0; c2 < 9; c2++) n[r][c2] = (int)ldg8(rf + (size_t)yy * stride + clip3(0, W - 1, X - 2 + c2));
            }
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int i = 0; i < 4; i++) pr[r * 4 + i] = qpel_from9(n, i, r, fx, fy);
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {
            int sy = y0 + by + r;
            sy = sy < vh ? sy : vh - 1;
            unsigned sw = ldg32(s + (size_t)sy * ss + x0 + bx);
#pragma unroll
            for (int i = 0; i < 4; i++) x[r * 4 + i] = byte_of(sw, i) - pr[r * 4 + i];
        }
        fdct4(x);
        bool nz = quant_dequant<0>(x, lev, q);
        store_levels(ctx->levels + (size_t)mbn * MB_LEVELS + L_LUMA + b * 16, lev);
        idct4(x);
        uint8_t *__restrict__ rec = ctx->rec_y;
#pragma unroll
        for (int r = 0; r < 4; r++)
            stg32(rec + (size_t)(y0 + by + r) * stride + x0 + bx,
                  pack4(clip255(pr[r * 4] + x[r * 4]), clip255(pr[r * 4 + 1] + x[r * 4 + 1]),
                        clip255(pr[r * 4 + 2] + x[r * 4 + 2]), clip255(pr[r * 4 + 3] + x[r * 4 + 3])));
        flags = nz ? 1 : 0;
    }
    if (is_chroma) { // all 16 lanes run (shuffles inside); stores are predicated by mb_ok via mbn clamp
        const int cl = (lane - 32) & 7, c = cl >> 2, b = cl & 3, bx = (b & 1) * 4, by = (b >> 1) * 4;
        const int cx0 = x0 >> 1, cy0 = y0 >> 1, cw = W >> 1, ch = H >> 1;
        // 8.4.1.4 / 8.4.2.2.2: the chroma vector is the luma vector read in 1/8 chroma-sample units
        const int xi = mvx >> 3, yi = mvy >> 3, xf = mvx & 7, yf = mvy & 7;
        const uint8_t *__restrict__ rf = ctx->ref_uv;
        int smp[5][5];
#pragma unroll
        for (int r = 0; r < 5; r++) {
            int yy = clip3(0, ch - 1, cy0 + by + r + yi);
#pragma unroll
            for (int i = 0; i < 5; i++) {
                int xx = clip3(0, cw - 1, cx0 + bx + i + xi);
                smp[r][i] = (int)ldg8(rf + (size_t)yy * stride + 2 * xx + c);
            }
        }
        int pr[16];
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int i = 0; i < 4; i++)
                pr[r * 4 + i] = ((8 - xf) * (8 - yf) * smp[r][i] + xf * (8 - yf) * smp[r][i + 1] +
                                 (8 - xf) * yf * smp[r + 1][i] + xf * yf * smp[r + 1][i + 1] + 32) >> 6;
        if (mb_ok) flags = chroma_block(ctx, &g_tab, mbn, cx0, cy0, cl, pr, qp, false); // (a quad is one macroblock's plane: valid or not as a whole)
    }
    const unsigned long long any = __ballot(flags & 1), dcm = __ballot(flags & 2);
    if ((lane == 0 || lane == 16) && mb_ok) {
        const int s2 = lane >> 4;
        unsigned nzm = (unsigned)((any >> (16 * s2)) & 0xFFFF) | ((unsigned)((any >> (32 + 8 * s2)) & 0xFF) << 16);
        if ((dcm >> (32 + 8 * s2)) & 0x0F) nzm |= NZ_CBDC;
        if ((dcm >> (32 + 8 * s2)) & 0xF0) nzm |= NZ_CRDC;
        if (t8 && (nzm & 0xFFFF)) nzm |= NZ_T8; // transform_size_8x8_flag exists only with luma cbp != 0
        mb_info_t *mb = &ctx->mbi[mbn];
        stg32(&mb->mb_type, 1u | ((unsigned)qp << 24)); // mb_type 1, modes 0, qp
        stg32(&mb->nzmask, nzm);
    }
    // luma DC levels are unused by P macroblocks but part of the record: keep them zero
    if (is_luma && mb_ok && (lane & 15) < 2) {
        uint4 z = make_uint4(0, 0, 0, 0);
        stg128(ctx->levels + (size_t)mbn * MB_LEVELS + L_LDC + 8 * (lane & 15), z);
    }
}

// sum over each row of 16 lanes, result in every lane: four DPP adds (no LDS crossbar round trips)
// value of lane k of this lane's quad (k = 0..3), and of row r of this lane's column in a 4x4 tile laid out on 16 lanes
template <int K> DEV int quad_bcast(int v) { return __builtin_amdgcn_update_dpp(0, v, K * 0x55, 0xF, 0xF, false); }

// =================================================================== intra analysis (open loop)
// SAD of every intra candidate of every macroblock, with predictions built from the SOURCE picture's
// neighbouring samples: no macroblock depends on another, so this is one flat launch (one wave per
// macroblock) instead of work inside the reconstruction wavefront.  Oracle: orc_intra_analyse.
DEV int wave16_min(int v) {
    int o;
    o = __builtin_amdgcn_update_dpp(0, v, 0x128, 0xF, 0xF, false); v = v < o ? v : o;
    o = __builtin_amdgcn_update_dpp(0, v, 0x124, 0xF, 0xF, false); v = v < o ? v : o;
    o = __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false);  v = v < o ? v : o;
    o = __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false);  v = v < o ? v : o;
    return v;
}
// one Intra_4x4 prediction sample (8.3.1.2) for pixel (px,py); E(i): ... l1 l0 | corner | t0 .. t7 with the
// top-right substitution already applied; dc4: the block's DC value
template <typename EF>
DEV int pred4_px(int md, int px, int py, EF E, int dc4) {
    if (md == 0) return E(px + 1);
    if (md == 1) return E(-(py + 1));
    if (md == 2) return dc4;
    if (md == 3) return (px == 3 && py == 3) ? (E(7) + 3 * E(8) + 2) >> 2 : (E(px + py + 1) + 2 * E(px + py + 2) + E(px + py + 3) + 2) >> 2;
    if (md == 4) return (E(px - py - 1) + 2 * E(px - py) + E(px - py + 1) + 2) >> 2;
    if (md == 5) {
        const int z = 2 * px - py, k = px - (py >> 1);
        return (z >= 0 && !(z & 1)) ? (E(k) + E(k + 1) + 1) >> 1 : z >= 0 ? (E(k - 1) + 2 * E(k) + E(k + 1) + 2) >> 2
               : z == -1 ? (E(-1) + 2 * E(0) + E(1) + 2) >> 2 : (E(-py) + 2 * E(-py + 1) + E(-py + 2) + 2) >> 2;
    }
    if (md == 6) {
        const int z = 2 * py - px, k = py - (px >> 1);
        return (z >= 0 && !(z & 1)) ? (E(-k) + E(-k - 1) + 1) >> 1 : z >= 0 ? (E(-k + 1) + 2 * E(-k) + E(-k - 1) + 2) >> 2
               : z == -1 ? (E(-1) + 2 * E(0) + E(1) + 2) >> 2 : (E(px) + 2 * E(px - 1) + E(px - 2) + 2) >> 2;
    }
    if (md == 7) {
        const int k = px + (py >> 1);
        return !(py & 1) ? (E(k + 1) + E(k + 2) + 1) >> 1 : (E(k + 1) + 2 * E(k + 2) + E(k + 3) + 2) >> 2;
    }
    const int z = px + 2 * py, k = py + (px >> 1);
    return z > 5 ? E(-4) : z == 5 ? (E(-3) + 3 * E(-4) + 2) >> 2 : !(z & 1) ? (E(-(k + 1)) + E(-(k + 2)) + 1) >> 1
           : (E(-(k + 1)) + 2 * E(-(k + 2)) + E(-(k + 3)) + 2) >> 2;
}
DEV bool mode4_ok(int b, int md, bool up, bool lf, bool ul) {
    const bool need_up = md == 0 || md == 3 || md == 7, need_left = md == 1 || md == 8, need_all = md >= 4 && md <= 6;
    return !((need_up && !up) || (need_left && !lf) || (need_all && !(up && lf && ul)) || (b == 5 && (md == 3 || md == 7)));
}
#define IA_S 24 /* luma tile stride: row 0 = y -1, col 0 = x -1 */
__global__ __launch_bounds__(256) void intra_analyse_kernel(const frame_ctx_t *__restrict__ ctx) {
    __shared__ uint16_t sh_sad[4][ISAD_PER_MB]; // this wave's macroblock: the same 152 values that go to ctx->isad
    __shared__ int sh_m4[4][16];                // Intra_4x4 modes chosen so far, raster order
    __shared__ __attribute__((aligned(4))) uint8_t SL[4][17 * IA_S];
    __shared__ __attribute__((aligned(4))) uint8_t SC[4][2][9 * 12]; // [plane][row 0 = y -1][col 0 = x -1]
    const int mbw = ctx->mbw, nmb = mbw * ctx->mbh;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int mbn = blockIdx.x * 4 + wave;
    const bool ok = mbn < nmb;
    if (!ok) mbn = nmb - 1;
    const int my = mbn / mbw, mx = mbn - my * mbw, x0 = mx * 16, y0 = my * 16, cx0 = x0 >> 1, cy0 = y0 >> 1;
    const bool has_top = my > 0, has_left = mx > 0;
    const uint8_t *__restrict__ sy = ctx->src_y;
    const uint8_t *__restrict__ suv = ctx->src_uv;
    const int ss = ctx->src_stride, vh = ctx->vis_h, vh2 = vh >> 1;
    uint8_t *S = SL[wave];
    // ---- source tile with its one-sample apron (rows beyond the visible picture repeat the last row, like every source read)
    {
        const int r = lane >> 2, q = lane & 3; // interior: 16 rows x 4 dwords
        int yy = y0 + r; yy = yy < vh ? yy : vh - 1;
        const unsigned w = ldg32(sy + (size_t)yy * ss + x0 + 4 * q);
#pragma unroll
        for (int i = 0; i < 4; i++) S[(r + 1) * IA_S + 1 + 4 * q + i] = (uint8_t)byte_of(w, i);
        if (lane < 17) { // top row incl. corner
            const int x = lane - 1;
            int yt = y0 - 1; yt = yt < vh ? yt : vh - 1;
            S[lane] = (has_top && (x >= 0 || has_left)) ? (uint8_t)ldg8(sy + (size_t)yt * ss + x0 + x) : 0;
        } else if (lane < 33) { // left column
            int yl = y0 + lane - 17; yl = yl < vh ? yl : vh - 1;
            S[(lane - 16) * IA_S] = has_left ? (uint8_t)ldg8(sy + (size_t)yl * ss + x0 - 1) : 0;
        }
        // chroma: interior 8 rows x 16 bytes (both planes interleaved) = 32 dwords
        if (lane < 32) {
            const int cr = lane >> 2, cq = lane & 3;
            int yc = cy0 + cr; yc = yc < vh2 ? yc : vh2 - 1;
            const unsigned cwd = ldg32(suv + (size_t)yc * ss + 2 * cx0 + 4 * cq);
            SC[wave][0][(cr + 1) * 12 + 1 + 2 * cq] = (uint8_t)byte_of(cwd, 0); SC[wave][1][(cr + 1) * 12 + 1 + 2 * cq] = (uint8_t)byte_of(cwd, 1);
            SC[wave][0][(cr + 1) * 12 + 2 + 2 * cq] = (uint8_t)byte_of(cwd, 2); SC[wave][1][(cr + 1) * 12 + 2 + 2 * cq] = (uint8_t)byte_of(cwd, 3);
        } else if (lane < 32 + 18) { // top rows incl. corner, both planes
            const int c = (lane - 32) / 9, x = (lane - 32) % 9 - 1;
            int yt = cy0 - 1; yt = yt < vh2 ? yt : vh2 - 1;
            SC[wave][c][x + 1] = (has_top && (x >= 0 || has_left)) ? (uint8_t)ldg8(suv + (size_t)yt * ss + 2 * (cx0 + x) + c) : 0;
        } else if (lane < 32 + 18 + 14) { // left columns, rows 0..6 of both planes (row 7 below)
            const int c = (lane - 50) / 7, y = (lane - 50) % 7;
            int yl = cy0 + y; yl = yl < vh2 ? yl : vh2 - 1;
            SC[wave][c][(y + 1) * 12] = has_left ? (uint8_t)ldg8(suv + (size_t)yl * ss + 2 * (cx0 - 1) + c) : 0;
        }
        if (lane < 2) {
            int yl = cy0 + 7; yl = yl < vh2 ? yl : vh2 - 1;
            SC[wave][lane][8 * 12] = has_left ? (uint8_t)ldg8(suv + (size_t)yl * ss + 2 * (cx0 - 1) + lane) : 0;
        }
    }
    WAVE_SYNC();
    uint16_t *out = ctx->isad + (size_t)mbn * ISAD_PER_MB;
    const unsigned NA = 0xFFFFu;
    // ---- Intra_16x16: lane = row lane>>2, columns 4*(lane&3)..+3
    {
        const int tl = lane < 16 ? S[lane + 1] : 0, ll = lane < 16 ? S[(lane + 1) * IA_S] : 0;
        const int st = __shfl(wave16_sum(tl), 0), sl = __shfl(wave16_sum(ll), 0);
        int hterm = 0, vterm = 0;
        if (lane < 8) {
            hterm = (lane + 1) * ((int)S[8 + lane + 1] - (int)S[6 - lane + 1]);                 // x' = lane: p[8+x',-1] - p[6-x',-1] (6-7 = -1 is the corner, col 0)
            vterm = (lane + 1) * ((int)S[(8 + lane + 1) * IA_S] - (int)S[(6 - lane + 1) * IA_S]);
        }
        const int Hh = __shfl(wave16_sum(hterm), 0), Vv = __shfl(wave16_sum(vterm), 0);
        const int dcv = (has_top && has_left) ? (st + sl + 16) >> 5 : has_top ? (st + 8) >> 4 : has_left ? (sl + 8) >> 4 : 128;
        const int pa = 16 * ((int)S[16 * IA_S] + (int)S[16]), pb = (5 * Hh + 32) >> 6, pc = (5 * Vv + 32) >> 6;
        const int r = lane >> 2, c0 = (lane & 3) * 4;
        int s0 = 0, s1 = 0, s2 = 0, s3 = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int sv = S[(r + 1) * IA_S + c0 + i + 1];
            s0 += iabs(sv - (int)S[c0 + i + 1]);
            s1 += iabs(sv - (int)S[(r + 1) * IA_S]);
            s2 += iabs(sv - dcv);
            s3 += iabs(sv - clip255((pa + pb * (c0 + i - 7) + pc * (r - 7) + 16) >> 5));
        }
        int packed01 = wave16_sum(s0 | (s1 << 16)), packed23 = wave16_sum(s2 | (s3 << 16)); // row sums <= 16*255 fit 16 bits each
        packed01 += __shfl_xor(packed01, 16); packed23 += __shfl_xor(packed23, 16); // 32 lanes: <= 8160
        const unsigned a01 = (unsigned)packed01, a23 = (unsigned)packed23;
        const unsigned o01 = (unsigned)__shfl_xor(packed01, 32), o23 = (unsigned)__shfl_xor(packed23, 32);
        const unsigned t0 = (a01 & 0xFFFF) + (o01 & 0xFFFF), t1 = (a01 >> 16) + (o01 >> 16), t2 = (a23 & 0xFFFF) + (o23 & 0xFFFF), t3 = (a23 >> 16) + (o23 >> 16);
        if (lane == 0 && ok) {
            const unsigned v0 = has_top ? t0 : NA, v1 = has_left ? t1 : NA, v3 = (has_top && has_left) ? t3 : NA;
            stg16(out + 0, (int)v0); stg16(out + 1, (int)v1); stg16(out + 2, (int)t2); stg16(out + 3, (int)v3);
            sh_sad[wave][0] = (uint16_t)v0; sh_sad[wave][1] = (uint16_t)v1; sh_sad[wave][2] = (uint16_t)t2; sh_sad[wave][3] = (uint16_t)v3;
        }
    }
    // ---- chroma 8x8 (both planes): lane = plane lane>>5, row (lane>>2)&7, columns 2*(lane&3)..+1
    {
        const int c = lane >> 5, r = (lane >> 2) & 7, c0 = (lane & 3) * 2;
        const uint8_t *P = SC[wave][c];
        int s0 = 0, s1 = 0, s2 = 0, s3 = 0;
        int Hh = 0, Vv = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) { Hh += (i + 1) * ((int)P[4 + i + 1] - (int)P[2 - i + 1]); Vv += (i + 1) * ((int)P[(4 + i + 1) * 12] - (int)P[(2 - i + 1) * 12]); }
        const int pa = 16 * ((int)P[8 * 12] + (int)P[8]), pb = (34 * Hh + 32) >> 6, pc = (34 * Vv + 32) >> 6;
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int x = c0 + i, sv = P[(r + 1) * 12 + x + 1];
            const int qx = x >> 2, qy = r >> 2; // 8.3.4.1-3 DC per 4x4 quadrant
            int stq = 0, slq = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) { stq += P[qx * 4 + k + 1]; slq += P[(qy * 4 + k + 1) * 12]; }
            bool ut = has_top, ul = has_left;
            if (qx == 1 && qy == 0 && has_top) ul = false;
            if (qx == 0 && qy == 1 && has_left) ut = false;
            const int dcv = (ut && ul) ? (stq + slq + 4) >> 3 : ut ? (stq + 2) >> 2 : ul ? (slq + 2) >> 2 : 128;
            s0 += iabs(sv - dcv);
            s1 += iabs(sv - (int)P[(r + 1) * 12]);
            s2 += iabs(sv - (int)P[x + 1]);
            s3 += iabs(sv - clip255((pa + pb * (x - 3) + pc * (r - 3) + 16) >> 5));
        }
        int p01 = wave16_sum(s0 | (s1 << 16)), p23 = wave16_sum(s2 | (s3 << 16));
        p01 += __shfl_xor(p01, 16); p23 += __shfl_xor(p23, 16);
        const unsigned a01 = (unsigned)p01, a23 = (unsigned)p23, o01 = (unsigned)__shfl_xor(p01, 32), o23 = (unsigned)__shfl_xor(p23, 32);
        const unsigned t0 = (a01 & 0xFFFF) + (o01 & 0xFFFF), t1 = (a01 >> 16) + (o01 >> 16), t2 = (a23 & 0xFFFF) + (o23 & 0xFFFF), t3 = (a23 >> 16) + (o23 >> 16);
        if (lane == 0 && ok) {
            const unsigned v1 = has_left ? t1 : NA, v2 = has_top ? t2 : NA, v3 = (has_top && has_left) ? t3 : NA;
            stg16(out + 4, (int)t0); stg16(out + 5, (int)v1); stg16(out + 6, (int)v2); stg16(out + 7, (int)v3);
            sh_sad[wave][4] = (uint16_t)t0; sh_sad[wave][5] = (uint16_t)v1; sh_sad[wave][6] = (uint16_t)v2; sh_sad[wave][7] = (uint16_t)v3;
        }
    }
    // ---- Intra_4x4: four blocks at a time, 16 lanes (pixels) each
    {
        const int px = lane & 3, py = (lane >> 2) & 3;
#pragma unroll 1
        for (int rnd = 0; rnd < 4; rnd++) {
            const int b = rnd * 4 + (lane >> 4);
            const int bx = blkx(b) >> 2, by = blky(b) >> 2;
            const bool up = by > 0 || has_top, lf = bx > 0 || has_left;
            const bool ul = (bx > 0 && by > 0) ? true : bx > 0 ? has_top : by > 0 ? has_left : (has_top && has_left);
            const int trb = by > 0 && bx < 3 ? ((((by - 1) >> 1) << 3) | (((bx + 1) >> 1) << 2) | (((by - 1) & 1) << 1) | ((bx + 1) & 1)) : 99;
            const bool ur = by == 0 ? (bx < 3 && has_top) : (bx < 3 && trb < b);
            const int emax = ur ? 8 : 4;
            const uint8_t *tb = &S[(by * 4) * IA_S + bx * 4];
            auto E = [&](int i) -> int { i = i > emax ? emax : i; return i < 0 ? (int)tb[(-i) * IA_S] : (int)tb[i]; };
            const int sv = S[(by * 4 + py + 1) * IA_S + bx * 4 + px + 1];
            const int sumT = E(1) + E(2) + E(3) + E(4), sumL = E(-1) + E(-2) + E(-3) + E(-4);
            const int dc4 = (up && lf) ? (sumT + sumL + 4) >> 3 : lf ? (sumL + 2) >> 2 : up ? (sumT + 2) >> 2 : 128;
#pragma unroll
            for (int md = 0; md < 9; md++) {
                const int sad = wave16_sum(iabs(sv - pred4_px(md, px, py, E, dc4)));
                if ((lane & 15) == 0) {
                    const unsigned v = mode4_ok(b, md, up, lf, ul) ? (unsigned)sad : NA;
                    if (ok) stg16(out + 8 + b * 9 + md, (int)v);
                    sh_sad[wave][8 + b * 9 + md] = (uint16_t)v;
                }
            }
        }
    }
    // ---- decisions (oracle: orc_intra_decide): nothing outside this macroblock is needed, so they are taken here, in
    // the flat launch, and the reconstruction wavefront only reads the 24-byte result.
    WAVE_SYNC();
    {
        const uint16_t *isad = sh_sad[wave];
        const unsigned BIG = 0x10000000u;
        const int lam = ctx->lambda;
        int mode16 = 0, cmode = 0;
        unsigned cost16 = BIG, costc = BIG;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const unsigned a = isad[q], c = isad[4 + q];
            if (a != 0xFFFFu && a < cost16) { cost16 = a; mode16 = q; }
            if (c != 0xFFFFu && c < costc) { costc = c; cmode = q; }
        }
        bool use_i4 = false;
        unsigned cost_luma = cost16;
        if (ctx->i4x4) { // Intra_4x4 modes block by block: SAD + lambda * (mode == expected ? 1 : 4); blocks visited along bx + 2*by
            unsigned cost4 = 0;
            const int half = (lane >> 4) & 1, cand = lane & 15;
            int (*m4)[16] = &sh_m4[wave];
#pragma unroll 1
            for (int s4 = 0; s4 < 10; s4++) {
                const int by_lo = s4 > 3 ? (s4 - 2) >> 1 : 0, by_hi = (s4 >> 1) < 3 ? (s4 >> 1) : 3;
                const bool two = by_lo + 1 <= by_hi;
                const bool valid = lane < 32 && (half == 0 || two);
                const int by = (valid && half) ? by_lo + 1 : by_lo, bx = s4 - 2 * by;
                const int b = ((by >> 1) << 3) | ((bx >> 1) << 2) | ((by & 1) << 1) | (bx & 1);
                const int ma = bx > 0 ? (*m4)[by * 4 + bx - 1] : (has_left ? 2 : -1), mb_ = by > 0 ? (*m4)[(by - 1) * 4 + bx] : (has_top ? 2 : -1);
                const int pm = (ma < 0 || mb_ < 0) ? 2 : (ma < mb_ ? ma : mb_);
                unsigned key = 0x7FFFFFFFu;
                if (cand < 9) {
                    const unsigned sd = isad[8 + b * 9 + cand];
                    if (sd != 0xFFFFu) key = ((sd + (unsigned)(lam * (cand == pm ? 1 : 4))) << 4) | (unsigned)cand;
                }
                key = (unsigned)wave16_min((int)key);
                if (valid && cand == 0) (*m4)[by * 4 + bx] = (int)(key & 15);
                cost4 += (unsigned)__shfl((int)(key >> 4), 0, 64) + (two ? (unsigned)__shfl((int)(key >> 4), 16, 64) : 0u);
                WAVE_SYNC();
            }
            use_i4 = cost4 + (unsigned)(32 * lam) < cost16;
            if (use_i4) cost_luma = cost4 + (unsigned)(32 * lam);
        }
        if (lane < 6 && ok) { // 24-byte record {u8 modes4[16] by blkIdx; u8 mode16, cmode, use_i4, 0; u32 cost}
            unsigned w = 0;
            if (lane < 4) {
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int bb = lane * 4 + i, r = (blky(bb) >> 2) * 4 + (blkx(bb) >> 2);
                    w |= (ctx->i4x4 ? (unsigned)sh_m4[wave][r] & 0xFF : 0u) << (8 * i);
                }
            } else if (lane == 4) w = (unsigned)mode16 | ((unsigned)cmode << 8) | ((use_i4 ? 1u : 0u) << 16);
            else w = cost_luma + costc;
            stg32(ctx->idec + (size_t)mbn * IDEC_BYTES + 4 * lane, w);
        }
    }
}

// =================================================================== intra (I) macroblocks
// One wave per macroblock, launched once per anti-diagonal x + y = diag (left, top and
// top-left neighbours are then complete).  Lanes 0-15: luma 4x4 blocks; lanes 16-23: chroma.
// Per-macroblock working set of the intra reconstruction, in LDS.  Filled by the caller: top / left (reconstructed
// neighbours, [plane 0 = Y, 1 = Cb, 2 = Cr][index i + 1 holds sample i, index 0 the corner]).  Produced for the neighbours
// to the right and below (persistent kernel): bottom rows into a 4-deep ring, the right column.
struct intra_lds {
    int top[3][17], left[3][17];
    int dc[16], ldc[16];
    __attribute__((aligned(4))) uint8_t T4[17 * 24]; // Intra_4x4: reconstructed samples incl. the row above / column left
    __attribute__((aligned(4))) uint8_t S4[256];     // source macroblock, raster
    __attribute__((aligned(4))) uint8_t crec[8 * 16]; // reconstructed chroma, interleaved Cb Cr (OUT only)
    int mode4[16];
    unsigned cflags[2];                               // chroma wave -> luma wave: ballots of its blocks' AC / DC flags
    unsigned cseq;                                    // ... valid once this equals macroblock number + 1
    __attribute__((aligned(4))) uint8_t bot_y[4][16], bot_c[4][16];
    __attribute__((aligned(4))) uint8_t right_y[16], right_c[2][8];
    int corner[3];                                    // bottom-right sample of the macroblock before the one in right_*: the next corner
};

// Reconstruction of one intra macroblock by two waves (wave 0 luma, wave 1 chroma; the planes share nothing after the
// decisions).  Needs L->top / L->left in place and visible; dec0/dec1: the 24-byte decision of intra_analyse_kernel.
template <bool OUT>
DEV void intra_compute(const frame_ctx_t *__restrict__ ctx, const dev_tables *T, intra_lds *L, const int mx, const int my, const int wave, const int lane,
                       const uint4 dec0, const uint2 dec1, const uint2 *presrc = nullptr) { // presrc: this lane's source rows, loaded ahead (luma: .x of 4; chroma: 4 pairs)
    int (*top)[17] = L->top;
    int (*left)[17] = L->left;
    int *sh_dc = L->dc, *sh_ldc = L->ldc, *sh_mode4 = L->mode4;
    uint8_t *T4 = L->T4, *S4 = L->S4;
    const int mbw = ctx->mbw, stride = ctx->stride, qp = ctx->qp;
    const int mbn = my * mbw + mx, x0 = mx * 16, y0 = my * 16, cx0 = x0 >> 1, cy0 = y0 >> 1;
    const bool has_top = my > 0, has_left = mx > 0;
    uint8_t *__restrict__ ry = ctx->rec_y;
    const bool is_luma = wave == 0 && lane < 16, is_chroma = wave == 1 && lane >= 16 && lane < 24;
    const int slot = mx & 3;
    int src[16];
    if (is_luma || is_chroma) {
        const int ss = ctx->src_stride;
        if (is_luma) {
            const uint8_t *__restrict__ s = ctx->src_y;
            const int bx = blkx(lane), by = blky(lane), vh = ctx->vis_h;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                int sy = y0 + by + r;
                sy = sy < vh ? sy : vh - 1;
                unsigned sw = presrc ? presrc[r].x : ldg32(s + (size_t)sy * ss + x0 + bx);
#pragma unroll
                for (int i = 0; i < 4; i++) src[r * 4 + i] = byte_of(sw, i);
                *(unsigned *)&S4[(by + r) * 16 + bx] = sw;
            }
        } else {
            const uint8_t *__restrict__ s = ctx->src_uv;
            const int cl = lane & 7, c = cl >> 2, b = cl & 3, bx = (b & 1) * 4, by = (b >> 1) * 4, vh2 = ctx->vis_h >> 1;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                int sy = cy0 + by + r;
                sy = sy < vh2 ? sy : vh2 - 1;
                uint2 w = presrc ? presrc[r] : ldg64(s + (size_t)sy * ss + 2 * (cx0 + bx));
                unsigned lo = c ? (w.x >> 8) : w.x, hi = c ? (w.y >> 8) : w.y;
                src[r * 4 + 0] = (int)(lo & 255); src[r * 4 + 1] = (int)((lo >> 16) & 255);
                src[r * 4 + 2] = (int)(hi & 255); src[r * 4 + 3] = (int)((hi >> 16) & 255);
            }
        }
    }
#define TOP(p, i) top[p][(i) + 1]
#define LEFT(p, i) left[p][(i) + 1]
    // ---- decisions were taken by intra_analyse_kernel (oracle: orc_intra_decide)
    const int mode16 = (int)(dec1.x & 255), cmode = (int)((dec1.x >> 8) & 255);
    const bool use_i4 = ((dec1.x >> 16) & 255) != 0;
    unsigned nz4 = 0;
    if (use_i4 && wave == 0 && lane < 16) { // raster order for the reconstruction loop
        const int bb = ((lane >> 3) << 3) | (((lane & 3) >> 1) << 2) | (((lane >> 2) & 1) << 1) | (lane & 1); // raster (by = lane>>2, bx = lane&3) -> blkIdx
        const unsigned w = bb < 4 ? dec0.x : bb < 8 ? dec0.y : bb < 12 ? dec0.z : dec0.w;
        sh_mode4[lane] = (int)((w >> (8 * (bb & 3))) & 255);
    }
    WAVE_SYNC();
    int pred[16];
    int flags = 0;
    if (use_i4 && wave == 0) {
        // ================================================================ Intra_4x4 reconstruction (8.3.1.2 + 8.5)
        // Same block order; up to two blocks per step, 16 lanes each, lane = one pixel; transforms across lanes.
        if (lane < 17) T4[lane] = (uint8_t)TOP(0, lane - 1);
        else if (lane < 33) T4[(lane - 16) * 24] = (uint8_t)LEFT(0, lane - 17);
        const int half = (lane >> 4) & 1, px = lane & 3, py = (lane >> 2) & 3;
        const qparams q4 = make_q(T, qp, true);
        const int cl4 = (!(px & 1) && !(py & 1)) ? 0 : ((px & 1) && (py & 1)) ? 1 : 2;
        const int mf4 = cl4 == 0 ? q4.mf[0] : cl4 == 1 ? q4.mf[1] : q4.mf[2], v4 = cl4 == 0 ? q4.v[0] : cl4 == 1 ? q4.v[1] : q4.v[2];
        const int kz4 = (int)((0xFEA9DB83C7426510ull >> (4 * (py * 4 + px))) & 15); // raster -> zig-zag position
        WAVE_SYNC();
#pragma unroll 1
        for (int s4 = 0; s4 < 10; s4++) {
            const int by_lo = s4 > 3 ? (s4 - 2) >> 1 : 0, by_hi = (s4 >> 1) < 3 ? (s4 >> 1) : 3;
            const bool two = by_lo + 1 <= by_hi;
            const bool valid = lane < 32 && (half == 0 || two);
            const int by = (valid && half) ? by_lo + 1 : by_lo, bx = s4 - 2 * by;
            const int b = ((by >> 1) << 3) | ((bx >> 1) << 2) | ((by & 1) << 1) | (bx & 1); // blkIdx
            const bool up = by > 0 || has_top, lf = bx > 0 || has_left;
            const int trb = by > 0 && bx < 3 ? ((((by - 1) >> 1) << 3) | (((bx + 1) >> 1) << 2) | (((by - 1) & 1) << 1) | ((bx + 1) & 1)) : 99;
            const bool ur = by == 0 ? (bx < 3 && has_top) : (bx < 3 && trb < b);
            const int emax = ur ? 8 : 4;
            const uint8_t *tb = &T4[(by * 4) * 24 + bx * 4];
            auto E = [&](int i) -> int { i = i > emax ? emax : i; return i < 0 ? (int)tb[(-i) * 24] : (int)tb[i]; };
            const int bmode = sh_mode4[by * 4 + bx];
            const int sv = S4[(by * 4 + py) * 16 + bx * 4 + px];
            const int sumT = E(1) + E(2) + E(3) + E(4), sumL = E(-1) + E(-2) + E(-3) + E(-4);
            const int dc4 = (up && lf) ? (sumT + sumL + 4) >> 3 : lf ? (sumL + 2) >> 2 : up ? (sumT + 2) >> 2 : 128;
            const int bpred = pred4_px(bmode, px, py, E, dc4);
            // residual -> 4x4 core transform across the 16 lanes (rows, then columns)
            const int res = sv - bpred;
            const int cbase = lane & ~12;
            int a0 = quad_bcast<0>(res), a1 = quad_bcast<1>(res), a2 = quad_bcast<2>(res), a3 = quad_bcast<3>(res);
            int tr = px == 0 ? a0 + a1 + a2 + a3 : px == 1 ? 2 * a0 + a1 - a2 - 2 * a3 : px == 2 ? a0 - a1 - a2 + a3 : a0 - 2 * a1 + 2 * a2 - a3;
            a0 = __shfl(tr, cbase, 64); a1 = __shfl(tr, cbase + 4, 64); a2 = __shfl(tr, cbase + 8, 64); a3 = __shfl(tr, cbase + 12, 64);
            const int coef = py == 0 ? a0 + a1 + a2 + a3 : py == 1 ? 2 * a0 + a1 - a2 - 2 * a3 : py == 2 ? a0 - a1 - a2 + a3 : a0 - 2 * a1 + 2 * a2 - a3;
            const int lv4 = quant1(coef, mf4, q4.f, q4.qbits);
            // 8.5.12: scale, inverse transform (rows then columns), round
            const int dq = (lv4 * v4) << q4.shift;
            a0 = quad_bcast<0>(dq); a1 = quad_bcast<1>(dq); a2 = quad_bcast<2>(dq); a3 = quad_bcast<3>(dq);
            {
                const int e0 = a0 + a2, e1 = a0 - a2, e2 = (a1 >> 1) - a3, e3 = a1 + (a3 >> 1);
                tr = px == 0 ? e0 + e3 : px == 1 ? e1 + e2 : px == 2 ? e1 - e2 : e0 - e3;
            }
            a0 = __shfl(tr, cbase, 64); a1 = __shfl(tr, cbase + 4, 64); a2 = __shfl(tr, cbase + 8, 64); a3 = __shfl(tr, cbase + 12, 64);
            int rr;
            {
                const int e0 = a0 + a2, e1 = a0 - a2, e2 = (a1 >> 1) - a3, e3 = a1 + (a3 >> 1);
                rr = py == 0 ? e0 + e3 : py == 1 ? e1 + e2 : py == 2 ? e1 - e2 : e0 - e3;
            }
            const int recp = clip255(bpred + ((rr + 32) >> 6));
            const unsigned long long bal = __ballot(valid && lv4 != 0);
            const int b0 = ((by_lo >> 1) << 3) | (((s4 - 2 * by_lo) >> 1) << 2) | ((by_lo & 1) << 1) | ((s4 - 2 * by_lo) & 1);
            if (bal & 0xFFFFull) nz4 |= 1u << b0;
            if (two) {
                const int by1 = by_lo + 1, bx1 = s4 - 2 * by1, b1 = ((by1 >> 1) << 3) | ((bx1 >> 1) << 2) | ((by1 & 1) << 1) | (bx1 & 1);
                if (bal & 0xFFFF0000ull) nz4 |= 1u << b1;
            }
            if (valid) {
                T4[(by * 4 + py + 1) * 24 + bx * 4 + px + 1] = (uint8_t)recp;
                stg8(ry + (size_t)(y0 + by * 4 + py) * stride + x0 + bx * 4 + px, (unsigned)recp);
                stg16(&ctx->levels[(size_t)mbn * MB_LEVELS + L_LUMA + b * 16 + kz4], lv4);
                if ((lane & 15) == 0) stg16(&ctx->levels[(size_t)mbn * MB_LEVELS + L_LDC + b], bmode);
            }
            WAVE_SYNC();
        }
        if (OUT && lane < 16) { L->bot_y[slot][lane] = T4[16 * 24 + lane + 1]; L->right_y[lane] = T4[(lane + 1) * 24 + 16]; }
    } else if (is_luma) {
        // ================================================================ Intra_16x16 reconstruction (8.3.3 + 8.5.10)
        const int b = lane, bx = blkx(b), by = blky(b), mode = mode16;
        // neighbour statistics once per macroblock, reduced over the 16 lanes (lane j holds top j / left j): sums for DC,
        // the weighted sums of 8.3.3.4 for Plane (weights j - 7, and -8 for the corner)
        const int tj = TOP(0, lane), lj = LEFT(0, lane), cor = TOP(0, -1);
        const int st = wave16_sum(tj), sl = wave16_sum(lj);
        if (mode == 0) { // wave-uniform: only the chosen predictor is evaluated
#pragma unroll
            for (int i = 0; i < 4; i++) { const int t = TOP(0, bx + i); pred[i] = t; pred[4 + i] = t; pred[8 + i] = t; pred[12 + i] = t; }
        } else if (mode == 1) {
#pragma unroll
            for (int r = 0; r < 4; r++) { const int l = LEFT(0, by + r); pred[r * 4] = l; pred[r * 4 + 1] = l; pred[r * 4 + 2] = l; pred[r * 4 + 3] = l; }
        } else if (mode == 2) {
            const int dcv = (has_top && has_left) ? (st + sl + 16) >> 5 : has_top ? (st + 8) >> 4 : has_left ? (sl + 8) >> 4 : 128;
#pragma unroll
            for (int k = 0; k < 16; k++) pred[k] = dcv;
        } else {
            const int Hh = wave16_sum((lane - 7) * tj) - 8 * cor, Vv = wave16_sum((lane - 7) * lj) - 8 * cor;
            const int pa = 16 * (LEFT(0, 15) + TOP(0, 15)), pb = (5 * Hh + 32) >> 6, pc = (5 * Vv + 32) >> 6;
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int i = 0; i < 4; i++) pred[r * 4 + i] = clip255((pa + pb * (bx + i - 7) + pc * (by + r - 7) + 16) >> 5);
        }
        const qparams q = make_q(T, qp, true);
        int x[16], lev[16];
#pragma unroll
        for (int k = 0; k < 16; k++) x[k] = src[k] - pred[k];
        fdct4(x);
        sh_dc[(by >> 2) * 4 + (bx >> 2)] = x[0];
        bool nz = quant_dequant<1>(x, lev, q);
        store_levels(ctx->levels + (size_t)mbn * MB_LEVELS + L_LUMA + b * 16, lev);
        flags = nz ? 1 : 0;
        WAVE_SYNC();
        // lane p = raster position (i,j): hd = (M X M^T + 1) >> 1, M = [[1,1,1,1],[1,1,-1,-1],[1,-1,-1,1],[1,-1,1,-1]]
        const int pi = lane >> 2, pj = lane & 3;
        const int Mi[4] = {1, pi < 2 ? 1 : -1, (pi == 0 || pi == 3) ? 1 : -1, (pi & 1) ? -1 : 1};
        const int Mj[4] = {1, pj < 2 ? 1 : -1, (pj == 0 || pj == 3) ? 1 : -1, (pj & 1) ? -1 : 1};
        int acc = 0;
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int c2 = 0; c2 < 4; c2++) acc += Mi[a] * Mj[c2] * sh_dc[a * 4 + c2];
        const int hd = (acc + 1) >> 1;
        const int ldc = quant1(hd, q.mf[0], 2 * q.f, q.qbits + 1);
        sh_ldc[lane] = ldc;
        const int kz = (int)((0xFEA9DB83C7426510ull >> (4 * lane)) & 15); // zig-zag position of raster index `lane` (inverse of zz)
        stg16(&ctx->levels[(size_t)mbn * MB_LEVELS + L_LDC + kz], ldc);
        if (ldc) flags |= 2;
        WAVE_SYNC();
        { // inverse for this lane's own block position (by/4, bx/4): f = M c M^T, then 8.5.10 scaling
            const int bi = by >> 2, bj = bx >> 2;
            const int Ni[4] = {1, bi < 2 ? 1 : -1, (bi == 0 || bi == 3) ? 1 : -1, (bi & 1) ? -1 : 1};
            const int Nj[4] = {1, bj < 2 ? 1 : -1, (bj == 0 || bj == 3) ? 1 : -1, (bj & 1) ? -1 : 1};
            int f = 0;
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int c2 = 0; c2 < 4; c2++) f += Ni[a] * Nj[c2] * sh_ldc[a * 4 + c2];
            const int ls = 16 * q.v[0];
            x[0] = qp >= 36 ? (f * ls) << (qp / 6 - 6) : (f * ls + (1 << (5 - qp / 6))) >> (6 - qp / 6);
        }
        idct4(x);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const unsigned rw = pack4(clip255(pred[r * 4] + x[r * 4]), clip255(pred[r * 4 + 1] + x[r * 4 + 1]),
                                      clip255(pred[r * 4 + 2] + x[r * 4 + 2]), clip255(pred[r * 4 + 3] + x[r * 4 + 3]));
            stg32(ry + (size_t)(y0 + by + r) * stride + x0 + bx, rw);
            if (OUT) {
                if (by == 12 && r == 3) *(unsigned *)&L->bot_y[slot][bx] = rw;
                if (bx == 12) L->right_y[by + r] = (uint8_t)(rw >> 24);
            }
        }
    }
    if (is_chroma) { // wave 1, lanes 16-23: the 8 chroma blocks (a plane's four blocks in one DPP quad)
        const int cl = lane & 7, c = cl >> 2, b = cl & 3, bx = (b & 1) * 4, by = (b >> 1) * 4;
        const int p = 1 + c;
        if (cmode == 0) { // wave-uniform: only the chosen predictor is evaluated.  DC of this 4x4 block (8.3.4.1-3)
            int st = 0, sl = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) { st += TOP(p, bx + i); sl += LEFT(p, by + i); }
            bool ut = has_top, ul = has_left;
            if (b == 1 && has_top) ul = false;
            if (b == 2 && has_left) ut = false;
            const int dcv = (ut && ul) ? (st + sl + 4) >> 3 : ut ? (st + 2) >> 2 : ul ? (sl + 2) >> 2 : 128;
#pragma unroll
            for (int k = 0; k < 16; k++) pred[k] = dcv;
        } else if (cmode == 1) {
#pragma unroll
            for (int r = 0; r < 4; r++) { const int l = LEFT(p, by + r); pred[r * 4] = l; pred[r * 4 + 1] = l; pred[r * 4 + 2] = l; pred[r * 4 + 3] = l; }
        } else if (cmode == 2) {
#pragma unroll
            for (int i = 0; i < 4; i++) { const int t = TOP(p, bx + i); pred[i] = t; pred[4 + i] = t; pred[8 + i] = t; pred[12 + i] = t; }
        } else {
            int Hh = 0, Vv = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                Hh += (i + 1) * (TOP(p, 4 + i) - TOP(p, 2 - i));
                Vv += (i + 1) * (LEFT(p, 4 + i) - LEFT(p, 2 - i));
            }
            const int pa = 16 * (LEFT(p, 7) + TOP(p, 7)), pb = (34 * Hh + 32) >> 6, pc = (34 * Vv + 32) >> 6;
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int i = 0; i < 4; i++) pred[r * 4 + i] = clip255((pa + pb * (bx + i - 3) + pc * (by + r - 3) + 16) >> 5);
        }
        if (is_chroma) flags = chroma_block(ctx, T, mbn, cx0, cy0, cl, pred, qp, true, OUT ? L->crec : nullptr, presrc);
    }
#undef TOP
#undef LEFT
    const unsigned long long any = __ballot(flags & 1), dcm = __ballot(flags & 2);
    if (wave == 1) {
        if (OUT) {
            WAVE_SYNC();
            if (lane >= 16 && lane < 32) {
                const int i = lane - 16;
                L->bot_c[slot][i] = L->crec[7 * 16 + i];
                L->right_c[i >> 3][i & 7] = L->crec[(i & 7) * 16 + 14 + (i >> 3)];
            }
        }
        if (lane == 0) {
            L->cflags[0] = (unsigned)((any >> 16) & 0xFF); L->cflags[1] = (unsigned)((dcm >> 16) & 0xFF);
            __hip_atomic_store(&L->cseq, (unsigned)mbn + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    } else if (lane == 0) { // the luma wave writes the record once the chroma wave's flags are in (both waves are resident: plain spin)
        while (__hip_atomic_load(&L->cseq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != (unsigned)mbn + 1u) __builtin_amdgcn_s_sleep(1);
        const unsigned cany = L->cflags[0], cdc = L->cflags[1];
        unsigned nzm = (use_i4 ? nz4 : (unsigned)(any & 0xFFFF)) | (cany << 16);
        if (!use_i4 && (dcm & 0xFFFF)) nzm |= NZ_LDC;
        if (cdc & 0x0F) nzm |= NZ_CBDC;
        if (cdc & 0xF0) nzm |= NZ_CRDC;
        mb_info_t mb;
        mb.mvx = 0; mb.mvy = 0; mb.mb_type = use_i4 ? 2 : 0; mb.i16_mode = use_i4 ? 0 : (uint8_t)mode16; mb.chroma_mode = (uint8_t)cmode;
        mb.qp = (uint8_t)qp; mb.nzmask = nzm; mb.cost = dec1.y;
        st_mbinfo(&ctx->mbi[mbn], mb);
    }
}

// One launch per anti-diagonal x + y (replayed as a hipGraph): neighbours come from the reconstructed picture in global
// memory.  intra_mode 1; kept as the plain form and cross-check of the persistent kernel below.
__global__ __launch_bounds__(128) void intra_kernel(const frame_ctx_t *__restrict__ ctx, int diag) {
    __shared__ intra_lds LD;
    __shared__ unsigned tabw[TAB_DWORDS];
    const dev_tables *T = (const dev_tables *)tabw;
    const int mbw = ctx->mbw, stride = ctx->stride;
    const int y_lo = diag - (mbw - 1) > 0 ? diag - (mbw - 1) : 0;
    const int my = y_lo + blockIdx.x, mx = diag - my;
    const int mbn = my * mbw + mx, x0 = mx * 16, y0 = my * 16, cx0 = x0 >> 1, cy0 = y0 >> 1;
    const bool has_top = my > 0, has_left = mx > 0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint8_t *__restrict__ ry = ctx->rec_y;
    const uint8_t *__restrict__ ruv = ctx->rec_uv;
    for (int i = threadIdx.x; i < TAB_DWORDS; i += 128) tabw[i] = ((const unsigned *)&g_tab)[i];
    if (threadIdx.x == 0) LD.cseq = 0;
    const uint4 dec0 = ldg128(ctx->idec + (size_t)mbn * IDEC_BYTES);              // modes4[16]
    const uint2 dec1 = ldg64(ctx->idec + (size_t)mbn * IDEC_BYTES + 16);          // mode16, cmode, use_i4 | cost
    if (wave == 0 && lane >= 24 && lane < 24 + 17) { // wave 0, lanes 24-40: luma neighbours; index i+1 holds sample i, index 0 the corner
        int i = lane - 24 - 1;
        LD.top[0][i + 1] = has_top && (i >= 0 || has_left) ? (int)ldg8(ry + (size_t)(y0 - 1) * stride + x0 + i) : 0;
        LD.left[0][i + 1] = has_left && (i >= 0 || has_top) ? (int)ldg8(ry + (size_t)(y0 + i) * stride + x0 - 1) : 0;
    } else if (wave == 1 && lane >= 41 && lane < 41 + 18) { // wave 1, lanes 41-58: chroma neighbours
        int c = (lane - 41) / 9, i = (lane - 41) % 9 - 1;
        LD.top[1 + c][i + 1] = has_top && (i >= 0 || has_left) ? (int)ldg8(ruv + (size_t)(cy0 - 1) * stride + 2 * (cx0 + i) + c) : 0;
        LD.left[1 + c][i + 1] = has_left && (i >= 0 || has_top) ? (int)ldg8(ruv + (size_t)(cy0 + i) * stride + 2 * (cx0 - 1) + c) : 0;
    }
    __syncthreads();
    intra_compute<false>(ctx, T, &LD, mx, my, wave, lane, dec0, dec1);
}

// Persistent form of the intra wavefront (intra_mode 0): one launch per picture.  A workgroup owns a band of IB_ROWS
// macroblock rows, two waves per row (luma, chroma); all rows advance in lock-step, one barrier per step, row r handling
// macroblock x = t - r at step t (x + y order: left, top-left and top neighbours are complete, and Intra_4x4 never looks
// past its own macroblock's columns in the row above).  Neighbour samples never go through global memory inside a band:
// the bottom row of a macroblock travels to the row below through a 4-deep LDS ring, its right column stays in the row's
// own LDS for the next step.  Between bands the bottom rows of the last row are stored with `sc1` and announced through a
// progress counter, exactly like the deblocking bands; the first row of a band prefetches them one step ahead.
#define IB_ROWS 4
struct ib_args { frame_ctx_t ctx; unsigned *progress; unsigned *err; };

__global__ __launch_bounds__(IB_ROWS * 128) void intra_band_kernel(ib_args a) {
    __shared__ intra_lds LD[IB_ROWS];
    __shared__ unsigned tabw[TAB_DWORDS];
    __shared__ unsigned stage[2][8]; // first row of a band: the prefetched samples of the band above (luma, chroma), 5 dwords each
    const dev_tables *T = (const dev_tables *)tabw;
    const frame_ctx_t *__restrict__ ctx = &a.ctx;
    const int mbw = ctx->mbw, mbh = ctx->mbh, stride = ctx->stride;
    const int band = blockIdx.x, w = threadIdx.x >> 6, lane = threadIdx.x & 63, r = w >> 1, role = w & 1;
    const int my = band * IB_ROWS + r;
    const bool row_ok = my < mbh, has_top = my > 0;
    const bool fed = row_ok && r == 0 && band > 0;                        // top samples come from the band above
    const bool feeds = row_ok && r == IB_ROWS - 1 && my != mbh - 1;       // bottom rows go to the band below
    intra_lds *L = &LD[r];
    const intra_lds *Lup = &LD[r > 0 ? r - 1 : 0];
    for (int i = threadIdx.x; i < TAB_DWORDS; i += IB_ROWS * 128) tabw[i] = ((const unsigned *)&g_tab)[i];
    if (lane == 0 && role == 0) L->cseq = 0;
    const uint8_t *__restrict__ ry = ctx->rec_y;
    const uint8_t *__restrict__ ruv = ctx->rec_uv;
    unsigned *prog_up = a.progress + (band > 0 ? band - 1 : 0), *prog_my = a.progress + band;
    // the lanes that move neighbour samples: luma wave 24..40 (i = -1..15), chroma wave 41..58 (plane c, i = -1..7)
    const bool mover = role == 0 ? (lane >= 24 && lane < 41) : (lane >= 41 && lane < 59);
    const int mi = role == 0 ? lane - 25 : (lane - 41) % 9 - 1, mc = role == 0 ? 0 : (lane - 41) / 9;
    // ... and the lanes that fetch for a fed row: 5 dwords starting 4 bytes left of the macroblock (the corner is byte 3 of dword 0)
    const bool fetcher = fed && lane >= 24 && lane < 29;
    const uint8_t *frow = role == 0 ? ry + (size_t)(my * 16 - 1) * stride : ruv + (size_t)(my * 8 - 1) * stride;
    unsigned gpre = 0;
    int avail = 0;
    uint4 dec0n = make_uint4(0, 0, 0, 0);
    uint2 dec1n = make_uint2(0, 0);
    uint2 srcn[4] = {make_uint2(0, 0), make_uint2(0, 0), make_uint2(0, 0), make_uint2(0, 0)}; // this lane's source rows of the next macroblock
    const bool src_luma = role == 0 && lane < 16, src_chroma = role == 1 && lane >= 16 && lane < 24;
    const int nsteps = mbw + IB_ROWS + 1;
#ifdef IB_PROF /* debug builds: cycles inside intra_compute per wave, and of the whole loop, left in ctx->isad */
    unsigned long long ib_cyc = 0, ib_n = 0;
    const unsigned long long ib_l0 = __builtin_readcyclecounter();
#endif
    for (int t = -1; t < nsteps; t++) { // step -1 only prefetches for the first row
        const int x = t - r, xn = x + 1;
        const bool act = row_ok && x >= 0 && x < mbw;
        const bool pf = row_ok && xn >= 0 && xn < mbw;
        if (feeds) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // both waves: the bottom rows stored in the previous step have landed
        __syncthreads(); // the rings and right columns written in the previous step are visible
        if (feeds && role == 0 && lane == 0 && x >= 1 && x <= mbw) st_sc1(prog_my, (unsigned)x); // ... announce them
        // ---- land what was prefetched for this step, prefetch for the next one
        const uint4 dec0 = dec0n;
        const uint2 dec1 = dec1n;
        const uint2 srcc[4] = {srcn[0], srcn[1], srcn[2], srcn[3]};
        if (fed && act && lane >= 24 && lane < 29) stage[role][lane - 24] = gpre;
        if (pf) {
            const size_t mbn_n = (size_t)my * mbw + xn;
            dec0n = ldg128(ctx->idec + mbn_n * IDEC_BYTES);
            dec1n = ldg64(ctx->idec + mbn_n * IDEC_BYTES + 16);
            if (src_luma) {
                const int bx = blkx(lane), by = blky(lane), vh = ctx->vis_h;
#pragma unroll
                for (int q = 0; q < 4; q++) { int sy = my * 16 + by + q; sy = sy < vh ? sy : vh - 1; srcn[q].x = ldg32(ctx->src_y + (size_t)sy * ctx->src_stride + xn * 16 + bx); }
            } else if (src_chroma) {
                const int cl = lane & 7, b = cl & 3, bx = (b & 1) * 4, by = (b >> 1) * 4, vh2 = ctx->vis_h >> 1;
#pragma unroll
                for (int q = 0; q < 4; q++) { int sy = my * 8 + by + q; sy = sy < vh2 ? sy : vh2 - 1; srcn[q] = ldg64(ctx->src_uv + (size_t)sy * ctx->src_stride + 2 * (xn * 8 + bx)); }
            }
            if (fed) {
                if (avail < xn + 1) avail = db_wait_get(prog_up, a.err, xn + 1);
                if (fetcher) gpre = xn > 0 || lane > 24 ? ld_sc1((const unsigned *)(frow + xn * 16 - 4 + 4 * (lane - 24))) : 0u;
            }
        }
        WAVE_SYNC();
        if (act) {
            // ---- neighbours of macroblock x into L->top / L->left
            if (mover) {
                const bool has_left = x > 0;
                int tv = 0, lv = 0;
                if (role == 0) {
                    if (has_top && (mi >= 0 || has_left))
                        tv = fed ? (int)((const uint8_t *)stage[0])[4 + mi] : (mi >= 0 ? (int)Lup->bot_y[x & 3][mi] : (int)Lup->bot_y[(x - 1) & 3][15]);
                    if (has_left && (mi >= 0 || has_top)) lv = mi >= 0 ? (int)L->right_y[mi] : tv;
                } else {
                    if (has_top && (mi >= 0 || has_left))
                        tv = fed ? (int)((const uint8_t *)stage[1])[4 + 2 * mi + mc] : (mi >= 0 ? (int)Lup->bot_c[x & 3][2 * mi + mc] : (int)Lup->bot_c[(x - 1) & 3][14 + mc]);
                    if (has_left && (mi >= 0 || has_top)) lv = mi >= 0 ? (int)L->right_c[mc][mi] : tv;
                }
                L->top[role ? 1 + mc : 0][mi + 1] = tv;
                L->left[role ? 1 + mc : 0][mi + 1] = lv;
            }
            WAVE_SYNC();
#ifdef IB_PROF
            const unsigned long long ib_t0 = __builtin_readcyclecounter();
#endif
            intra_compute<true>(ctx, T, L, x, my, role, lane, dec0, dec1, srcc);
#ifdef IB_PROF
            ib_cyc += __builtin_readcyclecounter() - ib_t0; ib_n++;
#endif
            // ---- last row of the band: its bottom rows go to the band below
            if (feeds) {
                WAVE_SYNC();
                if (lane < 4) {
                    if (role == 0) st_sc1((unsigned *)(ctx->rec_y + (size_t)(my * 16 + 15) * stride + x * 16) + lane, ((const unsigned *)L->bot_y[x & 3])[lane]);
                    else st_sc1((unsigned *)(ctx->rec_uv + (size_t)(my * 8 + 7) * stride + x * 16) + lane, ((const unsigned *)L->bot_c[x & 3])[lane]);
                }
            }
        }
    }
#ifdef IB_PROF
    if (lane == 0 && band < 2) {
        unsigned *o = (unsigned *)ctx->isad + (band * 8 + w) * 4;
        o[0] = (unsigned)ib_cyc; o[1] = (unsigned)ib_n; o[2] = (unsigned)(__builtin_readcyclecounter() - ib_l0); o[3] = (unsigned)nsteps;
    }
#endif
}
int k_intra_bands(int mbh) { return (mbh + IB_ROWS - 1) / IB_ROWS; }
// d_progress: one counter per band (cleared here), then the sticky error word
void k_launch_intra_band(const frame_ctx_t *h_ctx, int mbh, unsigned *d_progress, unsigned *d_err, hipStream_t s) {
    ib_args a;
    a.ctx = *h_ctx; a.progress = d_progress; a.err = d_err;
    (void)hipMemsetAsync(d_progress, 0, (size_t)k_intra_bands(mbh) * sizeof(unsigned), s);
    hipLaunchKernelGGL(intra_band_kernel, dim3(k_intra_bands(mbh)), dim3(IB_ROWS * 128), 0, s, a);
}

// =================================================================== deblocking (8.7)
DEV void filter_line(const dev_tables *T, uint8_t *pix, int step, int bS, int qp_p, int qp_q, bool chroma) {
    if (bS == 0) return;
    const int idx = clip3(0, 51, (qp_p + qp_q + 1) >> 1);
    const int alpha = T->alpha[idx], beta = T->beta[idx];
    const int p0 = pix[-step], p1 = pix[-2 * step], q0 = pix[0], q1 = pix[step];
    if (!(iabs(p0 - q0) < alpha && iabs(p1 - p0) < beta && iabs(q1 - q0) < beta)) return;
    if (chroma) {
        if (bS < 4) {
            const int tc = T->tc0[idx][bS - 1] + 1;
            const int dl = clip3(-tc, tc, (((q0 - p0) << 2) + (p1 - q1) + 4) >> 3);
            pix[-step] = (uint8_t)clip255(p0 + dl); pix[0] = (uint8_t)clip255(q0 - dl);
        } else {
            pix[-step] = (uint8_t)((2 * p1 + p0 + q1 + 2) >> 2); pix[0] = (uint8_t)((2 * q1 + q0 + p1 + 2) >> 2);
        }
        return;
    }
    const int p2 = pix[-3 * step], q2 = pix[2 * step];
    const bool ap = iabs(p2 - p0) < beta, aq = iabs(q2 - q0) < beta;
    if (bS < 4) {
        const int tc0 = T->tc0[idx][bS - 1];
        const int tc = tc0 + (ap ? 1 : 0) + (aq ? 1 : 0);
        const int dl = clip3(-tc, tc, (((q0 - p0) << 2) + (p1 - q1) + 4) >> 3);
        pix[-step] = (uint8_t)clip255(p0 + dl); pix[0] = (uint8_t)clip255(q0 - dl);
        const int avg = (p0 + q0 + 1) >> 1;
        if (ap) pix[-2 * step] = (uint8_t)(p1 + clip3(-tc0, tc0, (p2 + avg - (p1 << 1)) >> 1));
        if (aq) pix[step] = (uint8_t)(q1 + clip3(-tc0, tc0, (q2 + avg - (q1 << 1)) >> 1));
    } else {
        const int p3 = pix[-4 * step], q3 = pix[3 * step];
        const bool small = iabs(p0 - q0) < ((alpha >> 2) + 2);
        if (ap && small) {
            pix[-step] = (uint8_t)((p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3);
            pix[-2 * step] = (uint8_t)((p2 + p1 + p0 + q0 + 2) >> 2);
            pix[-3 * step] = (uint8_t)((2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3);
        } else pix[-step] = (uint8_t)((2 * p1 + p0 + q1 + 2) >> 2);
        if (aq && small) {
            pix[0] = (uint8_t)((p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3);
            pix[step] = (uint8_t)((p0 + q0 + q1 + q2 + 2) >> 2);
            pix[2 * step] = (uint8_t)((2 * q3 + 3 * q2 + q1 + q0 + p0 + 4) >> 3);
        } else pix[0] = (uint8_t)((2 * q1 + q0 + p1 + 2) >> 2);
    }
}
DEV int has_coef(const mb_info_t &m, int bx4, int by4) { // (bx4,by4) raster 4x4 position -> blkIdx bit
    const int b = ((by4 >> 1) << 3) | ((bx4 >> 1) << 2) | ((by4 & 1) << 1) | (bx4 & 1);
    if (m.nzmask & NZ_T8) return ((m.nzmask >> (b & ~3)) & 0xF) != 0; // 8.7.2.1: the 8x8 block containing the sample
    return (m.nzmask >> b) & 1;
}
DEV int bs_of(const mb_info_t &mp, int bxp, int byp, const mb_info_t &mq, int bxq, int byq, bool mb_edge) {
    if (mp.mb_type != 1 || mq.mb_type != 1) return mb_edge ? 4 : 3; // 0 (I16x16) and 2 (I4x4) are intra
    if (has_coef(mp, bxp, byp) || has_coef(mq, bxq, byq)) return 2;
    if (iabs(mp.mvx - mq.mvx) >= 4 || iabs(mp.mvy - mq.mvy) >= 4) return 1; // quarter-sample units
    return 0;
}
// One wave per macroblock, launched once per wavefront x + 2y = diag: then the left, top and
// top-right macroblocks (everything the raster-order process of 8.7 has touched before this
// macroblock that overlaps its support) are complete, and same-diagonal tiles are disjoint.
#define TLS 24 /* LDS tile row stride in bytes */
__global__ __launch_bounds__(64) void deblock_kernel(const frame_ctx_t *__restrict__ ctx, int diag) {
    __shared__ __attribute__((aligned(16))) uint8_t tl[20 * TLS]; // luma rows y0-4..y0+15, cols x0-4..x0+15
    __shared__ __attribute__((aligned(16))) uint8_t tc[10 * TLS]; // chroma rows cy0-2..cy0+7, bytes 2*(cx0-2)..2*(cx0+8)
    __shared__ unsigned tabw[TAB_DWORDS];
    const dev_tables *T = (const dev_tables *)tabw;
    const int mbw = ctx->mbw, mbh = ctx->mbh, stride = ctx->stride;
    int y_lo = diag - (mbw - 1);
    y_lo = y_lo > 0 ? (y_lo + 1) >> 1 : 0;
    const int my = y_lo + blockIdx.x, mx = diag - 2 * my;
    if (my >= mbh || mx < 0 || mx >= mbw) return;
    const int x0 = mx * 16, y0 = my * 16, cy0 = y0 >> 1;
    const int lane = threadIdx.x;
    uint8_t *__restrict__ ry = ctx->rec_y;
    uint8_t *__restrict__ ruv = ctx->rec_uv;
    const mb_info_t cur = ld_mbinfo(&ctx->mbi[my * mbw + mx]);
    const mb_info_t lft = ld_mbinfo(&ctx->mbi[my * mbw + (mx > 0 ? mx - 1 : mx)]);
    const mb_info_t upp = ld_mbinfo(&ctx->mbi[(my > 0 ? my - 1 : my) * mbw + mx]);
    // ---- every global load of this macroblock is issued here, before the first wait
    for (int i = lane; i < TAB_DWORDS; i += 64) tabw[i] = ((const unsigned *)&g_tab)[i];
    // ---- load tiles (skipping the corner, which this macroblock neither reads nor writes)
    for (int i = lane; i < 100; i += 64) {
        int r = i / 5, q = i - r * 5;
        int gy = y0 - 4 + r, gx = x0 - 4 + 4 * q;
        if (gy >= 0 && gx >= 0 && !(r < 4 && q == 0))
            *(unsigned *)&tl[r * TLS + 4 * q] = ldg32(ry + (size_t)gy * stride + gx);
    }
    if (lane < 50) {
        int r = lane / 5, q = lane - r * 5;
        int gy = cy0 - 2 + r, gb = x0 - 4 + 4 * q; // chroma byte offset 2*cx0 = x0
        if (gy >= 0 && gb >= 0 && !(r < 2 && q == 0))
            *(unsigned *)&tc[r * TLS + 4 * q] = ldg32(ruv + (size_t)gy * stride + gb);
    }
    __syncthreads();
    const int qpc_c = T->qpc[cur.qp], qpc_l = T->qpc[lft.qp], qpc_u = T->qpc[upp.qp];
    // ---- vertical edges, left to right
    if (lane < 16) {
        const int k = lane;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            if ((e == 0 && mx == 0) || ((cur.nzmask & NZ_T8) && (e & 1))) continue; // 8x8 transform: edges 1, 3 are not block edges
            const mb_info_t &mp = e == 0 ? lft : cur;
            int bS = bs_of(mp, e == 0 ? 3 : e - 1, k >> 2, cur, e, k >> 2, e == 0);
            filter_line(T, &tl[(4 + k) * TLS + 4 + 4 * e], 1, bS, mp.qp, cur.qp, false);
        }
    } else if (lane < 24) {
        const int k = lane - 16;
#pragma unroll
        for (int e = 0; e < 4; e += 2) {
            if (e == 0 && mx == 0) continue;
            const mb_info_t &mp = e == 0 ? lft : cur;
            int bS = bs_of(mp, e == 0 ? 3 : e - 1, k >> 1, cur, e, k >> 1, e == 0);
#pragma unroll
            for (int c = 0; c < 2; c++) filter_line(T, &tc[(2 + k) * TLS + 4 + 4 * e + c], 2, bS, e == 0 ? qpc_l : qpc_c, qpc_c, true);
        }
    }
    __syncthreads();
    // ---- horizontal edges, top to bottom
    if (lane < 16) {
        const int k = lane;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            if ((e == 0 && my == 0) || ((cur.nzmask & NZ_T8) && (e & 1))) continue;
            const mb_info_t &mp = e == 0 ? upp : cur;
            int bS = bs_of(mp, k >> 2, e == 0 ? 3 : e - 1, cur, k >> 2, e, e == 0);
            filter_line(T, &tl[(4 + 4 * e) * TLS + 4 + k], TLS, bS, mp.qp, cur.qp, false);
        }
    } else if (lane < 24) {
        const int k = lane - 16;
#pragma unroll
        for (int e = 0; e < 4; e += 2) {
            if (e == 0 && my == 0) continue;
            const mb_info_t &mp = e == 0 ? upp : cur;
            int bS = bs_of(mp, k >> 1, e == 0 ? 3 : e - 1, cur, k >> 1, e, e == 0);
#pragma unroll
            for (int c = 0; c < 2; c++) filter_line(T, &tc[(2 + 2 * e) * TLS + 4 + 2 * k + c], TLS, bS, e == 0 ? qpc_u : qpc_c, qpc_c, true);
        }
    }
    __syncthreads();
    // ---- write back
    for (int i = lane; i < 100; i += 64) {
        int r = i / 5, q = i - r * 5;
        int gy = y0 - 4 + r, gx = x0 - 4 + 4 * q;
        if (gy >= 0 && gx >= 0 && !(r < 4 && q == 0))
            stg32(ry + (size_t)gy * stride + gx, *(const unsigned *)&tl[r * TLS + 4 * q]);
    }
    if (lane < 50) {
        int r = lane / 5, q = lane - r * 5;
        int gy = cy0 - 2 + r, gb = x0 - 4 + 4 * q;
        if (gy >= 0 && gb >= 0 && !(r < 2 && q == 0))
            stg32(ruv + (size_t)gy * stride + gb, *(const unsigned *)&tc[r * TLS + 4 * q]);
    }
}

// ------------------------------------------------------------------ shared by the persistent band kernel
struct edge_par { int alpha, beta; unsigned tc0; }; // tc0: three bytes, bS 1..3 (kept packed: an indexable array would live in scratch)
struct db_args { frame_ctx_t ctx; unsigned *progress; unsigned *err; int band0, nb_total; }; // a launch covers bands band0 .. band0 + gridDim.x/2 - 1

// =================================================================== deblocking, persistent: 16-row bands in x + y order
// One launch per picture instead of one per wavefront.  Two observations shorten the dependency chain:
//  (1) Boundary strengths and the alpha/beta/tc0 triples depend only on the macroblock records,
//      so a flat kernel (deblock_prep_kernel) computes them for the whole picture up front:
//      64 bytes per macroblock {bS nibbles V/H, six packed parameter pairs}.
//  (2) x + 2y is sufficient but not necessary.  Macroblock (x, y) only conflicts with its
//      top-right neighbour (x+1, y-1) on the 3x3 corner of (x, y-1) that the neighbour's left
//      edge (a VERTICAL edge, first thing it filters) and this macroblock's top edge (a
//      HORIZONTAL edge, filtered after all vertical ones) both touch.  If every row filters its
//      vertical edges, all rows meet at one barrier, and then every row filters its horizontal
//      edges, (x, y) and (x+1, y-1) can share a step: the order x + y with ONE barrier per step
//      reproduces the raster-order result (mbw + mbh - 1 steps instead of mbw + 2(mbh - 1)).
// A wave serves four macroblock rows (16 lanes each: one lane per picture line / column), so a
// workgroup of 4 luma + 4 chroma waves owns a band of 16 rows and only every 16th row boundary
// crosses global memory: the bottom strip of a band's last row is stored with `sc1` (agent scope,
// L1-bypassing) stores, `s_waitcnt vmcnt(0)`, then an sc1 store of a monotonic progress counter per band and
// plane; the band below polls the counter with sc1 loads and reads the strip with sc1 loads
// (MI355X_MICROARCH.md, "Valid forms").  A band waits only on the band above it, so the wait graph is
// acyclic; every spin is bounded and reports through `err`.  Each lane
// group prefetches its next macroblock one step ahead into registers and lands it in LDS after
// the step's arithmetic, immediately before the step's own stores are issued, so the only
// `s_waitcnt vmcnt(0)` on the chain waits for loads that have had a whole step to arrive.
#define D3_ROWS 16
#define D3_TS 52 /* tile row stride (13 dwords: the 16 lines of a group fall on 16 different banks) */
struct d3_luma { uint8_t t[20 * D3_TS]; unsigned ring[4][16]; unsigned rec[16]; };   // 1360 B = 340 dwords (20 mod 32)
struct d3_chroma { uint8_t t[10 * D3_TS]; unsigned ring[4][8]; unsigned rec[16]; };  // 712 B = 178 dwords (18 mod 32)
#define DBREC_BYTES 64

DEV unsigned pack_par_ab(const dev_tables *T, int idx) { return (unsigned)T->alpha[idx] | ((unsigned)T->beta[idx] << 8); }
DEV unsigned pack_par_tc(const dev_tables *T, int idx) { return (unsigned)T->tc0[idx][0] | ((unsigned)T->tc0[idx][1] << 8) | ((unsigned)T->tc0[idx][2] << 16); }
DEV edge_par par_of(unsigned ab, unsigned tc) { edge_par p; p.alpha = (int)(ab & 255); p.beta = (int)(ab >> 8); p.tc0 = tc; return p; }

// One thread per macroblock: words 0-1 bS of the vertical edges (nibble 4*edge + segment), 2-3 of
// the horizontal edges, then {alpha|beta<<8, tc0 bytes} for luma left / top / inner and chroma
// left / top / inner.  Edges that are not filtered (picture border, 8x8-transform inner edges) get bS 0.
// Also clears the band progress counters of the launch that follows.
__global__ __launch_bounds__(256) void deblock_prep_kernel(const frame_ctx_t cv, unsigned *__restrict__ progress, int nprog) {
    const frame_ctx_t *__restrict__ ctx = &cv;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < nprog) progress[i] = 0;
    const int mbw = ctx->mbw, mbh = ctx->mbh;
    if (i >= mbw * mbh) return;
    const dev_tables *T = &g_tab;
    const int my = i / mbw, mx = i - my * mbw;
    const mb_info_t cur = ld_mbinfo(&ctx->mbi[i]);
    const mb_info_t lft = ld_mbinfo(&ctx->mbi[mx > 0 ? i - 1 : i]);
    const mb_info_t upp = ld_mbinfo(&ctx->mbi[my > 0 ? i - mbw : i]);
    const bool t8 = (cur.nzmask & NZ_T8) != 0;
    unsigned w[16];
    unsigned long long bv = 0, bh = 0;
#pragma unroll
    for (int e = 0; e < 4; e++)
#pragma unroll
        for (int sg = 0; sg < 4; sg++) {
            int v = 0, h = 0;
            if (!(t8 && (e & 1))) {
                if (!(e == 0 && mx == 0)) v = bs_of(e == 0 ? lft : cur, e == 0 ? 3 : e - 1, sg, cur, e, sg, e == 0);
                if (!(e == 0 && my == 0)) h = bs_of(e == 0 ? upp : cur, sg, e == 0 ? 3 : e - 1, cur, sg, e, e == 0);
            }
            bv |= (unsigned long long)v << (4 * (e * 4 + sg));
            bh |= (unsigned long long)h << (4 * (e * 4 + sg));
        }
    w[0] = (unsigned)bv; w[1] = (unsigned)(bv >> 32); w[2] = (unsigned)bh; w[3] = (unsigned)(bh >> 32);
    const int il = clip3(0, 51, (lft.qp + cur.qp + 1) >> 1), it = clip3(0, 51, (upp.qp + cur.qp + 1) >> 1), ii = cur.qp;
    const int qc = T->qpc[cur.qp], cl = (T->qpc[lft.qp] + qc + 1) >> 1, ct = (T->qpc[upp.qp] + qc + 1) >> 1;
    w[4] = pack_par_ab(T, il); w[5] = pack_par_tc(T, il); w[6] = pack_par_ab(T, it); w[7] = pack_par_tc(T, it);
    w[8] = pack_par_ab(T, ii); w[9] = pack_par_tc(T, ii); w[10] = pack_par_ab(T, cl); w[11] = pack_par_tc(T, cl);
    w[12] = pack_par_ab(T, ct); w[13] = pack_par_tc(T, ct); w[14] = pack_par_ab(T, qc); w[15] = pack_par_tc(T, qc);
    uint8_t *o = ctx->dbrec + (size_t)i * DBREC_BYTES;
#pragma unroll
    for (int q = 0; q < 4; q++) stg128(o + 16 * q, make_uint4(w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]));
}

typedef v4u v4u_a4 __attribute__((aligned(4)));
DEV void stg128u(void *p, unsigned a, unsigned b, unsigned c, unsigned d) { v4u t; t.x = a; t.y = b; t.z = c; t.w = d; *(GAS v4u_a4 *)p = t; } // 4-byte aligned

// Branch-free forms of the edge filters (8.7.2.3 / 8.7.2.4): every lane computes both candidates and
// selects, so a step costs the same few dozen VALU instructions whatever the lanes decide -- the
// early-outs of edge_luma() only pay when a whole wave agrees, which the uniform `any4` / ballot
// tests outside keep.  Samples are 0..255, so |a - b| is one v_sad_u8.
DEV int adiff(int a, int b) { return (int)__builtin_amdgcn_sad_u8((unsigned)a, (unsigned)b, 0u); }
template <bool MBEDGE>
DEV void edge_luma2(const edge_par &P, int p3, int &p2, int &p1, int &p0, int &q0, int &q1, int &q2, int q3, int bS, bool any4) {
    const int alpha = P.alpha, beta = P.beta;
    const int d = adiff(p0, q0);
    const bool f = (bS != 0) & (d < alpha) & (adiff(p1, p0) < beta) & (adiff(q1, q0) < beta);
    const bool ap = adiff(p2, p0) < beta, aq = adiff(q2, q0) < beta;
    const int tc0 = (int)((P.tc0 >> (8 * ((bS - 1) & 3))) & 0xFF); // bS 0 or 4 read a don't-care byte
    const int tc = tc0 + (ap ? 1 : 0) + (aq ? 1 : 0);
    const int dl = clip3(-tc, tc, (((q0 - p0) << 2) + (p1 - q1) + 4) >> 3);
    const int avg = (p0 + q0 + 1) >> 1;
    int np1 = p1 + clip3(-tc0, tc0, (p2 + avg - (p1 << 1)) >> 1);
    int nq1 = q1 + clip3(-tc0, tc0, (q2 + avg - (q1 << 1)) >> 1);
    int np0 = clip255(p0 + dl), nq0 = clip255(q0 - dl);
    int np2 = p2, nq2 = q2;
    bool wp1 = ap, wq1 = aq;
    if (MBEDGE && any4) { // bS 4 exists only on macroblock edges, and only if some lane of the wave is intra
        const bool s4 = bS == 4, small = d < ((alpha >> 2) + 2);
        const bool sp = ap & small, sq = aq & small;
        const int sp0 = sp ? (p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3 : (2 * p1 + p0 + q1 + 2) >> 2;
        const int sq0 = sq ? (p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3 : (2 * q1 + q0 + p1 + 2) >> 2;
        const int sp1 = (p2 + p1 + p0 + q0 + 2) >> 2, sp2 = (2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3;
        const int sq1 = (p0 + q0 + q1 + q2 + 2) >> 2, sq2 = (2 * q3 + 3 * q2 + q1 + q0 + p0 + 4) >> 3;
        np0 = s4 ? sp0 : np0; nq0 = s4 ? sq0 : nq0; np1 = s4 ? sp1 : np1; nq1 = s4 ? sq1 : nq1;
        wp1 = s4 ? sp : ap; wq1 = s4 ? sq : aq;
        np2 = (s4 & sp) ? sp2 : p2; nq2 = (s4 & sq) ? sq2 : q2;
    }
    p0 = f ? np0 : p0; q0 = f ? nq0 : q0;
    p1 = (f & wp1) ? np1 : p1; q1 = (f & wq1) ? nq1 : q1;
    p2 = f ? np2 : p2; q2 = f ? nq2 : q2;
}
DEV void edge_chroma2(const edge_par &P, int p1, int &p0, int &q0, int q1, int bS) {
    const bool f = (bS != 0) & (adiff(p0, q0) < P.alpha) & (adiff(p1, p0) < P.beta) & (adiff(q1, q0) < P.beta);
    const int tc = (int)((P.tc0 >> (8 * ((bS - 1) & 3))) & 0xFF) + 1;
    const int dl = clip3(-tc, tc, (((q0 - p0) << 2) + (p1 - q1) + 4) >> 3);
    const bool s4 = bS == 4;
    const int np0 = s4 ? (2 * p1 + p0 + q1 + 2) >> 2 : clip255(p0 + dl);
    const int nq0 = s4 ? (2 * q1 + q0 + p1 + 2) >> 2 : clip255(q0 - dl);
    p0 = f ? np0 : p0; q0 = f ? nq0 : q0;
}

// One workgroup = one band of 16 macroblock rows of ONE plane (blocks [0, nb): luma, [nb, 2nb):
// chroma -- the planes share nothing but the records, and on separate CUs neither steals issue
// slots from the other's dependency chain).  4 waves, one per SIMD; a wave serves four rows, 16
// lanes each.
template <bool CHROMA>
DEV void band16_body(const db_args &a, const int band, const int nb, uint8_t *lds) {
    constexpr int rows_mb = CHROMA ? 8 : 16, strip = CHROMA ? 2 : 4, ring_n = CHROMA ? 8 : 16;
    constexpr int ROW_LDS = CHROMA ? (int)sizeof(d3_chroma) : (int)sizeof(d3_luma);
    const frame_ctx_t *__restrict__ ctx = &a.ctx;
    const int mbw = ctx->mbw, mbh = ctx->mbh, stride = ctx->stride;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int g = lane >> 4, k = lane & 15, r = 4 * wave + g, my = band * D3_ROWS + r;
    const bool row_ok = my < mbh, last_row = my == mbh - 1;
    const bool fed = row_ok && r == 0 && band > 0;
    const bool feeds = row_ok && r == D3_ROWS - 1 && !last_row;
    unsigned *prog_up = a.progress + (CHROMA ? nb : 0) + (band > 0 ? band - 1 : 0), *prog_my = a.progress + (CHROMA ? nb : 0) + band;
    uint8_t *__restrict__ plane = CHROMA ? ctx->rec_uv : ctx->rec_y;
    const uint8_t *__restrict__ dbrec = ctx->dbrec;
    const size_t row0 = (size_t)my * rows_mb;
    uint8_t *tile = lds + r * ROW_LDS;                                     // d3_luma / d3_chroma of this row: t, ring, rec
    unsigned *ring = (unsigned *)(tile + (CHROMA ? 10 : 20) * D3_TS);
    unsigned *recw = ring + 4 * ring_n;
    const unsigned *ring_up = (const unsigned *)(lds + (r > 0 ? r - 1 : 0) * ROW_LDS + (CHROMA ? 10 : 20) * D3_TS);
    const int keep = last_row ? rows_mb : rows_mb - strip; // rows stored by this row itself; the strip below goes through the ring
    uint4 own = make_uint4(0, 0, 0, 0), recv = make_uint4(0, 0, 0, 0);
    unsigned stripv = 0, strip_next = 0;
    int avail = 0, avail_next = 0;
    const int nsteps = mbw + D3_ROWS + 2;
#if defined(D3_PROF) && D3_PROF == 1 /* debug builds only: per-phase cycle counters (perturbs: every tick drains lgkmcnt) */
    unsigned long long pc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tm0, tm1;
#define D3_TICK(i) do { tm1 = __builtin_readcyclecounter(); pc[i] += tm1 - tm0; tm0 = tm1; } while (0)
#else
#define D3_TICK(i) do { } while (0)
#endif
#if defined(D3_PROF) && D3_PROF == 2 /* whole loop only */
    unsigned long long pc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long loop_t0 = __builtin_readcyclecounter();
#endif
    for (int t = 0; t < nsteps; t++) {
        const int x = t - 1 - r, xn = x + 1;
        const bool act = row_ok && x >= 0 && x < mbw;
        const bool pf = row_ok && xn >= 0 && xn < mbw;
        const bool pub = feeds && x >= 1 && x <= mbw;         // strip of macroblock x-1 becomes final in this step's vertical phase
        const int x0b = x * 16;
#if defined(D3_PROF) && D3_PROF == 1
        tm0 = __builtin_readcyclecounter();
#endif
        // ---- A. prefetch macroblock x+1 (rows, record, strip of the band above)
        if (pf) {
            if (fed) {
                if (avail < xn + 1) avail = db_wait_get(prog_up, a.err, xn + 1);
                if (k < strip * 4) strip_next = ld_sc1((const unsigned *)(plane + (row0 - strip + (k >> 2)) * stride + xn * 16 + 4 * (k & 3)));
                avail_next = (int)ld_sc1(prog_up);
            }
            if (k < rows_mb) own = ldg128(plane + (row0 + k) * stride + xn * 16);
            if (k >= 12) recv = ldg128(dbrec + ((size_t)my * mbw + xn) * DBREC_BYTES + 16 * (k - 12));
        }
        D3_TICK(0);
        // ---- B. vertical edges.  All LDS reads of the phase are issued together (one round trip).
        const unsigned bvl = act ? recw[0] : 0u, bvh = act ? recw[1] : 0u, bhl = act ? recw[2] : 0u, bhh = act ? recw[3] : 0u;
        constexpr int o = CHROMA ? 10 : 4;
        const edge_par PL = par_of(recw[o], recw[o + 1]), PT = par_of(recw[o + 2], recw[o + 3]), PI = par_of(recw[o + 4], recw[o + 5]);
        if (!CHROMA) {
            unsigned w5[5];
#pragma unroll
            for (int i = 0; i < 5; i++) w5[i] = *(const unsigned *)&tile[(k + 4) * D3_TS + 12 + 4 * i];
            if (__ballot((bvl | bvh) != 0)) {
                const int sh = 4 * (k >> 2);
                int px[20];
#pragma unroll
                for (int i = 0; i < 20; i++) px[i] = byte_of(w5[i >> 2], i & 3);
                {
                    const int bS = (int)((bvl >> sh) & 15);
                    const unsigned long long nz = __ballot(bS != 0);
                    if (nz) edge_luma2<true>(PL, px[0], px[1], px[2], px[3], px[4], px[5], px[6], px[7], bS, __ballot(bS == 4) != 0);
                }
#pragma unroll
                for (int e = 1; e < 4; e++) {
                    const int bS = (int)(((e < 2 ? bvl : bvh) >> (16 * (e & 1) + sh)) & 15);
                    if (__ballot(bS != 0)) edge_luma2<false>(PI, px[4 * e], px[4 * e + 1], px[4 * e + 2], px[4 * e + 3], px[4 * e + 4], px[4 * e + 5], px[4 * e + 6], px[4 * e + 7], bS, false);
                }
                if (act) {
                    const unsigned l0 = pack4(px[0], px[1], px[2], px[3]);
                    *(unsigned *)&tile[(k + 4) * D3_TS + 12] = l0;
#pragma unroll
                    for (int i = 1; i < 5; i++) *(unsigned *)&tile[(k + 4) * D3_TS + 12 + 4 * i] = pack4(px[4 * i], px[4 * i + 1], px[4 * i + 2], px[4 * i + 3]);
                    if (k >= 12 && x > 0 && !last_row) ring[((x - 1) & 3) * 16 + (k - 12) * 4 + 3] = l0; // columns 12..15 of the previous macroblock's strip
                }
            }
        } else {
            const int kk = k & 7, c = k >> 3, sh = 4 * (kk >> 1);
            uint8_t *b = &tile[(kk + 2) * D3_TS + 12 + c]; // samples of plane c sit 2 bytes apart; q0 of edge e at byte 4 + 4e
            int s[8];
#pragma unroll
            for (int i = 0; i < 8; i++) s[i] = b[2 * i];
            if (__ballot((bvl | bvh) != 0)) {
                edge_chroma2(PL, s[0], s[1], s[2], s[3], (int)((bvl >> sh) & 15));
                edge_chroma2(PI, s[4], s[5], s[6], s[7], (int)((bvh >> sh) & 15));
                if (act) {
                    b[2] = (uint8_t)s[1]; b[4] = (uint8_t)s[2]; b[10] = (uint8_t)s[5]; b[12] = (uint8_t)s[6];
                    WAVE_SYNC();
                    if (k >= 6 && k < 8 && x > 0 && !last_row) ring[((x - 1) & 3) * 8 + (k - 6) * 4 + 3] = *(const unsigned *)&tile[(k + 2) * D3_TS + 12];
                }
            }
        }
        // ---- the strip of macroblock x-1 is final now: hand it to the band below
        if (pub) {
            WAVE_SYNC();
            if (k < strip * 4)
                st_sc1((unsigned *)(plane + (row0 + rows_mb - strip + (k >> 2)) * stride + (x - 1) * 16 + 4 * (k & 3)), ring[((x - 1) & 3) * ring_n + k]);
        }
        D3_TICK(1);
        // ---- C. the one barrier of the step: every vertical edge of this step precedes every horizontal edge
        BAND_BARRIER();
        D3_TICK(2);
        unsigned sa0 = 0, sa1 = 0, sa2 = 0, sa3 = 0, sb = 0;
        uint4 sc = make_uint4(0, 0, 0, 0);
        // ---- D. horizontal edges
        if (act && my > 0 && k < strip * 4) *(unsigned *)&tile[(k >> 2) * D3_TS + 16 + 4 * (k & 3)] = fed ? stripv : ring_up[(x & 3) * ring_n + k];
        WAVE_SYNC();
        {
            const int sh = 4 * (k >> 2);
            if (!CHROMA) {
                int px[20];
#pragma unroll
                for (int i = 0; i < 20; i++) px[i] = tile[i * D3_TS + 16 + k];
                if (__ballot((bhl | bhh) != 0)) {
                    {
                        const int bS = (int)((bhl >> sh) & 15);
                        if (__ballot(bS != 0)) edge_luma2<true>(PT, px[0], px[1], px[2], px[3], px[4], px[5], px[6], px[7], bS, __ballot(bS == 4) != 0);
                    }
#pragma unroll
                    for (int e = 1; e < 4; e++) {
                        const int bS = (int)(((e < 2 ? bhl : bhh) >> (16 * (e & 1) + sh)) & 15);
                        if (__ballot(bS != 0)) edge_luma2<false>(PI, px[4 * e], px[4 * e + 1], px[4 * e + 2], px[4 * e + 3], px[4 * e + 4], px[4 * e + 5], px[4 * e + 6], px[4 * e + 7], bS, false);
                    }
                    if (act) {
#pragma unroll
                        for (int i = 1; i < 19; i++) tile[i * D3_TS + 16 + k] = (uint8_t)px[i];
                    }
                }
            } else {
                int b[8];
#pragma unroll
                for (int i = 0; i < 8; i++) b[i] = tile[i * D3_TS + 16 + k];
                if (__ballot((bhl | bhh) != 0)) {
                    edge_chroma2(PT, b[0], b[1], b[2], b[3], (int)((bhl >> sh) & 15));
                    edge_chroma2(PI, b[4], b[5], b[6], b[7], (int)((bhh >> sh) & 15));
                    if (act) { tile[1 * D3_TS + 16 + k] = (uint8_t)b[1]; tile[2 * D3_TS + 16 + k] = (uint8_t)b[2]; tile[5 * D3_TS + 16 + k] = (uint8_t)b[5]; tile[6 * D3_TS + 16 + k] = (uint8_t)b[6]; }
                }
            }
        }
        WAVE_SYNC();
        D3_TICK(3);
        if (act) {
            // bottom strip -> ring (read by the row below after the next barrier)
            if (!last_row && k < strip * 4) ring[(x & 3) * ring_n + k] = *(const unsigned *)&tile[(rows_mb + (k >> 2)) * D3_TS + 16 + 4 * (k & 3)];
            // final samples into registers: row k, byte columns -4..11 (the left strip is final now), and the strip of the row above
            if (k < rows_mb) {
                unsigned *rp = (unsigned *)&tile[(k + strip) * D3_TS + 12];
                sa0 = rp[0]; sa1 = rp[1]; sa2 = rp[2]; sa3 = rp[3]; sb = rp[4];
                rp[0] = sb; // right strip becomes the next macroblock's left strip
            }
            if (my > 0 && k < strip) { const unsigned *tp = (const unsigned *)&tile[k * D3_TS + 16]; sc = make_uint4(tp[0], tp[1], tp[2], tp[3]); }
        }
        D3_TICK(4);
        // ---- E. land the prefetch (issued a whole step ago) before this step's stores queue up behind it
        // Unconditional, and the prefetched registers are "used" right here in uniform control flow: the compiler's
        // wait-count bookkeeping then knows that no load into them is pending when the stores below are issued.  Without
        // this it re-waits vmcnt(0) at the top of the next step (before overwriting them) -- i.e. for those stores.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("" ::"v"(own.x), "v"(own.y), "v"(own.z), "v"(own.w), "v"(recv.x), "v"(recv.y), "v"(recv.z), "v"(recv.w), "v"(strip_next), "v"(avail_next));
        D3_TICK(5);
        if (pub && k == 0) st_sc1(prog_my, (unsigned)x);
        if (pf) {
            if (k < rows_mb) { unsigned *d = (unsigned *)&tile[(k + strip) * D3_TS + 16]; d[0] = own.x; d[1] = own.y; d[2] = own.z; d[3] = own.w; }
            if (k >= 12) { unsigned *d = &recw[4 * (k - 12)]; d[0] = recv.x; d[1] = recv.y; d[2] = recv.z; d[3] = recv.w; }
            stripv = strip_next; avail = avail_next;
        }
        // ---- F. stores (nobody inside this launch reads them back)
        if (act) {
            if (k < keep) {
                uint8_t *dst = plane + (row0 + k) * stride + x0b;
                if (x > 0) stg128u(dst - 4, sa0, sa1, sa2, sa3);
                else { stg32(dst, sa1); stg32(dst + 4, sa2); stg32(dst + 8, sa3); }
                if (x == mbw - 1) stg32(dst + 12, sb); // no right neighbour will patch columns 12..15
            }
            if (my > 0 && k < strip) stg128(plane + (row0 - strip + k) * stride + x0b, sc); // the strip of the row above is final after this top edge
        }
        D3_TICK(6);
    }
#if defined(D3_PROF) && D3_PROF == 2
    pc[7] = __builtin_readcyclecounter() - loop_t0; pc[6] = (unsigned long long)nsteps;
#endif
#ifdef D3_PROF
    if (lane == 0 && (wave == 0 || wave == 3) && band < 2) {
        unsigned *o = (unsigned *)(ctx->dbrec) + (((CHROMA ? 2 : 0) + band) * 2 + (wave ? 1 : 0)) * 8; // debug build only: overwrites the first records after use
        for (int i = 0; i < 8; i++) o[i] = (unsigned)(pc[i] >> 0);
    }
#endif
}

__global__ __launch_bounds__(256) void deblock_band16_kernel(db_args a) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[D3_ROWS * sizeof(d3_luma)];
    const int nl = gridDim.x >> 1;
    if ((int)blockIdx.x < nl) band16_body<false>(a, a.band0 + blockIdx.x, a.nb_total, lds);
    else band16_body<true>(a, a.band0 + blockIdx.x - nl, a.nb_total, lds);
}

// =================================================================== hand-over to the host entropy coder
// The kernels above leave 408 int16 per macroblock in HBM; at streaming bit rates almost all of
// them are zero.  Instead of copying the dense array over PCIe (6.6 MB per 1080p picture) the
// device packs what the CAVLC writer will actually read, in the order it reads it, straight into
// the pinned host buffer: per macroblock, 32-byte blocks
//     [Intra4x4 modes, if mb_type == 2] [Intra16x16 DC, if NZ_LDC] [luma blkIdx b for every set bit b of nzmask]
//     [chroma DC (Cb 4 + Cr 4), if NZ_CBDC | NZ_CRDC] [chroma AC block i for every set bit 16 + i]
// The host walks the stream with a running pointer and needs no per-macroblock offsets.
#define PACK_CAND 27
DEV int pack_count(unsigned nz, unsigned mb_type) {
    return __popc(nz & 0x01FFFFFFu) + ((nz & (NZ_CBDC | NZ_CRDC)) ? 1 : 0) + (mb_type == 2 ? 1 : 0);
}
// exclusive prefix sum of the block counts: one workgroup, thread t owns a run of consecutive macroblocks
__global__ __launch_bounds__(1024) void levels_scan_kernel(const mb_info_t *__restrict__ mbi, int nmb, int mbw, unsigned *__restrict__ off,
                                                           unsigned *__restrict__ hdr, const unsigned *__restrict__ err) {
    __shared__ unsigned wsum[16];
    const int tid = threadIdx.x, per = (nmb + 1023) / 1024, base = tid * per;
    unsigned mine = 0;
    for (int i = 0; i < per; i++) {
        const int mb = base + i;
        if (mb < nmb) { const uint4 r = ldg128(&mbi[mb]); mine += (unsigned)pack_count(r.z, r.y & 255); }
    }
    unsigned incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const unsigned v = __shfl_up(incl, d); if ((tid & 63) >= d) incl += v; }
    if ((tid & 63) == 63) wsum[tid >> 6] = incl;
    __syncthreads();
    unsigned before = 0;
    for (int w = 0; w < (tid >> 6); w++) before += wsum[w];
    unsigned run = before + incl - mine;
    for (int i = 0; i < per; i++) {
        const int mb = base + i;
        if (mb < nmb) {
            off[mb] = run;
            if (mb % mbw == 0) hdr[2 + mb / mbw] = run; // where each macroblock row starts: lets the host code rows in parallel
            const uint4 r = ldg128(&mbi[mb]); run += (unsigned)pack_count(r.z, r.y & 255);
        }
    }
    if (tid == 1023) { hdr[0] = run; hdr[1] = ldg32(err); } // total blocks; sticky error word of the band deblocker
}
// one wave per macroblock: lane c < 27 is one candidate block of the stream order above
__global__ __launch_bounds__(256) void levels_pack_kernel(const mb_info_t *__restrict__ mbi, const int16_t *__restrict__ levels, int nmb,
                                                          const unsigned *__restrict__ off, mb_info_t *__restrict__ h_mbi, int16_t *__restrict__ h_packed) {
    const int mb = blockIdx.x * 4 + (threadIdx.x >> 6), c = threadIdx.x & 63;
    if (mb >= nmb) return;
    const uint4 r = ldg128(&mbi[mb]);
    const unsigned nz = r.z, type = r.y & 255;
    bool present = false;
    int src = 0; // int16 offset inside the macroblock's 408 levels
    if (c == 0) { present = type == 2; src = L_LDC; }
    else if (c == 1) { present = (nz & NZ_LDC) != 0; src = L_LDC; }
    else if (c < 18) { present = (nz >> (c - 2)) & 1; src = L_LUMA + (c - 2) * 16; }
    else if (c == 18) { present = (nz & (NZ_CBDC | NZ_CRDC)) != 0; src = L_CDC; }
    else if (c < PACK_CAND) { present = (nz >> (16 + c - 19)) & 1; src = L_CAC + (c - 19) * 16; }
    const unsigned long long m = __ballot(present);
    if (present) {
        const int rank = __popcll(m & ((1ull << c) - 1));
        const int16_t *sp = levels + (size_t)mb * MB_LEVELS + src;
        int16_t *dp = h_packed + ((size_t)ldg32(&off[mb]) + rank) * 16;
        const uint4 a = ldg128(sp), b = ldg128(sp + 8);
        stg128(dp, a); stg128(dp + 8, b);
    }
    if (c == PACK_CAND) stg128(&h_mbi[mb], r);
}
void k_launch_pack(const mb_info_t *d_mbi, const int16_t *d_levels, int nmb, int mbw, unsigned *d_off, mb_info_t *h_mbi, int16_t *h_packed,
                   unsigned *h_hdr, const unsigned *d_err, hipStream_t s) {
    hipLaunchKernelGGL(levels_scan_kernel, dim3(1), dim3(1024), 0, s, d_mbi, nmb, mbw, d_off, h_hdr, d_err);
    hipLaunchKernelGGL(levels_pack_kernel, dim3((nmb + 3) / 4), dim3(256), 0, s, d_mbi, d_levels, nmb, d_off, h_mbi, h_packed);
}

// =================================================================== staging helper
// Replicate the last visible column/row into the coded-size margin of a staged source surface.
__global__ void pad_kernel(uint8_t *y, uint8_t *uv, int stride, int vw, int vh, int W, int H) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int nY = W * H;
    if (i < nY) {
        int r = i / W, c = i - r * W;
        if (r >= vh || c >= vw) y[(size_t)r * stride + c] = y[(size_t)(r < vh ? r : vh - 1) * stride + (c < vw ? c : vw - 1)];
    } else if (i < nY + nY / 2) {
        int j = i - nY, r = j / W, c = j - r * W;
        if (r >= vh / 2 || c >= vw) {
            int sc = c < vw ? c : vw - 2 + (c & 1);
            uv[(size_t)r * stride + c] = uv[(size_t)(r < vh / 2 ? r : vh / 2 - 1) * stride + sc];
        }
    }
}

// =================================================================== launchers
int k_intra_diags(int mbw, int mbh) { return mbw + mbh - 1; }
int k_deblock_diags(int mbw, int mbh) { return mbw + 2 * (mbh - 1); }

// The three P-picture kernels take a macroblock-row range [row0, row1): the host overlaps the upper part of picture n+1
// with the tail of picture n's deblocking (mi355enc.cpp, enqueue_picture).
void k_launch_me(const frame_ctx_t *h_ctx, int mbw, int row0, int row1, hipStream_t s) {
    int strips = (mbw + ME_MBS - 1) / ME_MBS;
    if (row1 > row0) hipLaunchKernelGGL(me_kernel, dim3(strips * (row1 - row0)), dim3(64 * ME_MBS), 0, s, *h_ctx, row0);
}
void k_launch_subpel(const frame_ctx_t *h_ctx, int mbw, int row0, int row1, hipStream_t s) {
    if (row1 > row0) hipLaunchKernelGGL(subpel_kernel, dim3((mbw * (row1 - row0) + 3) / 4), dim3(256), 0, s, *h_ctx, row0 * mbw, row1 * mbw);
}
void k_launch_inter(const frame_ctx_t *h_ctx, int mbw, int row0, int row1, hipStream_t s) {
    int pairs = (mbw * (row1 - row0) + 1) / 2;
    if (row1 > row0) hipLaunchKernelGGL(inter_kernel, dim3((pairs + 3) / 4), dim3(256), 0, s, *h_ctx, row0 * mbw, row1 * mbw);
}
void k_launch_intra_analyse(const frame_ctx_t *d_ctx, int mbw, int mbh, hipStream_t s) {
    hipLaunchKernelGGL(intra_analyse_kernel, dim3((mbw * mbh + 3) / 4), dim3(256), 0, s, d_ctx);
}
void k_launch_intra_diag(const frame_ctx_t *d_ctx, int mbw, int mbh, int diag, hipStream_t s) {
    int y_lo = diag - (mbw - 1) > 0 ? diag - (mbw - 1) : 0;
    int y_hi = diag < mbh - 1 ? diag : mbh - 1;
    if (y_hi < y_lo) return;
    hipLaunchKernelGGL(intra_kernel, dim3(y_hi - y_lo + 1), dim3(128), 0, s, d_ctx, diag);
}
void k_launch_deblock_diag(const frame_ctx_t *d_ctx, int mbw, int mbh, int diag, hipStream_t s) {
    int y_lo = diag - (mbw - 1);
    y_lo = y_lo > 0 ? (y_lo + 1) >> 1 : 0;
    int y_hi = diag / 2 < mbh - 1 ? diag / 2 : mbh - 1;
    if (y_hi < y_lo) return;
    hipLaunchKernelGGL(deblock_kernel, dim3(y_hi - y_lo + 1), dim3(64), 0, s, d_ctx, diag);
}
int k_deblock_bands16(int mbh) { return (mbh + D3_ROWS - 1) / D3_ROWS; }
// `d_progress` holds 2 * bands counters (luma, chroma) followed by the sticky error word at d_err.  The prep kernel clears
// the counters; the band kernel may be launched in several pieces (bands [band0, band1)): a band only ever waits for the
// band above it, so pieces may run concurrently on different streams as long as the upper piece is submitted first.
void k_launch_deblock_prep(const frame_ctx_t *h_ctx, int mbw, int mbh, unsigned *d_progress, int nprog, hipStream_t s) {
    hipLaunchKernelGGL(deblock_prep_kernel, dim3((mbw * mbh + 255) / 256), dim3(256), 0, s, *h_ctx, d_progress, nprog);
}
void k_launch_deblock_bands(const frame_ctx_t *h_ctx, int mbh, int band0, int band1, unsigned *d_progress, unsigned *d_err, hipStream_t s) {
    db_args a;
    a.ctx = *h_ctx; a.progress = d_progress; a.err = d_err; a.band0 = band0; a.nb_total = k_deblock_bands16(mbh);
    if (band1 > band0) hipLaunchKernelGGL(deblock_band16_kernel, dim3(2 * (band1 - band0)), dim3(256), 0, s, a);
}
// =================================================================== input conversion to NV12
// Replaces the `videoconvert` hop of the reference's pipelines for the raw formats its sources deliver
// (/root/reference/pipeline/generic/x264_superfast_camlink:4: v4l2src ... ! videoconvert ! x264enc): planar I420
// (jpegdec, videotestsrc) and packed 4:2:2 YUY2 / UYVY (capture cards).  One thread converts an 8 x 2 luma patch and
// its 4 chroma pairs: every global access is an aligned 8- or 16-byte word, reads and writes are contiguous per
// row, so the kernel runs at copy speed (pure HBM traffic: 1.5 P in + 1.5 P out for I420, 2 P + 1.5 P for 4:2:2).
// 4:2:2 -> 4:2:0 takes the rounded mean of the two chroma rows.  The coded-size margin (width/height not a
// multiple of 16) is filled by clamping the source coordinate, so no separate padding pass is needed.
struct csc_args {
    const uint8_t *p0, *p1, *p2; // I420: Y, U, V planes; packed formats: p0 only
    int s0, s1, s2;              // their strides in bytes
    uint8_t *dy, *duv;           // NV12 destination, coded size W x H, stride W
    int vw, vh, W, H;            // visible and coded size
};
template <int FMT> // 1 I420, 2 YUY2 (Y0 U Y1 V), 3 UYVY (U Y0 V Y1)
__global__ __launch_bounds__(256) void csc_kernel(csc_args a) {
    const int tx = blockIdx.x * 256 + threadIdx.x, per_row = a.W >> 3, rows2 = a.H >> 1;
    if (tx >= per_row * rows2) return;
    const int ry = tx / per_row, cx = tx - ry * per_row; // output luma rows 2ry, 2ry+1; luma columns 8cx..8cx+7
    // visible width is even; a patch is either fully visible, or clamped per byte through the slow path
    const int x0 = cx * 8;
    const bool fast = x0 + 8 <= a.vw && (a.s0 & 7) == 0 && (((uintptr_t)a.p0) & 7) == 0;
    uint2 yrow[2];
    unsigned uvw[2]; // chroma of this patch: 4 (U,V) pairs = 8 bytes
    if (FMT == 1) {
        const int cy = (2 * ry < a.vh ? 2 * ry : a.vh - 2) >> 1;
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const int sy = 2 * ry + r < a.vh ? 2 * ry + r : a.vh - 1;
            const uint8_t *sp = a.p0 + (size_t)sy * a.s0;
            if (fast) yrow[r] = ldg64(sp + x0);
            else {
                unsigned w[2] = {0, 0};
                for (int i = 0; i < 8; i++) { const int sx = x0 + i < a.vw ? x0 + i : a.vw - 1; w[i >> 2] |= ldg8(sp + sx) << (8 * (i & 3)); }
                yrow[r] = make_uint2(w[0], w[1]);
            }
        }
        unsigned u = 0, v = 0;
        const uint8_t *up = a.p1 + (size_t)cy * a.s1, *vp = a.p2 + (size_t)cy * a.s2;
        const int cw = a.vw >> 1;
        if (cx * 4 + 4 <= cw && ((a.s1 | a.s2) & 3) == 0 && ((((uintptr_t)a.p1) | ((uintptr_t)a.p2)) & 3) == 0) { u = ldg32(up + cx * 4); v = ldg32(vp + cx * 4); }
        else
            for (int i = 0; i < 4; i++) { const int sx = cx * 4 + i < cw ? cx * 4 + i : cw - 1; u |= ldg8(up + sx) << (8 * i); v |= ldg8(vp + sx) << (8 * i); }
        uvw[0] = (u & 0xFF) | ((v & 0xFF) << 8) | ((u & 0xFF00) << 8) | ((v & 0xFF00) << 16);
        uvw[1] = ((u >> 16) & 0xFF) | (((v >> 16) & 0xFF) << 8) | (((u >> 24) & 0xFF) << 16) | ((v >> 24) << 24);
    } else {
        unsigned c[2][2]; // per source row: 4 (U,V) pairs
        const int base = 2 * ry < a.vh ? 2 * ry : a.vh - 2; // margin rows repeat the last chroma row (mean of the last two source rows)
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const int sy = base + r;
            const uint8_t *sp = a.p0 + (size_t)sy * a.s0;
            unsigned w[4];
            if (fast && (a.s0 & 15) == 0 && (((uintptr_t)a.p0) & 15) == 0) { const uint4 q = ldg128(sp + 2 * x0); w[0] = q.x; w[1] = q.y; w[2] = q.z; w[3] = q.w; }
            else
                for (int i = 0; i < 4; i++) { const int sx = x0 + 2 * i < a.vw ? x0 + 2 * i : a.vw - 2; w[i] = ldg8(sp + 2 * sx) | (ldg8(sp + 2 * sx + 1) << 8) | (ldg8(sp + 2 * sx + 2) << 16) | (ldg8(sp + 2 * sx + 3) << 24); }
            unsigned yy[2] = {0, 0}, cc[2] = {0, 0};
#pragma unroll
            for (int i = 0; i < 4; i++) { // one macropixel: 2 luma + (U,V)
                const unsigned m = w[i];
                const unsigned y1 = FMT == 2 ? (m >> 16) & 0xFF : m >> 24;
                const unsigned y0 = x0 + 2 * i >= a.vw ? y1 : (FMT == 2 ? m & 0xFF : (m >> 8) & 0xFF); // margin: the last visible sample, not the last pair
                const unsigned u = FMT == 2 ? (m >> 8) & 0xFF : m & 0xFF, v = FMT == 2 ? m >> 24 : (m >> 16) & 0xFF;
                yy[i >> 1] |= (y0 | (y1 << 8)) << (16 * (i & 1));
                cc[i >> 1] |= (u | (v << 8)) << (16 * (i & 1));
            }
            yrow[r] = make_uint2(yy[0], yy[1]);
            c[r][0] = cc[0]; c[r][1] = cc[1];
        }
        if (2 * ry >= a.vh) yrow[0] = yrow[1]; // ... and the last luma row
        uvw[0] = avg4(c[0][0], c[1][0]); uvw[1] = avg4(c[0][1], c[1][1]);
    }
    v2u t;
    t.x = yrow[0].x; t.y = yrow[0].y; *(GAS v2u *)(a.dy + (size_t)(2 * ry) * a.W + x0) = t;
    t.x = yrow[1].x; t.y = yrow[1].y; *(GAS v2u *)(a.dy + (size_t)(2 * ry + 1) * a.W + x0) = t;
    t.x = uvw[0]; t.y = uvw[1]; *(GAS v2u *)(a.duv + (size_t)ry * a.W + x0) = t;
}
int k_launch_csc(int fmt, const uint8_t *p0, const uint8_t *p1, const uint8_t *p2, int s0, int s1, int s2, uint8_t *dy, uint8_t *duv,
                 int vw, int vh, int W, int H, hipStream_t s) {
    csc_args a;
    a.p0 = p0; a.p1 = p1; a.p2 = p2; a.s0 = s0; a.s1 = s1; a.s2 = s2; a.dy = dy; a.duv = duv; a.vw = vw; a.vh = vh; a.W = W; a.H = H;
    const int n = (W >> 3) * (H >> 1);
    if (fmt == 1) hipLaunchKernelGGL(csc_kernel<1>, dim3((n + 255) / 256), dim3(256), 0, s, a);
    else if (fmt == 2) hipLaunchKernelGGL(csc_kernel<2>, dim3((n + 255) / 256), dim3(256), 0, s, a);
    else if (fmt == 3) hipLaunchKernelGGL(csc_kernel<3>, dim3((n + 255) / 256), dim3(256), 0, s, a);
    else return -1;
    return 0;
}
void k_launch_pad(uint8_t *y, uint8_t *uv, int stride, int vis_w, int vis_h, int W, int H, hipStream_t s) {
    int n = W * H + W * H / 2;
    hipLaunchKernelGGL(pad_kernel, dim3((n + 255) / 256), dim3(256), 0, s, y, uv, stride, vis_w, vis_h, W, H);
}
