// kernels_common.hpp -- what every kernel translation unit of libmi355enc shares: the constant tables, global-memory
// accessors that force global_* (not flat_*) instructions, the cross-workgroup hand-off primitives of the persistent
// kernels, small integer / DPP helpers and the 4x4 transform + quantiser + chroma block code used by both the inter and
// the intra path.  Everything is static / force-inlined: each .hip file is compiled on its own (no relocatable
// device code).  Integer arithmetic throughout (u8 samples, 16-bit levels, 32-bit accumulators): results must equal
// oracle/h264_enc_oracle.c byte for byte.  No MFMA: nothing on this path is a dense contraction.
#ifndef MI355ENC_KERNELS_COMMON_HPP
#define MI355ENC_KERNELS_COMMON_HPP
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
    uint8_t i4tab[9 * 16];                            // Intra_4x4 predictors: [mode][pixel] = (position on the neighbour line + 5) | kind << 4 (k_intra.hip)
};
static_assert(sizeof(dev_tables) % 4 == 0, "dev_tables is copied as dwords");
#define TAB_DWORDS ((int)(sizeof(dev_tables) / 4))
static __device__ const dev_tables g_tab = {
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
    {0, 1, 5, 6, 14, 15, 27, 28, 2, 4, 7, 13, 16, 26, 29, 42, 3, 8, 12, 17, 25, 30, 41, 43, 9, 11, 18, 24, 31, 40, 44, 53, 10, 19, 23, 32, 39, 45, 52, 54, 20, 22, 33, 38, 46, 51, 55, 60, 21, 34, 37, 47, 50, 56, 59, 61, 35, 36, 48, 49, 57, 58, 62, 63},
    {0x06, 0x07, 0x08, 0x09, 0x06, 0x07, 0x08, 0x09, 0x06, 0x07, 0x08, 0x09, 0x06, 0x07, 0x08, 0x09,   // 0 vertical: copy E(1 + x)
     0x04, 0x04, 0x04, 0x04, 0x03, 0x03, 0x03, 0x03, 0x02, 0x02, 0x02, 0x02, 0x01, 0x01, 0x01, 0x01,   // 1 horizontal: copy E(-1 - y)
     0x35, 0x35, 0x35, 0x35, 0x35, 0x35, 0x35, 0x35, 0x35, 0x35, 0x35, 0x35, 0x35, 0x35, 0x35, 0x35,   // 2 DC
     0x26, 0x27, 0x28, 0x29, 0x27, 0x28, 0x29, 0x2A, 0x28, 0x29, 0x2A, 0x2B, 0x29, 0x2A, 0x2B, 0x2C,   // 3 diagonal down-left
     0x24, 0x25, 0x26, 0x27, 0x23, 0x24, 0x25, 0x26, 0x22, 0x23, 0x24, 0x25, 0x21, 0x22, 0x23, 0x24,   // 4 diagonal down-right
     0x15, 0x16, 0x17, 0x18, 0x24, 0x25, 0x26, 0x27, 0x23, 0x15, 0x16, 0x17, 0x22, 0x24, 0x25, 0x26,   // 5 vertical-right
     0x14, 0x24, 0x25, 0x26, 0x13, 0x23, 0x14, 0x24, 0x12, 0x22, 0x13, 0x23, 0x11, 0x21, 0x12, 0x22,   // 6 horizontal-down
     0x16, 0x17, 0x18, 0x19, 0x26, 0x27, 0x28, 0x29, 0x17, 0x18, 0x19, 0x1A, 0x27, 0x28, 0x29, 0x2A,   // 7 vertical-left
     0x13, 0x22, 0x12, 0x21, 0x12, 0x21, 0x11, 0x20, 0x11, 0x20, 0x01, 0x01, 0x01, 0x01, 0x01, 0x01}}; // 8 horizontal-up


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
// 16 bytes at a wave-uniform address of memory nothing writes while the kernel runs, through the scalar cache (s_load_dwordx4: the result is in SGPRs, no VMEM, no VALU)
DEV uint4 ldc128(const void *p) { const v4u v = *(const __attribute__((address_space(4))) v4u *)p; return make_uint4(v.x, v.y, v.z, v.w); }
DEV int ldg16(const void *p) { return *(const GAS int16_t *)p; }
DEV void stg8(void *p, unsigned v) { *(GAS uint8_t *)p = (uint8_t)v; }
DEV void stg16(void *p, int v) { *(GAS int16_t *)p = (int16_t)v; }
DEV void stg32(void *p, unsigned v) { *(GAS unsigned *)p = v; }
DEV void stg64(void *p, uint2 v) { v2u t; t.x = v.x; t.y = v.y; *(GAS v2u *)p = t; }
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

// QP_Y of a macroblock: the picture's, or with adaptive quantisation the picture's plus the macroblock's offset (oracle: mb_qp)
DEV int mb_qp_dev(const frame_ctx_t *ctx, int mbn) {
    if (!ctx->qp_off) return ctx->qp; // (uniform)
    const int q = ctx->qp + (int)(int8_t)ldg8(ctx->qp_off + mbn);
    return q < 0 ? 0 : (q > 51 ? 51 : q);
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
// 6.4.8 for the row above: another slice's macroblocks are not available (pictures cut into slices of ctx->slice_rows rows; oracle: top_ok)
DEV bool row_has_top(const frame_ctx_t *ctx, int my) { return my > 0 && (ctx->slice_rows <= 0 || my % ctx->slice_rows != 0); }
// 8.7 filterTopMbEdgeFlag: the top macroblock edge is filtered unless it is the picture's, or (disable_deblocking_filter_idc 2) a slice's; and whether row my is
// the last of its slice in that sense (nothing below it touches its bottom lines)
DEV bool db_has_top(const frame_ctx_t *ctx, int my) { return my > 0 && !(ctx->slice_dbf == 2 && ctx->slice_rows > 0 && my % ctx->slice_rows == 0); }
DEV bool db_slice_last(const frame_ctx_t *ctx, int my) { return ctx->slice_dbf == 2 && ctx->slice_rows > 0 && (my + 1) % ctx->slice_rows == 0; }
// where the last of the ME_ITERS selection iterations leaves the whole-sample vector field (they walk imv_a -> imv_b -> imv_c -> imv_a ...)
DEV const imv_t *k_final_imv_dev(const frame_ctx_t *ctx) { return ME_ITERS % 3 == 0 ? ctx->imv_a : ME_ITERS % 3 == 1 ? ctx->imv_b : ctx->imv_c; }

// ------------------------------------------------------------------ cross-workgroup hand-off inside a persistent launch
// Agent-scope (sc1, L1-bypassing) accesses for data one workgroup produces and another consumes while both run, and a bounded
// wait on a monotonic progress counter (MI355X_MICROARCH.md, "Valid forms": producer stores the data sc1, s_waitcnt vmcnt(0),
// then stores the counter sc1; consumer polls the counter and reads the data with sc1 loads).
#define DB_SPIN_MAX (1 << 20)
// band-done words of the band deblocker (read by the next picture's pmb_kernel): DB_DONE_COPIES copies, DB_DONE_STRIDE words apart
// (4 KB: different memory channels), each {luma epoch, chroma epoch} per band
#define DB_DONE_COPIES 16
#define DB_DONE_STRIDE 1024
// development builds (-DTL_PROF, tests/devtools/timeline.py): wall-clock marks (100 MHz) per picture epoch, left in ctx->dbrec
#ifdef TL_PROF
DEV void tl_first(const frame_ctx_t *ctx, int slot) { if (threadIdx.x == 0) *((volatile unsigned long long *)ctx->dbrec + (ctx->epoch & 63) * 16 + slot) = wall_clock64(); }
DEV void tl_last(const frame_ctx_t *ctx, int slot) { if (threadIdx.x == 0) atomicMax((unsigned long long *)ctx->dbrec + (ctx->epoch & 63) * 16 + slot, (unsigned long long)wall_clock64()); }
#else
#define tl_first(c, s) ((void)0)
#define tl_last(c, s) ((void)0)
#endif
DEV unsigned ld_sc1(const unsigned *p) { return __hip_atomic_load((const GAS unsigned *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
DEV void st_sc1(unsigned *p, unsigned v) { __hip_atomic_store((GAS unsigned *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
DEV uint2 ld64_sc1(const uint2 *p) { const unsigned long long v = __hip_atomic_load((const GAS unsigned long long *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return make_uint2((unsigned)v, (unsigned)(v >> 32)); }
DEV void st64_sc1(uint2 *p, uint2 v) { __hip_atomic_store((GAS unsigned long long *)p, (unsigned long long)v.x | ((unsigned long long)v.y << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// a store the consumer may read while this kernel still runs (SC1), or an ordinary one
template <bool SC1> DEV void stx32(void *p, unsigned v) { if (SC1) st_sc1((unsigned *)p, v); else stg32(p, v); }
template <bool SC1> DEV void stx64(void *p, uint2 v) { if (SC1) st64_sc1((uint2 *)p, v); else stg64(p, v); }
template <bool SC1> DEV void st_mbinfo_x(mb_info_t *p, const mb_info_t &m) {
    if (!SC1) { st_mbinfo(p, m); return; }
    st64_sc1((uint2 *)p, make_uint2(((unsigned)(uint16_t)m.mvx) | ((unsigned)(uint16_t)m.mvy << 16), (unsigned)m.mb_type | ((unsigned)m.i16_mode << 8) | ((unsigned)m.chroma_mode << 16) | ((unsigned)m.qp << 24)));
    st64_sc1((uint2 *)p + 1, make_uint2(m.nzmask, m.cost));
}
DEV int db_wait_get(unsigned *progress, unsigned *err, int need) {
    int spins = 0, v;
    while ((v = (int)ld_sc1(progress)) < need) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > DB_SPIN_MAX) { st_sc1(err, 16u); return 0x7FFFFFFF; } if ((spins & 1023) == 0 && ld_sc1(err)) { return 0x7FFFFFFF; } // bounded; once tripped, nobody waits again
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

// sum over the 16 lanes of a DPP row (every lane receives it)
DEV int wave16_sum(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x128, 0xF, 0xF, false); // row_ror:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x124, 0xF, 0xF, false); // row_ror:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false);  // quad_perm:[2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false);  // quad_perm:[1,0,3,2]
    return v;
}
// value of lane k of this lane's quad (k = 0..3), and of row r of this lane's column in a 4x4 tile laid out on 16 lanes
template <int K> DEV int quad_bcast(int v) { return __builtin_amdgcn_update_dpp(0, v, K * 0x55, 0xF, 0xF, false); }

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
DEV int mad24(int a, int b, int c) { return __mul24(a, b) + c; } // v_mad_i32_i24: operands must fit 24 bits
// sum over the whole wave, returned uniformly (an SGPR): row sums by DPP, then row_bcast15 / row_bcast31 carry them down
// the rows (GFX9 DPP controls) and lane 63 holds the total -- no LDS-crossbar shuffles
DEV int wave64_sum(int v) {
    v = wave16_sum(v);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false); // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false); // row_bcast:31 into rows 2 and 3
    return __builtin_amdgcn_readlane(v, 63);
}
// minimum over the whole wave, returned uniformly: the same DPP ladder (six v_min_u32_dpp and a readlane instead of six LDS-crossbar shuffles)
DEV unsigned wave64_umin(unsigned v) {
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp(-1, (int)v, 0x128, 0xF, 0xF, false)); // row_ror:8
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp(-1, (int)v, 0x124, 0xF, 0xF, false)); // row_ror:4
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp(-1, (int)v, 0x4E, 0xF, 0xF, false));  // quad_perm:[2,3,0,1]
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp(-1, (int)v, 0xB1, 0xF, 0xF, false));  // quad_perm:[1,0,3,2]
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp(-1, (int)v, 0x142, 0xA, 0xF, false)); // row_bcast:15 into rows 1 and 3
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp(-1, (int)v, 0x143, 0xC, 0xF, false)); // row_bcast:31 into rows 2 and 3
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
// lane ^ 4 / lane ^ 12 inside a row of 16 lanes, two DPP moves each: quad reversal (lane ^ 3) followed by the half-row mirror
// (lane ^ 7) or the row mirror (lane ^ 15)
DEV int row_xor4(int v) { return __builtin_amdgcn_update_dpp(0, __builtin_amdgcn_update_dpp(0, v, 0x1B, 0xF, 0xF, false), 0x141, 0xF, 0xF, false); }
DEV int row_xor12(int v) { return __builtin_amdgcn_update_dpp(0, __builtin_amdgcn_update_dpp(0, v, 0x1B, 0xF, 0xF, false), 0x140, 0xF, 0xF, false); }
DEV int row_xor8(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x128, 0xF, 0xF, false); } // row_ror:8 == lane ^ 8 within the row
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

// ---- 4x4 blocks on four lanes (lane bits 3:2 = row in block): shared by the P-macroblock and the intra kernels
DEV void fwd_rows4(int *r) { // 4-point forward core transform, in place
    const int e0 = r[0] + r[3], e1 = r[1] + r[2], e2 = r[1] - r[2], e3 = r[0] - r[3];
    r[0] = e0 + e1; r[1] = 2 * e3 + e2; r[2] = e0 - e1; r[3] = e3 - 2 * e2;
}
DEV void inv_rows4(int *d) { // 8.5.12.2, one row
    const int e0 = d[0] + d[2], e1 = d[0] - d[2], e2 = (d[1] >> 1) - d[3], e3 = d[1] + (d[3] >> 1);
    d[0] = e0 + e3; d[1] = e1 + e2; d[2] = e1 - e2; d[3] = e0 - e3;
}
struct col_bf { int s1, fo, fp, is, io, ip; }; // per-lane butterfly coefficients of the column direction (see k_intra.hip, Intra_4x4)
DEV col_bf make_col_bf(int py) {
    col_bf c;
    c.s1 = py < 2 ? 1 : -1; c.fo = py == 1 ? -1 : 1; c.fp = py < 2 ? 1 : py == 2 ? 2 : -2;
    c.is = py >> 1; c.io = py == 1 ? -1 : 1; c.ip = py == 2 ? -1 : 1;
    return c;
}
DEV int fwd_col(int v, const col_bf &c) { // rows of the block are 4 lanes apart; result: frequency F[py], F = 0 2 1 3
    int t = mad24(v, c.s1, row_xor12(v));
    return mad24(t, c.fo, __mul24(row_xor4(t), c.fp));
}
DEV int inv_col(int v, const col_bf &c) { // takes frequency order F[py], returns row py
    int t = mad24(v >> c.is, c.io, __mul24(row_xor4(v), c.ip));
    return mad24(t, c.s1, row_xor12(t));
}
// Chroma of one macroblock on lanes 0..31 of a wave, four lanes per 4x4 block: lane bit 4 = block row, bits 3:2 = row in
// block, bit 1 = plane (Cb, Cr), bit 0 = block column -- a 16-lane row holds Cb0 Cb1 Cr0 Cr1 of one block row, so the column
// butterflies stay inside the row.  pd: prediction of this lane's four samples, sv: its source samples.  Writes levels and
// the reconstruction (Cb lanes fetch their Cr partners with one DPP move and store interleaved 8-byte segments; also into
// the 8 x 16 LDS tile `lrec` if given).  Every lane of the wave must call (DPP, ballots).  nz8: blocks with AC levels in
// record order 4 c + 2 by + bx; dc2: bit 0 Cb / bit 1 Cr have DC levels.
// decimate (inter macroblocks, x264 dct-decimate): a plane whose four blocks' AC run/level score stays below 7 loses its AC levels.
// ok == false: nothing is stored (the skip probe only wants nz8 / dc2).
DEV int dec_score_mask(unsigned M) { // decimate score of a +-1-only block from its significance mask with a sentinel bit below the first position
    const unsigned c0 = (unsigned)__popc(M & (M << 1));
    const unsigned a1 = M & ~(M << 1) & ~1u, a3 = a1 & ~(M << 2) & ~(M << 3), a6 = a3 & ~(M << 4) & ~(M << 5) & ~(M << 6);
    return (int)(3 * c0 + 2 * (unsigned)__popc(a1) - (unsigned)__popc(a3) - (unsigned)__popc(a6));
}
DEV void chroma_rows4(const frame_ctx_t *ctx, const dev_tables *T, int16_t *lv, int cx0, int cy0, int lane, const int *pd, const int *sv, int qp,
                      bool intra, bool ok, uint8_t *lrec, unsigned &nz8, unsigned &dc2, bool decimate = false, int all_drop = 0, bool sc1 = false) {
    const bool cl = lane < 32;
    const int py = (lane >> 2) & 3, fy = ((py & 1) << 1) | (py >> 1);
    const int cby = (lane >> 4) & 1, c = (lane >> 1) & 1, cbx = lane & 1, cy = cby * 4 + py, cxb = cbx * 4;
    const col_bf cb = make_col_bf(py);
    const int kz0 = (int)((0xFEA9DB83C7426510ull >> (16 * fy)) & 0xFFFF); // zig-zag positions of raster 4 fy + 0 .. 3, a nibble each
    const qparams q = make_q(T, T->qpc[qp], intra);
    const int mfe = py < 2 ? q.mf[0] : q.mf[2], mfo = py < 2 ? q.mf[2] : q.mf[1], ve = py < 2 ? q.v[0] : q.v[2], vo = py < 2 ? q.v[2] : q.v[1];
    int x[4], lev[4], cf[4];
#pragma unroll
    for (int i = 0; i < 4; i++) x[i] = sv[i] - pd[i];
    fwd_rows4(x);
#pragma unroll
    for (int i = 0; i < 4; i++) cf[i] = fwd_col(x[i], cb);
    // 8.5.11: the plane's four DC terms (element 0 of the py == 0 lanes) through the 2x2 Hadamard -- block column is lane
    // bit 0, block row lane bit 4 -- quantised with doubled rounding, transformed back and scaled
    const int sbx = cbx ? -1 : 1, sby = cby ? -1 : 1;
    int hd = mad24(cf[0], sbx, quad_xor<1>(cf[0]));
    hd = mad24(hd, sby, __shfl_xor(hd, 16, 64));
    int ldc = quant1(hd, q.mf[0], 2 * q.f, q.qbits + 1);
    const bool dcl = py == 0; // this lane holds a DC term
    unsigned sig = 0, big = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        lev[i] = (i == 0 && dcl) ? 0 : quant1(cf[i], (i & 1) ? mfo : mfe, q.f, q.qbits);
        sig |= (lev[i] != 0 ? 1u : 0u) << ((kz0 >> (4 * i)) & 15);
        big |= (lev[i] > 1 || lev[i] < -1) ? 1u : 0u;
    }
    if (decimate) {
        int mm = (int)(sig | (big << 16)); // OR over the block's four lanes (l, l+4, l+8, l+12 of the 16-lane row)
        mm |= __builtin_amdgcn_update_dpp(0, mm, 0x128, 0xF, 0xF, false);
        mm |= __builtin_amdgcn_update_dpp(0, mm, 0x124, 0xF, 0xF, false);
        int s = ((unsigned)mm >> 16) ? 9 : dec_score_mask(((unsigned)mm & 0xFFFFu) | 1u); // AC only: position 0 is the sentinel
        s += quad_xor<1>(s);            // the plane's two block columns ...
        s += __shfl_xor(s, 16, 64);     // ... and two block rows
        if (s < 7) { lev[0] = lev[1] = lev[2] = lev[3] = 0; }
    }
    if (all_drop) { // rate control's ladder for I pictures: chroma levels (both planes, DC and AC) summing to no more than the threshold are not sent
        const int sm = wave64_sum(cl ? (dcl ? iabs(ldc) : 0) + iabs(lev[0]) + iabs(lev[1]) + iabs(lev[2]) + iabs(lev[3]) : 0);
        if (sm <= all_drop) { ldc = 0; lev[0] = lev[1] = lev[2] = lev[3] = 0; }
    }
    int g = mad24(ldc, sbx, quad_xor<1>(ldc));
    g = mad24(g, sby, __shfl_xor(g, 16, 64));
    const int dcc = ((g * 16 * q.v[0]) << q.shift) >> 5;
#pragma unroll
    for (int i = 0; i < 4; i++) x[i] = (lev[i] * ((i & 1) ? vo : ve)) << q.shift;
    if (dcl) x[0] = dcc;
    const int b = cby * 2 + cbx;
    if (ok && cl) {
#pragma unroll
        for (int i = 0; i < 4; i++) stg16(&lv[L_CAC + (4 * c + b) * 16 + ((kz0 >> (4 * i)) & 15)], lev[i]);
        if (dcl) stg16(&lv[L_CDC + 4 * c + b], ldc);
    }
    inv_rows4(x);
    int o[4];
#pragma unroll
    for (int i = 0; i < 4; i++) o[i] = clip255(pd[i] + ((inv_col(x[i], cb) + 32) >> 6));
    const unsigned mine = pack4(o[0], o[1], o[2], o[3]), other = (unsigned)quad_xor<2>((int)mine);
    if (ok && cl && c == 0) {
        uint2 out;
        out.x = __builtin_amdgcn_perm(other, mine, 0x05010400u); // U0 V0 U1 V1
        out.y = __builtin_amdgcn_perm(other, mine, 0x07030602u); // U2 V2 U3 V3
        uint8_t *dst = ctx->rec_uv + (size_t)(cy0 + cy) * ctx->stride + 2 * (cx0 + cxb);
        if (sc1) st64_sc1((uint2 *)dst, out); // intra macroblocks of P pictures: the deblocker of the same picture follows the reconstruction while it is written
        else stg64(dst, out);
        if (lrec) *(uint2 *)&lrec[cy * 16 + 2 * cxb] = out;
    }
    const unsigned long long bal = __ballot(cl && (lev[0] | lev[1] | lev[2] | lev[3]) != 0);
    const unsigned long long t = bal | (bal >> 4) | (bal >> 8) | (bal >> 12); // bit 16 cby + 2 c + cbx
    const int k8 = lane & 7, kc = k8 >> 2, kby = (k8 >> 1) & 1, kbx = k8 & 1;
    nz8 = (unsigned)(__ballot(lane < 8 && ((t >> (16 * kby + 2 * kc + kbx)) & 1)) & 0xFFull);
    const unsigned long long dcb = __ballot(cl && dcl && ldc != 0);
    dc2 = ((dcb & 0x00030003ull) ? 1u : 0u) | ((dcb & 0x000C000Cull) ? 2u : 0u); // py == 0 lanes with c == 0 / c == 1
}

// 6-tap filter of 8.4.2.2.1 (unrounded), and the rounded per-byte mean of two packed words
DEV int tap6(int a, int b, int c, int d, int e, int f) { return a - 5 * b + 20 * c + 20 * d - 5 * e + f; }
DEV unsigned avg4(unsigned a, unsigned b) { return (a | b) - (((a ^ b) >> 1) & 0x7F7F7F7Fu); } // per byte (a + b + 1) >> 1
#endif
