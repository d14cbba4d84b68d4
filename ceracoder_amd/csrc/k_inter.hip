// k_inter.hip -- P macroblocks: motion compensation, transform, quantisation, reconstruction
// Hand-written HIP for gfx950 (CDNA4, wave64); part of libmi355enc (see kernels_common.hpp).
#include "kernels_common.hpp"

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
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); // an SGPR: what is derived from it is scalar control flow
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
// =================================================================== launcher
void k_launch_inter(const frame_ctx_t *h_ctx, int mbw, int row0, int row1, hipStream_t s) {
    int pairs = (mbw * (row1 - row0) + 1) / 2;
    if (row1 > row0) hipLaunchKernelGGL(inter_kernel, dim3((pairs + 3) / 4), dim3(256), 0, s, *h_ctx, row0 * mbw, row1 * mbw);
}
