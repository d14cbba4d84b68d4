// intra_mb.hpp -- reconstruction of intra macroblocks, shared by the intra kernels (k_intra.hip) and by the band deblocker's launch
// (k_deblock.hip), which carries the intra macroblocks of P pictures as workgroups of its own.  Everything is force-inlined: each .hip
// file is compiled on its own (no relocatable device code).
#ifndef MI355ENC_INTRA_MB_HPP
#define MI355ENC_INTRA_MB_HPP
#include "kernels_common.hpp"
#include <cstddef>

// =================================================================== Intra_8x8 (High profile; oracle: pred8x8 / i8_refs / intra8x8_recon)
// The 25 reference samples of an 8x8 block, filtered (8.3.2.2.1), as one edge array in LDS: E(0) = p'[-1,-1], E(k) = p'[k-1,-1] (k = 1..16),
// E(-k) = p'[-1,k-1] (k = 1..8), stored at e[i + 8].  rt[0..15] / rl[0..7] / rc: the raw samples above (the right half already replaced by
// rt[7] where the block above-right is not available), to the left, and the corner; lanes 0..24 compute one entry each.
DEV void i8_edges(int lane, const int *rt, const int *rl, int rc, bool up, bool lf, bool ul, int *e) {
    if (lane < 25) {
        const int i = lane - 8;
        int v = 0;
        if (i == 0) { if (ul) v = (up && lf) ? (rt[0] + 2 * rc + rl[0] + 2) >> 2 : up ? (3 * rc + rt[0] + 2) >> 2 : lf ? (3 * rc + rl[0] + 2) >> 2 : rc; }
        else if (i > 0) {
            const int x = i - 1;
            if (up) v = x == 0 ? (ul ? (rc + 2 * rt[0] + rt[1] + 2) >> 2 : (3 * rt[0] + rt[1] + 2) >> 2) : x == 15 ? (rt[14] + 3 * rt[15] + 2) >> 2 : (rt[x - 1] + 2 * rt[x] + rt[x + 1] + 2) >> 2;
        } else {
            const int y = -i - 1;
            if (lf) v = y == 0 ? (ul ? (rc + 2 * rl[0] + rl[1] + 2) >> 2 : (3 * rl[0] + rl[1] + 2) >> 2) : y == 7 ? (rl[6] + 3 * rl[7] + 2) >> 2 : (rl[y - 1] + 2 * rl[y] + rl[y + 1] + 2) >> 2;
        }
        e[lane] = v;
    }
}
DEV bool i8_mode_ok(int mode, bool up, bool lf, bool ul) {
    const bool need_up = mode == 0 || mode == 3 || mode == 7, need_left = mode == 1 || mode == 8, need_all = mode >= 4 && mode <= 6;
    return !((need_up && !up) || (need_left && !lf) || (need_all && !(up && lf && ul)));
}
// sample (x, y) of the prediction of mode `mode` (8.3.2.2.2 .. 8.3.2.2.10); dcv: the DC value (computed once per block by the caller)
DEV int i8_pred_px(const int *e, int mode, int x, int y, int dcv) {
#define E(i) e[(i) + 8]
    switch (mode) {
    case 0: return E(x + 1);
    case 1: return E(-(y + 1));
    case 2: return dcv;
    case 3: return (x == 7 && y == 7) ? (E(15) + 3 * E(16) + 2) >> 2 : (E(x + y + 1) + 2 * E(x + y + 2) + E(x + y + 3) + 2) >> 2;
    case 4: { const int i = x - y; return (E(i - 1) + 2 * E(i) + E(i + 1) + 2) >> 2; }
    case 5: {
        const int z = 2 * x - y, j = x - (y >> 1);
        if (z >= 0 && !(z & 1)) return (E(j) + E(j + 1) + 1) >> 1;
        if (z >= 0) return (E(j - 1) + 2 * E(j) + E(j + 1) + 2) >> 2;
        if (z == -1) return (E(-1) + 2 * E(0) + E(1) + 2) >> 2;
        const int k = y - 2 * x - 1;
        return (E(-(k + 1)) + 2 * E(-k) + E(-(k - 1)) + 2) >> 2; }
    case 6: {
        const int z = 2 * y - x, j = y - (x >> 1);
        if (z >= 0 && !(z & 1)) return (E(-j) + E(-(j + 1)) + 1) >> 1;
        if (z >= 0) return (E(-(j - 1)) + 2 * E(-j) + E(-(j + 1)) + 2) >> 2;
        if (z == -1) return (E(-1) + 2 * E(0) + E(1) + 2) >> 2;
        const int k = x - 2 * y - 1;
        return (E(k + 1) + 2 * E(k) + E(k - 1) + 2) >> 2; }
    case 7: { const int j = x + (y >> 1); return !(y & 1) ? (E(j + 1) + E(j + 2) + 1) >> 1 : (E(j + 1) + 2 * E(j + 2) + E(j + 3) + 2) >> 2; }
    default: {
        const int z = x + 2 * y, j = y + (x >> 1);
        if (z > 13) return E(-8);
        if (z == 13) return (E(-7) + 3 * E(-8) + 2) >> 2;
        if (!(z & 1)) return (E(-(j + 1)) + E(-(j + 2)) + 1) >> 1;
        return (E(-(j + 1)) + 2 * E(-(j + 2)) + E(-(j + 3)) + 2) >> 2; }
    }
#undef E
}
DEV int i8_dc(const int *e, bool up, bool lf) { // (every lane sums: 16 LDS reads; the block is a dependency chain of its own anyway)
    int s = 0;
    if (up) for (int i = 1; i <= 8; i++) s += e[8 + i];
    if (lf) for (int i = 1; i <= 8; i++) s += e[8 - i];
    return (up && lf) ? (s + 8) >> 4 : (up || lf) ? (s + 4) >> 3 : 128;
}
// availability of the neighbours of 8x8 block b of a macroblock (oracle: blk8_avail)
DEV void i8_avail(int b, bool has_top, bool has_left, bool has_tr, bool &up, bool &lf, bool &ul, bool &ur) {
    up = b >= 2 || has_top; lf = (b & 1) || has_left;
    ul = b == 0 ? (has_top && has_left) : b == 1 ? has_top : b == 2 ? has_left : true;
    ur = b == 0 ? has_top : b == 1 ? has_tr : b == 2;
}

// the raw neighbours of 8x8 block (bx8, by8) out of a sample tile (sample (r, c) of the macroblock, r, c >= -1, at tile[(r + 1) * stride + c + off]; the top line runs
// to c = 23: the first eight samples of the macroblock above-right): lanes 0..15 rt[], lanes 16..23 rl[]; returns the corner (every lane)
DEV int i8_gather(int lane, const uint8_t *tile, int stride, int off, int bx8, int by8, bool ur, int *rt, int *rl) {
    if (lane < 16) rt[lane] = (int)tile[by8 * stride + bx8 + ((lane >= 8 && !ur) ? 7 : lane) + off];
    else if (lane < 24) rl[lane - 16] = (int)tile[(by8 + lane - 16 + 1) * stride + bx8 - 1 + off];
    return (int)tile[by8 * stride + bx8 - 1 + off];
}
// One 8x8 residual block of an Intra_8x8 macroblock through transform, quantiser (intra rounding 1/3) and back (oracle: tq8_block_i, intra = 1; the passes of
// pmb_luma_t8 in k_motion.hip, one block): lane (x, y) hands in its residual sample and gets the reconstructed residual back; lanes 0..7 run the separable
// passes through `tile` (64 ints of LDS).  Levels go out de-interleaved the way CAVLC sends them.  Returns the mask of the 4x4 "sub-blocks" with levels.
DEV unsigned i8_tq8(const frame_ctx_t *__restrict__ ctx, const dev_tables *T, int *tile, int lane, int res, int qp, int mbn, int b8, int &rres) {
    const bool act = lane < 8;
    const int j = lane & 7, m6 = qp % 6, k6 = qp / 6;
    int v[8];
    unsigned submask = 0;
    tile[lane] = res;
    WAVE_SYNC();
    if (act) {
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = tile[j * 8 + i];
        fdct8_1d(v);
#pragma unroll
        for (int i = 0; i < 8; i++) tile[j * 8 + i] = v[i]; // (a lane's own row: no other lane reads or writes it in this pass)
    }
    WAVE_SYNC();
    if (act) {
#pragma unroll
        for (int r = 0; r < 8; r++) v[r] = tile[r * 8 + j];
        fdct8_1d(v);
#pragma unroll
        for (int r = 0; r < 8; r++) tile[r * 8 + j] = v[r];
    }
    WAVE_SYNC();
    { // quantiser and scaling: one coefficient per lane (row lane >> 3, column lane & 7)
        const int r = lane >> 3, c = lane & 7, cf = tile[lane];
        const int qbits = 16 + k6, f = (1 << qbits) / 3;
        const int cl = pos_class8(r, c), a = iabs(cf);
        int l = (int)(((long long)a * T->mf8[m6][cl] + f) >> qbits);
        l = l > 2047 ? 2047 : l;
        l = cf < 0 ? -l : l;
        const int kk = T->izz8[lane];
        stg16(&ctx->levels[(size_t)mbn * MB_LEVELS + L_LUMA + (4 * b8 + (kk & 3)) * 16 + (kk >> 2)], l);
        const int ls = 16 * T->v8[m6][cl];
        tile[lane] = qp >= 36 ? (l * ls) << (k6 - 6) : (l * ls + (1 << (5 - k6))) >> (6 - k6);
#pragma unroll
        for (int k = 0; k < 4; k++) if (__ballot(l != 0 && (kk & 3) == k)) submask |= 1u << k;
    }
    WAVE_SYNC();
    if (act) { // 8.5.13: rows, then columns
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = tile[j * 8 + i];
        idct8_1d(v);
#pragma unroll
        for (int i = 0; i < 8; i++) tile[j * 8 + i] = v[i];
    }
    WAVE_SYNC();
    if (act) {
#pragma unroll
        for (int r = 0; r < 8; r++) v[r] = tile[r * 8 + j];
        idct8_1d(v);
#pragma unroll
        for (int r = 0; r < 8; r++) tile[r * 8 + j] = (v[r] + 32) >> 6;
    }
    WAVE_SYNC();
    rres = tile[lane];
    return submask;
}

// =================================================================== intra (I) macroblocks
// Two waves per macroblock (luma, chroma), run in x + y order (left, top and top-left neighbours are then complete).
// Per-macroblock working set of the intra reconstruction, in LDS.  Filled by the caller: top / left (reconstructed
// neighbours, [plane 0 = Y, 1 = Cb, 2 = Cr][index i + 1 holds sample i, index 0 the corner]).  Produced for the neighbours
// to the right and below (persistent kernel): bottom rows into a 4-deep ring, the right column.
struct intra_lds {
    int top[3][17], left[3][17];
    // Intra_4x4: reconstructed samples incl. the row above / the column to the left, twice: sample (r, c), r, c = -1 .. 15 (top line up to
    // c = 19), lives at T4[(r + 1) * 24 + c + 4] and, transposed, at T4t[(c + 1) * 24 + r + 4].  The four samples above a block and the four
    // to its left are then one aligned dword each, and E(i) of 8.3.1.2 is T4[(4 by) * 24 + 4 bx + 3 + i] for i >= 0 (corner, top, top-right:
    // contiguous) and T4t[(4 bx) * 24 + 4 by + 3 - i] for i < 0 (the left column bottom-up: contiguous).
    __attribute__((aligned(4))) uint8_t T4[17 * 24], T4t[17 * 24];
    __attribute__((aligned(4))) uint8_t S4[256];     // source macroblock, raster
    __attribute__((aligned(8))) uint8_t crec[8 * 16]; // reconstructed chroma, interleaved Cb Cr (OUT only)
    int mode4[16];
    __attribute__((aligned(4))) uint8_t z4[2 * 16];   // Intra_4x4: the neighbour line of the (up to two) blocks of a sub-step
    unsigned cflags[2];                               // chroma wave -> luma wave: ballots of its blocks' AC / DC flags
    unsigned cseq;                                    // ... valid once this equals macroblock number + 1
    __attribute__((aligned(4))) uint8_t bot_y[4][16], bot_c[4][16];
    __attribute__((aligned(4))) uint8_t right_y[16], right_c[2][8];
    int corner[3];                                    // bottom-right sample of the macroblock before the one in right_*: the next corner
};

// Hooks of intra_compute for a caller that overlaps macroblocks at 4x4-block granularity (intra_rows_kernel): before(s) runs at the top of
// Intra_4x4 sub-step s (the blocks with bx + 2 by = s) and has to leave the neighbour samples those blocks read in L->T4; after(s) runs once
// their reconstruction is in L->T4.  own_record: the caller assembles the macroblock record itself from what luma_done / chroma_done report.
struct ic_nohook {
    static constexpr bool own_record = false;
    DEV void before(int) {}
    DEV void after(int, int, int, int, int, bool, int) {}
    DEV void luma_done(unsigned, bool) {}
    DEV void chroma_done(unsigned, unsigned) {}
};
static_assert(offsetof(intra_lds, T4t) == offsetof(intra_lds, T4) + 17 * 24, "the transposed tile follows the tile (addressed as one array)");
// Reconstruction of one intra macroblock by two waves (wave 0 luma, wave 1 chroma; the planes share nothing after the
// decisions).  Needs L->top / L->left in place and visible; dec0/dec1: the 24-byte decision of intra_analyse_kernel.
template <bool OUT, bool SC1, class HK, bool I8 = false> // I8: Intra_8x8 macroblocks may occur (the rows kernel of I pictures; everywhere else the branch is compiled out).  SC1: reconstruction stored write-through (sc1): intra_p_kernel, whose samples the deblocker reads while the kernel runs
DEV void intra_compute(const frame_ctx_t *__restrict__ ctx, const dev_tables *T, intra_lds *L, const int mx, const int my, const int wave, const int lane,
                       const uint4 dec0, const uint2 dec1, const uint2 *presrc, HK &hk) { // presrc: this lane's source samples, loaded ahead (luma: .x, one word; chroma: the 8 interleaved bytes)
    int (*top)[17] = L->top;
    int (*left)[17] = L->left;
    uint8_t *T4 = L->T4, *S4 = L->S4;
    const int mbw = ctx->mbw, stride = ctx->stride;
    const int mbn = my * mbw + mx, qp = mb_qp_dev(ctx, mbn), x0 = mx * 16, y0 = my * 16, cx0 = x0 >> 1, cy0 = y0 >> 1;
    const bool has_top = row_has_top(ctx, my), has_left = mx > 0;
    uint8_t *__restrict__ ry = ctx->rec_y;
    // Four lanes per 4x4 block (kernels_common.hpp): luma on all 64 lanes of wave 0 -- lane bits 5:4 block row, 3:2 row in block,
    // 1:0 block column -- and chroma on lanes 0..31 of wave 1 (bit 4 block row, 3:2 row in block, 1 plane, 0 block column).
    // The Intra_4x4 loop keeps its own arrangement (one pixel per lane) and reads the source from S4.
    const int py = (lane >> 2) & 3;
    const int slot = mx & 3;
    unsigned srcw = 0;   // luma wave: this lane's four source samples (row 4 by + py, columns 4 bx ..)
    int csv[4] = {0, 0, 0, 0}; // chroma wave: this lane's four source samples of its plane
    {
        const int ss = ctx->src_stride;
        if (wave == 0) {
            const int row = 4 * (lane >> 4) + py, col = 4 * (lane & 3), vh = ctx->vis_h;
            int sy = y0 + row;
            sy = sy < vh ? sy : vh - 1;
            srcw = presrc ? presrc[0].x : ldg32(ctx->src_y + (size_t)sy * ss + x0 + col);
            *(unsigned *)&S4[row * 16 + col] = srcw;
        } else {
            const int c = (lane >> 1) & 1, vh2 = ctx->vis_h >> 1;
            int sy = cy0 + 4 * ((lane >> 4) & 1) + py;
            sy = sy < vh2 ? sy : vh2 - 1;
            const uint2 w = presrc ? presrc[0] : ldg64(ctx->src_uv + (size_t)sy * ss + 2 * (cx0 + 4 * (lane & 1)));
            const unsigned lo = c ? (w.x >> 8) : w.x, hi = c ? (w.y >> 8) : w.y;
            csv[0] = (int)(lo & 255); csv[1] = (int)((lo >> 16) & 255); csv[2] = (int)(hi & 255); csv[3] = (int)((hi >> 16) & 255);
        }
    }
#define TOP(p, i) top[p][(i) + 1]
#define LEFT(p, i) left[p][(i) + 1]
    // ---- decisions were taken by intra_analyse_kernel (oracle: orc_intra_decide)
    const int mode16 = (int)(dec1.x & 255), cmode = (int)((dec1.x >> 8) & 255);
    const int itype = (int)((dec1.x >> 16) & 255);   // 0 Intra_16x16, 1 Intra_4x4, 2 Intra_8x8 (wave-uniform)
    const bool use_i4 = itype == 1, use_i8 = I8 && itype == 2;
    unsigned nz4 = 0;
    WAVE_SYNC(); // S4 is in place
    unsigned nz16 = 0, ldc_any = 0, cnz8 = 0, cdc2 = 0;
    if (use_i4 && wave == 0) {
        // ================================================================ Intra_4x4 reconstruction (8.3.1.2 + 8.5)
        // Blocks in the order bx + 2 by = s, up to two per sub-step, 16 lanes each, lane = one pixel; transforms across lanes.
        // A sub-step is a dependency chain (the next one predicts from this one's reconstruction), so everything that does not depend on
        // reconstructed samples is taken off it: mode, predictor-table entry and source sample of all ten sub-steps are gathered into
        // packed registers up front; what is left per sub-step is ONE round of LDS reads (three predictor taps at per-lane addresses, the
        // dwords above and to the left for DC), arithmetic, and the two byte writes of the reconstruction (tile and transposed tile).
        uint8_t *T4t = L->T4t;
        if (lane < 21) T4[lane + 3] = (uint8_t)(lane < 17 ? TOP(0, lane - 1) : TOP(0, 15));            // corner, top line (the four beyond it are never read: block 5 skips the modes that would)
        else if (lane >= 32 && lane < 48) { const uint8_t v = (uint8_t)LEFT(0, lane - 32); T4t[lane - 32 + 4] = v; T4[(lane - 32 + 1) * 24 + 3] = v; }
        const int half = (lane >> 4) & 1, px = lane & 3, py = (lane >> 2) & 3;
        const qparams q4 = make_q(T, qp, true);
        // The 4x4 transforms run as two-stage butterflies over DPP lane exchanges (partner x^3 then x^1 forward, x^1 then x^3
        // inverse; rows within the quad, columns across the 16-lane row), which leaves the coefficients in the lane order
        // 0, 2, 1, 3 per dimension: lane (px, py) holds frequency (fx, fy) = (F[px], F[py]).  Quantiser class and zig-zag
        // position follow the frequency; the inverse butterfly takes that order and returns samples in natural order.
        const int fx = ((px & 1) << 1) | (px >> 1), fy = ((py & 1) << 1) | (py >> 1);
        // butterfly coefficients per lane position p = 0..3 (x for rows, y for columns):
        //   stage "x^3": own * s1 + partner, s1 = +1 +1 -1 -1            (a0+a3, a1+a2, a1-a2, a0-a3; and the inverse's second stage)
        //   forward "x^1": own * fo + partner * fp, (fo, fp) = (1,1) (-1,1) (1,2) (1,-2)
        //   inverse "x^1": (own >> is) * io + partner * ip, is = 0 0 1 1, (io, ip) = (1,1) (-1,1) (1,-1) (1,1)
        const int sx1 = px < 2 ? 1 : -1, sy1 = py < 2 ? 1 : -1;
        const int fxo = px == 1 ? -1 : 1, fxp = px < 2 ? 1 : px == 2 ? 2 : -2, fyo = py == 1 ? -1 : 1, fyp = py < 2 ? 1 : py == 2 ? 2 : -2;
        const int ixs = px >> 1, ixo = px == 1 ? -1 : 1, ixp = px == 2 ? -1 : 1, iys = py >> 1, iyo = py == 1 ? -1 : 1, iyp = py == 2 ? -1 : 1;
        const int cl4 = (!(fx & 1) && !(fy & 1)) ? 0 : ((fx & 1) && (fy & 1)) ? 1 : 2;
        const int mf4 = cl4 == 0 ? q4.mf[0] : cl4 == 1 ? q4.mf[1] : q4.mf[2], v4 = cl4 == 0 ? q4.v[0] : cl4 == 1 ? q4.v[1] : q4.v[2];
        const int kz4 = (int)((0xFEA9DB83C7426510ull >> (4 * (fy * 4 + fx))) & 15); // raster -> zig-zag position
        // ---- per sub-step, per lane: {predictor-table entry, source sample}, 16 bits each, five registers of two sub-steps
        unsigned pk0 = 0, pk1 = 0, pk2 = 0, pk3 = 0, pk4 = 0;
#pragma unroll
        for (int s4 = 0; s4 < 10; s4++) {
            const int by_lo = s4 > 3 ? (s4 - 2) >> 1 : 0, by_hi = (s4 >> 1) < 3 ? (s4 >> 1) : 3;
            const bool two = by_lo + 1 <= by_hi;
            const int by = (half && two) ? by_lo + 1 : by_lo, bx = s4 - 2 * by;
            const int b = ((by >> 1) << 3) | ((bx >> 1) << 2) | ((by & 1) << 1) | (bx & 1); // blkIdx
            const unsigned mw = b < 4 ? dec0.x : b < 8 ? dec0.y : b < 12 ? dec0.z : dec0.w;
            const unsigned bmode = (mw >> (8 * (b & 3))) & 255u;
            const unsigned ent = T->i4tab[bmode * 16 + py * 4 + px];
            const unsigned sv = S4[(by * 4 + py) * 16 + bx * 4 + px];
            const unsigned e = (ent | (sv << 8)) << (16 * (s4 & 1));
            if (s4 < 2) pk0 |= e; else if (s4 < 4) pk1 |= e; else if (s4 < 6) pk2 |= e; else if (s4 < 8) pk3 |= e; else pk4 |= e;
            if ((lane & 15) == 0 && lane < 32 && (half == 0 || two)) stg16(&ctx->levels[(size_t)mbn * MB_LEVELS + L_LDC + b], (int)bmode); // the mode, for the entropy coder
        }
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
            const unsigned pw = s4 < 2 ? pk0 : s4 < 4 ? pk1 : s4 < 6 ? pk2 : s4 < 8 ? pk3 : pk4;
            const unsigned pe = (pw >> (16 * (s4 & 1))) & 0xFFFFu;
            const int j0 = (int)(pe & 15u), kind = (int)((pe >> 4) & 3u), sv = (int)(pe >> 8);
            // E(i) of 8.3.1.2, i = -4 .. emax (beyond: replicated), see intra_lds: one byte address per tap
            const int A0 = (4 * by) * 24 + 4 * bx + 3, B0 = 17 * 24 + (4 * bx) * 24 + 4 * by + 3;
            const int ia = clip3(-4, emax, j0 - 5), ic = clip3(-4, emax, j0 - 4), id = clip3(-4, emax, j0 - 3);
            const int oa = ia >= 0 ? A0 + ia : B0 - ia, oc = ic >= 0 ? A0 + ic : B0 - ic, od = id >= 0 ? A0 + id : B0 - id;
            hk.before(s4);
            const int za = T4[oa], zc = T4[oc], zd = T4[od];
            const unsigned topw = *(const unsigned *)&T4[A0 + 1], leftw = *(const unsigned *)&T4[B0 + 1];
            const int sumT = (int)__builtin_amdgcn_sad_u8(topw, 0u, 0u), sumL = (int)__builtin_amdgcn_sad_u8(leftw, 0u, 0u);
            const int dcb = (sumT + sumL + 4) >> 3, dct = (sumT + 2) >> 2, dcl = (sumL + 2) >> 2;
            const int dc4 = (up && lf) ? dcb : lf ? dcl : up ? dct : 128;
            // one weighted sum serves copy (4,0,0)/4, 2-tap (2,2,0 | +2)/4 and 3-tap (1,2,1 | +2)/4; selects written as
            // arithmetic on per-lane constants: as ?: chains the compiler turns them into exec-mask branches (~25
            // instructions each on this dependency chain)
            const int w0 = 4 >> kind, w1 = kind ? 2 : 0, w2 = kind >> 1;
            const int bdir = mad24(za, w0, mad24(zc, w1, mad24(zd, w2, w1))) >> 2;
            const int bpred = kind == 3 ? dc4 : bdir;
            // residual -> 4x4 core transform (8.5.12's forward counterpart): rows, then columns
            const int res = sv - bpred;
            int pr = quad_xor<3>(res);
            int tr = mad24(res, sx1, pr);                                  // e0 e1 e2 e3
            pr = quad_xor<1>(tr);
            tr = mad24(tr, fxo, __mul24(pr, fxp));                        // f0 f2 f1 f3
            pr = row_xor12(tr);
            int tc = mad24(tr, sy1, pr);
            pr = row_xor4(tc);
            const int coef = mad24(tc, fyo, __mul24(pr, fyp));
            const int lv4 = quant1(coef, mf4, q4.f, q4.qbits);
            // 8.5.12: scale, inverse transform (rows then columns), round
            const int dq = (lv4 * v4) << q4.shift;
            pr = quad_xor<1>(dq);
            tr = mad24(dq >> ixs, ixo, __mul24(pr, ixp));                 // e0 e1 e2 e3
            pr = quad_xor<3>(tr);
            tr = mad24(tr, sx1, pr);                                      // natural order again
            pr = row_xor4(tr);
            tc = mad24(tr >> iys, iyo, __mul24(pr, iyp));
            pr = row_xor12(tc);
            const int rr = mad24(tc, sy1, pr);
            const int recp = clip255(bpred + ((rr + 32) >> 6));
            if (valid) {
                T4[(by * 4 + py + 1) * 24 + bx * 4 + px + 4] = (uint8_t)recp;
                T4[17 * 24 + (bx * 4 + px + 1) * 24 + by * 4 + py + 4] = (uint8_t)recp;
            }
            WAVE_SYNC();
            hk.after(s4, bx, by, px, py, valid, recp);
            const unsigned long long bal = __ballot(valid && lv4 != 0);
            const int b0 = ((by_lo >> 1) << 3) | (((s4 - 2 * by_lo) >> 1) << 2) | ((by_lo & 1) << 1) | ((s4 - 2 * by_lo) & 1);
            if (bal & 0xFFFFull) nz4 |= 1u << b0;
            if (two) {
                const int by1 = by_lo + 1, bx1 = s4 - 2 * by1, b1 = ((by1 >> 1) << 3) | ((bx1 >> 1) << 2) | ((by1 & 1) << 1) | (bx1 & 1);
                if (bal & 0xFFFF0000ull) nz4 |= 1u << b1;
            }
            if (valid) { // off the chain: nothing waits for these
                if (!SC1) stg8(ry + (size_t)(y0 + by * 4 + py) * stride + x0 + bx * 4 + px, (unsigned)recp);
                stg16(&ctx->levels[(size_t)mbn * MB_LEVELS + L_LUMA + b * 16 + kz4], lv4);
            }
        }
        if (SC1) // the deblocker of the same picture follows the reconstruction while it is written: the whole tile, write-through, a word per lane
            st_sc1((unsigned *)(ry + (size_t)(y0 + (lane >> 2)) * stride + x0 + 4 * (lane & 3)), *(const unsigned *)&T4[((lane >> 2) + 1) * 24 + 4 + 4 * (lane & 3)]);
        if (OUT && lane < 16) { L->bot_y[slot][lane] = T4[16 * 24 + lane + 4]; L->right_y[lane] = T4[17 * 24 + 16 * 24 + lane + 4]; }
    } else if (use_i8 && wave == 0) {
        // ================================================================ Intra_8x8 reconstruction (8.3.2 + 8.5.13; oracle: intra8x8_recon)
        // The four blocks one after another, lane = one sample.  The caller has the whole top line and the corner in L->top (as for Intra_16x16); everything else arrives
        // through the hook: hk.before8(b, R) in front of blocks 0 and 2 leaves rows 8 (b >> 1) .. + 7 of the left neighbour's right column in the tile, in front of
        // block 1 the first eight samples of the macroblock above-right in R[20..27] (zero where that macroblock is not available); hk.after8(b, ...) publishes this macroblock's right column behind blocks 1 and 3 -- so macroblock x + 1
        // runs its upper half while x runs its lower one.  T4 + T4t serve as one tile of stride 32 (17 rows; sample (r, c) at R[(r + 1) * 32 + c + 4])
        // followed by the neighbour / edge arrays; S4 becomes the transform's tile once every lane holds its four source samples.
      if constexpr (I8) {
        uint8_t *R = L->T4;
        int *rt = (int *)(L->T4 + 17 * 32), *rl = rt + 16, *e8 = rt + 24;
        static_assert(17 * 32 + 4 * (24 + 25) <= 2 * 17 * 24, "tile + arrays fit T4 + T4t");
        if (lane < 17) R[lane + 3] = (uint8_t)TOP(0, lane - 1);
        const int x = lane & 7, y = lane >> 3;
        int sv[4];
#pragma unroll
        for (int b = 0; b < 4; b++) sv[b] = (int)S4[((b >> 1) * 8 + y) * 16 + (b & 1) * 8 + x];
        if (lane < 16) stg16(&ctx->levels[(size_t)mbn * MB_LEVELS + L_LDC + lane], lane < 4 ? (int)((dec0.x >> (8 * lane)) & 255u) : 0); // the modes, for the entropy coder
        const bool has_tr = has_top && mx + 1 < mbw;
#pragma unroll 1
        for (int b = 0; b < 4; b++) {
            const int bx8 = (b & 1) * 8, by8 = (b >> 1) * 8, mode = (int)((dec0.x >> (8 * b)) & 255u);
            bool up, lf, ul, ur;
            i8_avail(b, has_top, has_left, has_tr, up, lf, ul, ur);
            hk.before8(b, R);
            WAVE_SYNC();
            const int rc = i8_gather(lane, R, 32, 4, bx8, by8, ur, rt, rl);
            WAVE_SYNC();
            i8_edges(lane, rt, rl, rc, up, lf, ul, e8);
            WAVE_SYNC();
            const int pred = i8_pred_px(e8, mode, x, y, mode == 2 ? i8_dc(e8, up, lf) : 0);
            int rres;
            const unsigned m = i8_tq8(ctx, T, (int *)S4, lane, (b == 0 ? sv[0] : b == 1 ? sv[1] : b == 2 ? sv[2] : sv[3]) - pred, qp, mbn, b, rres);
            const int recp = clip255(pred + rres);
            R[(by8 + y + 1) * 32 + bx8 + x + 4] = (uint8_t)recp;
            hk.after8(b, x, y, recp);
            nz4 |= m << (4 * b);
        }
        nz4 |= NZ_T8; // transform_size_8x8_flag of an I_NxN macroblock is sent whatever its levels
        WAVE_SYNC();
        {
            const unsigned rw = *(const unsigned *)&R[((lane >> 2) + 1) * 32 + 4 + 4 * (lane & 3)];
            if (SC1) st_sc1((unsigned *)(ry + (size_t)(y0 + (lane >> 2)) * stride + x0 + 4 * (lane & 3)), rw);
            else stg32(ry + (size_t)(y0 + (lane >> 2)) * stride + x0 + 4 * (lane & 3), rw);
        }
        if (OUT && lane < 16) { L->bot_y[slot][lane] = R[16 * 32 + lane + 4]; L->right_y[lane] = R[(lane + 1) * 32 + 19]; }
      }
    } else if (wave == 0) {
        // ================================================================ Intra_16x16 reconstruction (8.3.3 + 8.5.10)
        const int bx = lane & 3, by = lane >> 4, mode = mode16, yy = 4 * by + py;
        // neighbour statistics: every 16-lane row reduces the same 16 top / left samples (lane & 15 = j), so each lane ends up
        // with the sums for DC and the weighted sums of 8.3.3.4 for Plane (weights j - 7, and -8 for the corner)
        const int j16 = lane & 15, tj = TOP(0, j16), lj = LEFT(0, j16), cor = TOP(0, -1);
        int pd[4];
        if (mode == 0) { // wave-uniform: only the chosen predictor is evaluated
#pragma unroll
            for (int i = 0; i < 4; i++) pd[i] = TOP(0, 4 * bx + i);
        } else if (mode == 1) {
            const int l = LEFT(0, yy);
            pd[0] = l; pd[1] = l; pd[2] = l; pd[3] = l;
        } else if (mode == 2) {
            const int st = wave16_sum(tj), sl = wave16_sum(lj);
            const int dcv = (has_top && has_left) ? (st + sl + 16) >> 5 : has_top ? (st + 8) >> 4 : has_left ? (sl + 8) >> 4 : 128;
            pd[0] = dcv; pd[1] = dcv; pd[2] = dcv; pd[3] = dcv;
        } else {
            const int Hh = wave16_sum((j16 - 7) * tj) - 8 * cor, Vv = wave16_sum((j16 - 7) * lj) - 8 * cor;
            const int pa = 16 * (LEFT(0, 15) + TOP(0, 15)), pb = (5 * Hh + 32) >> 6, pc = (5 * Vv + 32) >> 6;
#pragma unroll
            for (int i = 0; i < 4; i++) pd[i] = clip255((pa + pb * (4 * bx + i - 7) + pc * (yy - 7) + 16) >> 5);
        }
        const qparams q = make_q(T, qp, true);
        const int fy = ((py & 1) << 1) | (py >> 1);
        const col_bf cb = make_col_bf(py);
        const int kz0 = (int)((0xFEA9DB83C7426510ull >> (16 * fy)) & 0xFFFF); // zig-zag positions of raster 4 fy + 0 .. 3
        const int mfe = py < 2 ? q.mf[0] : q.mf[2], mfo = py < 2 ? q.mf[2] : q.mf[1], ve = py < 2 ? q.v[0] : q.v[2], vo = py < 2 ? q.v[2] : q.v[1];
        int x[4], lev[4], cf[4];
#pragma unroll
        for (int i = 0; i < 4; i++) x[i] = byte_of(srcw, i) - pd[i];
        fwd_rows4(x);
#pragma unroll
        for (int i = 0; i < 4; i++) cf[i] = fwd_col(x[i], cb);
        // 4x4 Hadamard of the 16 DC terms (element 0 of the py == 0 lanes; 8.5.10's forward counterpart): block column = lane
        // bits 1:0, block row = lane bits 5:4, four butterfly stages (two DPP, two cross-row shuffles).  The butterflies give
        // the natural-ordered transform (H2 x H2 per dimension); with M = rows (++++, ++--, +--+, +-+-) of the standard,
        // natural index k holds M-frequency G[k], G = 0 3 1 2, so this lane's coefficient belongs at raster position
        // (G[by], G[bx]).  M is symmetric: the same stages on the quantised levels return this lane's own block's term.
        const int s0 = (lane & 1) ? -1 : 1, s1 = (lane & 2) ? -1 : 1, s2 = (lane & 16) ? -1 : 1, s3 = (lane & 32) ? -1 : 1;
        int hv = cf[0];
        hv = mad24(hv, s0, quad_xor<1>(hv));
        hv = mad24(hv, s1, quad_xor<2>(hv));
        hv = mad24(hv, s2, __shfl_xor(hv, 16, 64));
        hv = mad24(hv, s3, __shfl_xor(hv, 32, 64));
        int ldc = quant1((hv + 1) >> 1, q.mf[0], 2 * q.f, q.qbits + 1);
        const bool dcl = py == 0; // this lane holds a DC term
#pragma unroll
        for (int i = 0; i < 4; i++) lev[i] = (i == 0 && dcl) ? 0 : quant1(cf[i], (i & 1) ? mfo : mfe, q.f, q.qbits);
        if (ctx->iac_drop) { // rate control's ladder for I pictures: luma levels (DC and AC) summing to no more than the threshold are not sent
            const int sm = wave64_sum((dcl ? iabs(ldc) : 0) + iabs(lev[0]) + iabs(lev[1]) + iabs(lev[2]) + iabs(lev[3]));
            if (sm <= ctx->iac_drop) { ldc = 0; lev[0] = lev[1] = lev[2] = lev[3] = 0; }
        }
        int f = ldc;
        f = mad24(f, s0, quad_xor<1>(f));
        f = mad24(f, s1, quad_xor<2>(f));
        f = mad24(f, s2, __shfl_xor(f, 16, 64));
        f = mad24(f, s3, __shfl_xor(f, 32, 64));
        const int ls = 16 * q.v[0];
        const int dcv2 = qp >= 36 ? (f * ls) << (qp / 6 - 6) : (f * ls + (1 << (5 - qp / 6))) >> (6 - qp / 6);
#pragma unroll
        for (int i = 0; i < 4; i++) x[i] = (lev[i] * ((i & 1) ? vo : ve)) << q.shift;
        if (dcl) x[0] = dcv2;
        const int b = ((by >> 1) << 3) | ((bx >> 1) << 2) | ((by & 1) << 1) | (bx & 1);
        int16_t *lv = ctx->levels + (size_t)mbn * MB_LEVELS;
#pragma unroll
        for (int i = 0; i < 4; i++) stg16(&lv[L_LUMA + b * 16 + ((kz0 >> (4 * i)) & 15)], lev[i]);
        if (dcl) {
            const int gx = (0x2130 >> (4 * bx)) & 3, gy = (0x2130 >> (4 * by)) & 3; // G[bx], G[by]
            stg16(&lv[L_LDC + (int)((0xFEA9DB83C7426510ull >> (4 * (gy * 4 + gx))) & 15)], ldc);
        }
        inv_rows4(x);
        int o[4];
#pragma unroll
        for (int i = 0; i < 4; i++) o[i] = clip255(pd[i] + ((inv_col(x[i], cb) + 32) >> 6));
        const unsigned rw = pack4(o[0], o[1], o[2], o[3]);
        if (SC1) st_sc1((unsigned *)(ry + (size_t)(y0 + yy) * stride + x0 + 4 * bx), rw);
        else stg32(ry + (size_t)(y0 + yy) * stride + x0 + 4 * bx, rw);
        if (OUT) {
            if (yy == 15) *(unsigned *)&L->bot_y[slot][4 * bx] = rw;
            if (bx == 3) L->right_y[yy] = (uint8_t)(rw >> 24);
        }
        const unsigned long long bal = __ballot((lev[0] | lev[1] | lev[2] | lev[3]) != 0);
        const unsigned long long t = bal | (bal >> 4) | (bal >> 8) | (bal >> 12); // bit 16 by + bx
        const int rb = lane & 15, rbx = blkx(rb) >> 2, rby = blky(rb) >> 2;
        nz16 = (unsigned)(__ballot(lane < 16 && ((t >> (16 * rby + rbx)) & 1)) & 0xFFFFull);
        ldc_any = __ballot(dcl && ldc != 0) != 0 ? 1u : 0u;
    }
    if (wave == 1) { // ================================================================ chroma (8.3.4 + 8.5.11)
        const int cby = (lane >> 4) & 1, c = (lane >> 1) & 1, cbx = lane & 1, p = 1 + c;
        const int bx = 4 * cbx, by = 4 * cby, yy = by + py, b = 2 * cby + cbx;
        int pd[4];
        if (cmode == 0) { // wave-uniform: only the chosen predictor is evaluated.  DC of this lane's 4x4 block (8.3.4.1-3)
            int st = 0, sl = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) { st += TOP(p, bx + i); sl += LEFT(p, by + i); }
            bool ut = has_top, ul = has_left;
            if (b == 1 && has_top) ul = false;
            if (b == 2 && has_left) ut = false;
            const int dcv = (ut && ul) ? (st + sl + 4) >> 3 : ut ? (st + 2) >> 2 : ul ? (sl + 2) >> 2 : 128;
            pd[0] = dcv; pd[1] = dcv; pd[2] = dcv; pd[3] = dcv;
        } else if (cmode == 1) {
            const int l = LEFT(p, yy);
            pd[0] = l; pd[1] = l; pd[2] = l; pd[3] = l;
        } else if (cmode == 2) {
#pragma unroll
            for (int i = 0; i < 4; i++) pd[i] = TOP(p, bx + i);
        } else {
            int Hh = 0, Vv = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                Hh += (i + 1) * (TOP(p, 4 + i) - TOP(p, 2 - i));
                Vv += (i + 1) * (LEFT(p, 4 + i) - LEFT(p, 2 - i));
            }
            const int pa = 16 * (LEFT(p, 7) + TOP(p, 7)), pb = (34 * Hh + 32) >> 6, pc = (34 * Vv + 32) >> 6;
#pragma unroll
            for (int i = 0; i < 4; i++) pd[i] = clip255((pa + pb * (bx + i - 3) + pc * (yy - 3) + 16) >> 5);
        }
        chroma_rows4(ctx, T, ctx->levels + (size_t)mbn * MB_LEVELS, cx0, cy0, lane, pd, csv, qp, true, true, OUT ? L->crec : nullptr, cnz8, cdc2, false, ctx->iac_drop, SC1);
    }
#undef TOP
#undef LEFT
    if (HK::own_record) {
        if (wave == 1) {
            if (OUT) {
                WAVE_SYNC();
                if (lane >= 16 && lane < 32) {
                    const int i = lane - 16;
                    L->bot_c[slot][i] = L->crec[7 * 16 + i];
                    L->right_c[i >> 3][i & 7] = L->crec[(i & 7) * 16 + 14 + (i >> 3)];
                }
            }
            hk.chroma_done(cnz8, cdc2);
        } else hk.luma_done(itype ? nz4 : nz16, !itype && ldc_any != 0);
        return;
    }
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
            L->cflags[0] = cnz8; L->cflags[1] = cdc2;
            __hip_atomic_store(&L->cseq, (unsigned)mbn + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    } else if (lane == 0) { // the luma wave writes the record once the chroma wave's flags are in (both waves are resident: plain spin)
        while (__hip_atomic_load(&L->cseq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != (unsigned)mbn + 1u) __builtin_amdgcn_s_sleep(1);
        const unsigned cany = L->cflags[0], cdc = L->cflags[1];
        unsigned nzm = (itype ? nz4 : nz16) | (cany << 16);
        if (!itype && ldc_any) nzm |= NZ_LDC;
        if (cdc & 1) nzm |= NZ_CBDC;
        if (cdc & 2) nzm |= NZ_CRDC;
        mb_info_t mb;
        mb.mvx = 0; mb.mvy = 0; mb.mb_type = itype ? 2 : 0; mb.i16_mode = itype ? 0 : (uint8_t)mode16; mb.chroma_mode = (uint8_t)cmode;
        mb.qp = (uint8_t)qp; mb.nzmask = nzm; mb.cost = dec1.y;
        st_mbinfo(&ctx->mbi[mbn], mb);
    }
}

template <bool OUT, bool SC1 = false>
DEV void intra_compute(const frame_ctx_t *__restrict__ ctx, const dev_tables *T, intra_lds *L, const int mx, const int my, const int wave, const int lane,
                       const uint4 dec0, const uint2 dec1, const uint2 *presrc = nullptr) {
    ic_nohook hk;
    intra_compute<OUT, SC1, ic_nohook>(ctx, T, L, mx, my, wave, lane, dec0, dec1, presrc, hk);
}


// =================================================================== intra macroblocks of P pictures
// pmb_kernel has reconstructed every inter macroblock and left type + modes in the records of the ones it decided to code
// intra (Intra_16x16 only); those predict from their neighbours' reconstructed samples (8.3; constrained_intra_pred_flag = 0).
// One workgroup (luma wave + chroma wave, as everywhere in this file) per macroblock ROW walks the row's intra
// macroblocks from left to right:
//  * the left neighbour, when it is intra itself, was this workgroup's previous macroblock: its right column is still in LDS;
//  * the row above publishes, per macroblock it finishes, a 32-byte strip (bottom luma line, bottom chroma line) with `sc1`
//    stores and then a progress word = "every macroblock left of this column is final" (epoch in the upper bits, so nothing is
//    ever cleared); a macroblock whose top or top-left neighbour is intra polls that word and reads the strips with `sc1`
//    loads (MI355X_MICROARCH.md, "Valid forms": sc1 stores, vmcnt(0), barrier, sc1 flag / sc1 poll, sc1 loads);
//  * every other neighbour sample comes from the picture: inter macroblocks were final before this launch.
// Rows only ever wait for the row above, so the wait graph is acyclic whatever the dispatch order (every spin is bounded and
// reports through `err`).  A row without intra macroblocks publishes "done" and leaves.  Oracle: orc_intra_p_frame.
struct ip_args { frame_ctx_t ctx; unsigned *progress; uint8_t *strips; unsigned *err; };
#define IP_EPOCH(e) (((e) & 0xFFFFFu) << 12)
// One macroblock row, run by threads 0..127 of the calling workgroup (luma wave, chroma wave; any other wave of the workgroup must have ended:
// the barriers below then count these two).  row_done / row_need (may be null): the fused P stage of the same picture may still be running;
// the row starts when its own macroblocks and those of the row above are complete (pmb_kernel<GATED, ROWS> counts them).
DEV void intra_p_row(const ip_args &a, const int my, const unsigned *row_done, const unsigned row_need) {
    const frame_ctx_t *__restrict__ ctx = &a.ctx;
    if (my == 0) tl_first(ctx, 5);
    __shared__ intra_lds LD;
    __shared__ unsigned tabw[TAB_DWORDS];
    __shared__ unsigned ibits[16]; // which macroblocks of this row are intra (mbw <= 512)
    __shared__ int sh_bad;
    const dev_tables *T = (const dev_tables *)tabw;
    const int mbw = ctx->mbw, stride = ctx->stride;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); // an SGPR: what is derived from it is scalar control flow
    const uint8_t *__restrict__ ry = ctx->rec_y;
    const uint8_t *__restrict__ ruv = ctx->rec_uv;
    const unsigned ep = IP_EPOCH(ctx->epoch);
    if (threadIdx.x < 16) ibits[threadIdx.x] = 0;
    if (threadIdx.x == 0) sh_bad = 0;
    if (row_done) { // the records of this row say which macroblocks are intra; the samples of this row and the row above are what they predict from
        if (threadIdx.x < 64) {
            const int r = my - 1 + (int)threadIdx.x;
            const bool mine = threadIdx.x < 2 && r >= 0;
            const unsigned *w = row_done + (mine ? r * MI355_PROG_STRIDE : 0);
            int spins = 0;
            while (__ballot(mine && (int)(ld_sc1(w) - row_need) < 0)) {
                __builtin_amdgcn_s_sleep(8);
                if (++spins > DB_SPIN_MAX) { st_sc1(a.err, 21u); break; }
                if ((spins & 1023) == 0 && ld_sc1(a.err)) break;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    __syncthreads();
    for (int x = threadIdx.x; x < ((mbw + 63) & ~63); x += 128) {
        const bool in = x < mbw && (ldg32(&ctx->mbi[my * mbw + x].mb_type) & 255u) != 1u;
        const unsigned long long b = __ballot(in);
        if (lane == 0) { ibits[(x >> 5)] = (unsigned)b; ibits[(x >> 5) + 1] = (unsigned)(b >> 32); }
    }
    for (int i = threadIdx.x; i < TAB_DWORDS; i += 128) tabw[i] = ((const unsigned *)&g_tab)[i];
    __syncthreads();
    if (threadIdx.x == 0) { // everything left of the row's first intra macroblock is final already (pmb_kernel): the deblocker may start on it
        int nx = mbw;
        for (int w2 = (mbw + 31) / 32 - 1; w2 >= 0; w2--) if (ibits[w2]) nx = 32 * w2 + __builtin_ctz(ibits[w2]);
        if (nx < mbw) st_sc1(&a.progress[my * MI355_PROG_STRIDE], ep | (unsigned)nx);
    }
    int prev_x = -2; // the macroblock this workgroup reconstructed last (its right column is in LD)
    for (int w = 0; w < (mbw + 31) / 32; w++) {
        unsigned bits = ibits[w];
        while (bits) {
            const int mx = 32 * w + __builtin_ctz(bits);
            bits &= bits - 1;
            const int mbn = my * mbw + mx, x0 = mx * 16, y0 = my * 16, cx0 = x0 >> 1, cy0 = y0 >> 1;
            const bool has_top = row_has_top(ctx, my), has_left = mx > 0; // (a P slice's first row: nothing above it is available, 6.4.8)
            const bool b_intra = has_top && (ldg32(&ctx->mbi[mbn - mbw].mb_type) & 255u) != 1u;
            const bool d_intra = has_top && has_left && (ldg32(&ctx->mbi[mbn - mbw - 1].mb_type) & 255u) != 1u;
            const bool a_intra = has_left && prev_x == mx - 1;
            __syncthreads(); // the previous macroblock is done with LD.top / LD.left (its right column and bottom lines stay)
            if (threadIdx.x == 0) {
                LD.cseq = 0;
                if (b_intra || d_intra) { // the row above must have passed column mx
                    int spins = 0;
                    for (;;) {
                        const unsigned v = ld_sc1(&a.progress[(my - 1) * MI355_PROG_STRIDE]);
                        if ((v & ~0xFFFu) == ep && (int)(v & 0xFFFu) > mx) break;
                        __builtin_amdgcn_s_sleep(1);
                        if (++spins > DB_SPIN_MAX) { st_sc1(a.err, 15u); sh_bad = 1; break; } if ((spins & 1023) == 0 && ld_sc1(a.err)) { sh_bad = 1; break; }
                    }
                }
            }
            __syncthreads();
            if (sh_bad) return; // uniform: the error word is set, the host reports it
            const uint4 dec0 = ldg128(ctx->idec + (size_t)mbn * IDEC_BYTES); // the sixteen Intra_4x4 modes (ctx->intra_p == 2; zeros otherwise)
            const uint2 dec1 = ldg64(ctx->idec + (size_t)mbn * IDEC_BYTES + 16);
            // neighbour samples into LD.top / LD.left ([plane][i + 1] = sample i, [0] = corner)
            if (wave == 0 && lane < 17) { // luma top line + corner
                const int i = lane - 1;
                int v = 0;
                if (has_top && (i >= 0 || has_left)) {
                    const bool from_strip = i >= 0 ? b_intra : d_intra;
                    if (from_strip) v = (int)(ld_sc1((const unsigned *)(a.strips + (size_t)(mbn - mbw + (i >= 0 ? 0 : -1)) * 32) + ((i >= 0 ? i : 15) >> 2)) >> (8 * ((i >= 0 ? i : 15) & 3))) & 255;
                    else v = (int)ldg8(ry + (size_t)(y0 - 1) * stride + x0 + i);
                }
                LD.top[0][i + 1] = v;
            } else if (wave == 0 && lane >= 32 && lane < 48) { // luma left column
                const int i = lane - 32;
                LD.left[0][i + 1] = !has_left ? 0 : a_intra ? (int)LD.right_y[i] : (int)ldg8(ry + (size_t)(y0 + i) * stride + x0 - 1);
            } else if (wave == 1 && lane < 18) { // chroma top lines + corners
                const int c = lane / 9, i = lane % 9 - 1;
                int v = 0;
                if (has_top && (i >= 0 || has_left)) {
                    const bool from_strip = i >= 0 ? b_intra : d_intra;
                    const int k = 2 * (i >= 0 ? i : 7) + c; // byte of the interleaved 16-byte bottom chroma line
                    if (from_strip) v = (int)(ld_sc1((const unsigned *)(a.strips + (size_t)(mbn - mbw + (i >= 0 ? 0 : -1)) * 32 + 16) + (k >> 2)) >> (8 * (k & 3))) & 255;
                    else v = (int)ldg8(ruv + (size_t)(cy0 - 1) * stride + 2 * (cx0 + i) + c);
                }
                LD.top[1 + c][i + 1] = v;
            } else if (wave == 1 && lane >= 32 && lane < 48) { // chroma left columns
                const int c = (lane - 32) >> 3, i = (lane - 32) & 7;
                LD.left[1 + c][i + 1] = !has_left ? 0 : a_intra ? (int)LD.right_c[c][i] : (int)ldg8(ruv + (size_t)(cy0 + i) * stride + 2 * (cx0 - 1) + c);
            }
            __syncthreads();
            if (threadIdx.x == 0) { LD.left[0][0] = LD.top[0][0]; LD.left[1][0] = LD.top[1][0]; LD.left[2][0] = LD.top[2][0]; }
            __syncthreads();
            intra_compute<true, true>(ctx, T, &LD, mx, my, wave, lane, dec0, dec1);
            __syncthreads(); // bot_y / bot_c / right_* of this macroblock are in LD
            // publish the bottom lines for the row below: 32 bytes, sc1
            if (threadIdx.x < 8) {
                const int slot = mx & 3;
                const unsigned v = threadIdx.x < 4 ? *(const unsigned *)&LD.bot_y[slot][4 * threadIdx.x] : *(const unsigned *)&LD.bot_c[slot][4 * (threadIdx.x - 4)];
                st_sc1((unsigned *)(a.strips + (size_t)mbn * 32) + threadIdx.x, v);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
#ifdef DBG_DELAY_IP /* adversarial-schedule build: every progress word comes ~20 us late, so the deblocker that follows this kernel catches up with it at every intra macroblock */
            for (int i = 0; i < 6; i++) __builtin_amdgcn_s_sleep(127);
#endif
            if (threadIdx.x == 0) { // everything left of the next intra macroblock of this row (or the whole row) is final now
                unsigned rest = bits;
                int nx = rest ? 32 * w + __builtin_ctz(rest) : -1;
                for (int w2 = w + 1; nx < 0 && w2 < (mbw + 31) / 32; w2++) if (ibits[w2]) nx = 32 * w2 + __builtin_ctz(ibits[w2]);
                st_sc1(&a.progress[my * MI355_PROG_STRIDE], ep | (unsigned)(nx < 0 ? mbw : nx));
            }
            prev_x = mx;
        }
    }
    if (prev_x == -2 && threadIdx.x == 0) st_sc1(&a.progress[my * MI355_PROG_STRIDE], ep | (unsigned)mbw); // no intra macroblock in this row
    tl_last(ctx, 6);
}
#endif
