// k_intra.hip -- I pictures: open-loop analysis + decisions, reconstruction wavefront (persistent bands / per diagonal)
// Hand-written HIP for gfx950 (CDNA4, wave64); part of libmi355enc (see kernels_common.hpp).
#include "intra_mb.hpp"

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
// P pictures (gate_p): only macroblocks whose whole-sample search cost reaches INTRA_GATE are analysed -- the fused P stage
// never considers the others for intra (oracle: orc_pmb_frame, step 4).
// one macroblock on one wave (wave = its slot in the workgroup's shared arrays); ok: the macroblock exists (stores are made)
template <bool GATE_P>
DEV void intra_analyse_mb(const frame_ctx_t *__restrict__ ctx, int mbn, const bool ok, const int lane, const int wave) {
    constexpr bool gate_p = GATE_P; // P picture: the macroblocks at or above the gate (Intra_4x4 only with ctx->intra_p == 2, no Intra_8x8)
    __shared__ uint16_t sh_sad[4][ISAD_PER_MB]; // this wave's macroblock: the same 152 values that go to ctx->isad
    __shared__ int sh_m4[4][16];                // Intra_4x4 modes chosen so far, raster order
    __shared__ int sh_i8[4][16 + 8 + 8 + 32];   // Intra_8x8: raw samples above (16) / to the left (8) / above-right of the macroblock (8), the filtered edge array (25)
    __shared__ __attribute__((aligned(4))) uint8_t SL[4][17 * IA_S];
    __shared__ __attribute__((aligned(4))) uint8_t SC[4][2][9 * 12]; // [plane][row 0 = y -1][col 0 = x -1]
    const int mbw = ctx->mbw;
    const int my = mbn / mbw, mx = mbn - my * mbw, x0 = mx * 16, y0 = my * 16, cx0 = x0 >> 1, cy0 = y0 >> 1;
    const bool has_top = row_has_top(ctx, my), has_left = mx > 0;
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
    // ---- Intra_4x4: four blocks at a time, 16 lanes (pixels) each (P pictures: for the gated macroblocks, when ctx->intra_p == 2)
    const bool do4 = !gate_p || ctx->intra_p == 2;
    if (do4) {
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
        if (ctx->i4x4 && do4) { // Intra_4x4 modes block by block: SAD + lambda * (mode == expected ? 1 : 4); blocks visited along bx + 2*by
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
        // ---- Intra_8x8 (High profile, I pictures): open loop like Intra_4x4 -- filtered SOURCE neighbours, SAD + lambda * (mode == expected ? 1 : 4) per 8x8 block,
        // the macroblock taken when the total + 10 lambda is strictly below what stands (oracle: orc_intra_decide8)
        unsigned m8w = 0;
        bool use_i8 = false;
        if (ctx->i8 && !gate_p && !ctx->iac_drop) {
            int *rt = sh_i8[wave], *rl = rt + 16, *tr = rt + 24, *e8 = rt + 32;
            const bool has_tr = has_top && mx + 1 < mbw;
            if (lane < 8) { int yt = y0 - 1; yt = yt < vh ? yt : vh - 1; tr[lane] = has_tr ? (int)ldg8(sy + (size_t)yt * ss + x0 + 16 + lane) : 0; }
            unsigned cost8 = 0;
            int m8[4] = {2, 2, 2, 2};
#pragma unroll 1
            for (int b = 0; b < 4; b++) {
                const int bx8 = (b & 1) * 8, by8 = (b >> 1) * 8;
                bool up, lf, ul, ur;
                i8_avail(b, has_top, has_left, has_tr, up, lf, ul, ur);
                WAVE_SYNC();
                if (lane < 16) { const int col = bx8 + ((lane >= 8 && !ur) ? 7 : lane); rt[lane] = col < 16 ? (int)S[by8 * IA_S + col + 1] : tr[col - 16]; }
                else if (lane < 24) rl[lane - 16] = (int)S[(by8 + lane - 16 + 1) * IA_S + bx8];
                const int rc = (int)S[by8 * IA_S + bx8];
                WAVE_SYNC();
                i8_edges(lane, rt, rl, rc, up, lf, ul, e8);
                WAVE_SYNC();
                const int dcv = i8_dc(e8, up, lf), x = lane & 7, y = lane >> 3;
                const int sv = (int)S[(by8 + y + 1) * IA_S + bx8 + x + 1];
                const int ma = (b & 1) ? m8[b - 1] : (has_left ? 2 : -1), mb_ = (b >> 1) ? m8[b - 2] : (has_top ? 2 : -1);
                const int pm = (ma < 0 || mb_ < 0) ? 2 : (ma < mb_ ? ma : mb_);
                unsigned best = 0xFFFFFFFFu; int bm = 2;
#pragma unroll 1
                for (int md = 0; md < 9; md++) {
                    if (!i8_mode_ok(md, up, lf, ul)) continue; // (wave-uniform)
                    const unsigned sad = (unsigned)wave64_sum(iabs(sv - i8_pred_px(e8, md, x, y, dcv)));
                    const unsigned cost = sad + (unsigned)(lam * (md == pm ? 1 : 4));
                    if (cost < best) { best = cost; bm = md; }
                }
                m8[b] = bm; cost8 += best;
            }
            if (cost8 + (unsigned)(10 * lam) < cost_luma) {
                use_i8 = true; cost_luma = cost8 + (unsigned)(10 * lam);
                m8w = (unsigned)m8[0] | ((unsigned)m8[1] << 8) | ((unsigned)m8[2] << 16) | ((unsigned)m8[3] << 24);
            }
        }
        if (lane < 8 && ok) { // 32-byte record {u8 modes4[16] by blkIdx (Intra_8x8: four modes, then zeros); u8 mode16, cmode, use_i4 (2: Intra_8x8), 0; u32 cost, cost_luma, 0}
            unsigned w = 0;
            if (use_i8 && lane < 5) w = lane == 0 ? m8w : lane == 4 ? ((unsigned)mode16 | ((unsigned)cmode << 8) | (2u << 16)) : 0u;
            else if (lane < 4) {
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int bb = lane * 4 + i, r = (blky(bb) >> 2) * 4 + (blkx(bb) >> 2);
                    w |= ((ctx->i4x4 && do4) ? (unsigned)sh_m4[wave][r] & 0xFF : 0u) << (8 * i);
                }
            } else if (lane == 4) w = (unsigned)mode16 | ((unsigned)cmode << 8) | ((use_i4 ? 1u : 0u) << 16);
            else if (lane == 5) w = cost_luma + costc;
            else if (lane == 6) w = cost_luma;
            stg32(ctx->idec + (size_t)mbn * IDEC_BYTES + 4 * lane, w);
        }
    }
}

// I pictures: every macroblock, one wave each.
__global__ __launch_bounds__(256) void intra_analyse_kernel(const frame_ctx_t cv) {
    const frame_ctx_t *__restrict__ ctx = &cv;
    const int nmb = ctx->mbw * ctx->mbh;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); // an SGPR: what is derived from it is scalar control flow
    const int mbn = blockIdx.x * 4 + wave;
    intra_analyse_mb<false>(ctx, mbn < nmb ? mbn : nmb - 1, mbn < nmb, lane, wave);
}
// P pictures: only the macroblocks whose search cost reaches the gate; the others leave at once.  (r04 tried a wave per EIGHT macroblocks, analysing the gated ones one after
// the other, as the later selection passes do: 5 % FEWER frames/s at 1080p, +-0 at 2160p -- an analysis is a microsecond or two of work, and a wave with three of them in a row
// makes the launch, which sits between the selection and the fused stage, longer than the 8 160 waves it saves are worth.)
__global__ __launch_bounds__(256) void intra_analyse_gated_kernel(const frame_ctx_t cv) {
    const frame_ctx_t *__restrict__ ctx = &cv;
    const int nmb = ctx->mbw * ctx->mbh;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int mbn = blockIdx.x * 4 + wave;
    if (mbn >= nmb) return;
    const uint2 iv = ldg64(k_final_imv_dev(ctx) + mbn);
    if ((iv.y & 0xFFFFu) + (unsigned)ctx->lambda * (iv.y >> 16) < INTRA_GATE(ctx->lambda)) return;
    intra_analyse_mb<true>(ctx, mbn, true, lane, wave);
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
    const bool has_top = row_has_top(ctx, my), has_left = mx > 0;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); // an SGPR: what is derived from it is scalar control flow
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
#ifndef IB_ROWS
#define IB_ROWS 2 /* rows per band: four waves, one per SIMD.  With four rows two compute-heavy waves shared every SIMD and a step cost their sum
                     (an Intra_16x16-only 1080p picture 0.51 ms, 0.44 with two rows; one row 0.49, three 0.51): since the bottom lines
                     between bands travel as granules a band boundary costs little more than a row boundary inside a band */
#endif
struct ib_args { frame_ctx_t ctx; uint2 *gran; unsigned *err; unsigned *band_done; }; // gran: the bottom lines between bands, 8 granules {4 samples, tag} per macroblock and boundary

template <bool PAIR> // which waves share a SIMD, see below
__global__ __launch_bounds__(IB_ROWS * 128) void intra_band_kernel(ib_args a) {
    __shared__ intra_lds LD[IB_ROWS];
    __shared__ unsigned tabw[TAB_DWORDS];
    __shared__ unsigned stage[2][8]; // first row of a band: the prefetched samples of the band above (luma, chroma), 5 dwords each
    const dev_tables *T = (const dev_tables *)tabw;
    const frame_ctx_t *__restrict__ ctx = &a.ctx;
    const int mbw = ctx->mbw, mbh = ctx->mbh;
    const int band = blockIdx.x, w = threadIdx.x >> 6, lane = threadIdx.x & 63, r = PAIR ? w % IB_ROWS : w >> 1, role = PAIR ? w / IB_ROWS : w & 1; // (w through readfirstlane, i.e. scalar control flow, made this kernel 14 % SLOWER: measured, left as it is)
    // (waves go to SIMD w % 4.  With Intra_4x4 in play a luma wave is 3-4x a chroma wave: every SIMD gets one row's luma and another row's
    // chroma wave, 1.29 -> 1.20 ms per picture.  Intra_16x16 only (the rate-control ladder's pictures): the two are equal, and the pairing
    // luma+luma / chroma+chroma was measured 5 % faster)
    const int my = band * IB_ROWS + r;
    const bool row_ok = my < mbh, has_top = row_has_top(ctx, my);
    const bool fed = row_ok && r == 0 && band > 0;                        // top samples come from the band above
    const bool feeds = row_ok && r == IB_ROWS - 1 && my != mbh - 1;       // bottom rows go to the band below
    intra_lds *L = &LD[r];
    const intra_lds *Lup = &LD[r > 0 ? r - 1 : 0];
    for (int i = threadIdx.x; i < TAB_DWORDS; i += IB_ROWS * 128) tabw[i] = ((const unsigned *)&g_tab)[i];
    if (lane == 0 && role == 0) L->cseq = 0;
    // Between bands the bottom lines travel as 8-byte granules {4 samples, tag}, one sc1 store each, polled directly by the band below
    // (MI355X_MICROARCH.md hand-off R2): no drain on the producer, no counter.  The tag is the picture's epoch inverted -- the band
    // deblocker of the same picture uses the same buffer with the plain epoch.
    const unsigned tag = ~ctx->epoch;
    uint2 *gran_up = a.gran + (size_t)(band > 0 ? band - 1 : 0) * mbw * 8 + 4 * role, *gran_my = a.gran + (size_t)band * mbw * 8 + 4 * role;
    // the lanes that move neighbour samples: luma wave 24..40 (i = -1..15), chroma wave 41..58 (plane c, i = -1..7)
    const bool mover = role == 0 ? (lane >= 24 && lane < 41) : (lane >= 41 && lane < 59);
    const int mi = role == 0 ? lane - 25 : (lane - 41) % 9 - 1, mc = role == 0 ? 0 : (lane - 41) / 9;
    // ... and the lanes that fetch for a fed row: 5 dwords starting 4 bytes left of the macroblock (the corner is byte 3 of dword 0)
    uint2 gpre = make_uint2(0, 0);
    uint4 dec0n = make_uint4(0, 0, 0, 0);
    uint2 dec1n = make_uint2(0, 0);
    // this lane's source samples of the next macroblock (layout: intra_compute).  Two variables, not one filled by an if / else: with
    // one, the luma side's address arithmetic reuses the registers the (predicated-off) chroma load writes, and the compiler guards
    // that with an s_waitcnt vmcnt(0) -- issued right after the decision loads, i.e. a full memory latency on the chain.
    unsigned srcn_y = 0;
    uint2 srcn_c = make_uint2(0, 0);
    const int spy = (lane >> 2) & 3;
    const int nsteps = mbw + IB_ROWS + 1;
#ifdef IB_PROF /* debug builds: cycles inside intra_compute per wave, and of the whole loop, left in ctx->isad */
    unsigned long long ib_cyc = 0, ib_n = 0;
    const unsigned long long ib_l0 = __builtin_readcyclecounter();
#endif
    for (int t = -1; t < nsteps; t++) { // step -1 only prefetches for the first row
        const int x = t - r, xn = x + 1;
        const bool act = row_ok && x >= 0 && x < mbw;
        __syncthreads(); // the rings and right columns written in the previous step are visible
        // ---- land what was prefetched for this step, prefetch for the next one
        const uint4 dec0 = dec0n;
        const uint2 dec1 = dec1n;
        const uint2 srcc = role == 0 ? make_uint2(srcn_y, 0u) : srcn_c;
        if (fed && act) { // lanes 24..28: the corner (last word of macroblock x-1's line) and the four words of macroblock x's line, asked for a step ago
            const bool mine = lane >= 24 && lane < 29 && (x > 0 || lane > 24);
            if (__ballot(mine && gpre.y != tag)) { // not there yet: poll (bounded)
                int spins = 0;
                do {
                    __builtin_amdgcn_s_sleep(1);
                    if (mine) gpre = ld64_sc1(gran_up + (size_t)(x - (lane == 24 ? 1 : 0)) * 8 + (lane == 24 ? 3 : lane - 25));
                    if (++spins > DB_SPIN_MAX) { st_sc1(a.err, 14u); break; } if ((spins & 1023) == 0 && ld_sc1(a.err)) { break; }
                } while (__ballot(mine && gpre.y != tag));
            }
            if (lane >= 24 && lane < 29) stage[role][lane - 24] = mine ? gpre.x : 0u;
        }
        // Every load below is issued unconditionally, from an address clamped into range (results of steps that have no next
        // macroblock are never read).  As conditional assignments to loop-carried values they would need a merge with the old value
        // after the load -- a move into the register the load is still filling, which the compiler guards with s_waitcnt
        // vmcnt: a full memory latency right after issuing, every step, on the luma wave (~1800 cycles; per-phase cycle
        // counters of an -DIB_PROF build).
        {
            const int xc = xn < 0 ? 0 : (xn < mbw ? xn : mbw - 1), myc = row_ok ? my : mbh - 1;
            const size_t mbn_n = (size_t)myc * mbw + xc;
            dec0n = ldg128(ctx->idec + mbn_n * IDEC_BYTES);
            dec1n = ldg64(ctx->idec + mbn_n * IDEC_BYTES + 16);
            int sy = myc * 16 + 4 * (lane >> 4) + spy;
            sy = sy < ctx->vis_h ? sy : ctx->vis_h - 1;
            srcn_y = ldg32(ctx->src_y + (size_t)sy * ctx->src_stride + xc * 16 + 4 * (lane & 3));
            const int vh2 = ctx->vis_h >> 1;
            int cy = myc * 8 + 4 * ((lane >> 4) & 1) + spy;
            cy = cy < vh2 ? cy : vh2 - 1;
            srcn_c = ldg64(ctx->src_uv + (size_t)cy * ctx->src_stride + 2 * (xc * 8 + 4 * (lane & 1)));
            const int fl = lane < 24 ? 0 : (lane < 29 ? lane - 24 : 4);                        // 0: the corner, 1..4: the line's words
            const int gx = fl == 0 ? (xc > 0 ? xc - 1 : 0) : xc;
            gpre = ld64_sc1(gran_up + (size_t)gx * 8 + (fl == 0 ? 3 : fl - 1)); // every row loads (rows that are not fed never look at it)
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
            intra_compute<true>(ctx, T, L, x, my, role, lane, dec0, dec1, &srcc);
#ifdef IB_PROF
            ib_cyc += __builtin_readcyclecounter() - ib_t0; ib_n++;
#endif
            // ---- last row of the band: its bottom rows go to the band below
            if (feeds) {
                WAVE_SYNC();
                if (lane < 4) st64_sc1(gran_my + (size_t)x * 8 + lane, make_uint2(role == 0 ? ((const unsigned *)L->bot_y[x & 3])[lane] : ((const unsigned *)L->bot_c[x & 3])[lane], tag));
            }
        }
    }
    // ---- this band's reconstruction and records are complete: tell the band deblocker, which may be waiting on another stream
    // (release pattern of MI355X_MICROARCH.md: every storing wave drains, the workgroup meets, one lane writes this XCD's L2 back and
    // only then publishes the tagged flag)
    if (a.band_done) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            st_sc1(a.band_done + band, tag);
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
int k_intra_band_rows(void) { return IB_ROWS; }
// d_gran: 8 granules per macroblock and band boundary (a buffer of its own: the band deblocker of the PREVIOUS picture may still be running)
// d_band_done (may be null): one word per band, set to the inverted epoch once the band is complete in memory (the band deblocker's gate)
void k_launch_intra_band(const frame_ctx_t *h_ctx, int mbh, uint2 *d_gran, unsigned *d_err, unsigned *d_band_done, hipStream_t s) {
    ib_args a;
    a.ctx = *h_ctx; a.gran = d_gran; a.err = d_err; a.band_done = d_band_done;
    if (h_ctx->i4x4) hipLaunchKernelGGL(intra_band_kernel<true>, dim3(k_intra_bands(mbh)), dim3(IB_ROWS * 128), 0, s, a);
    else hipLaunchKernelGGL(intra_band_kernel<false>, dim3(k_intra_bands(mbh)), dim3(IB_ROWS * 128), 0, s, a);
}

// =================================================================== intra pictures, dataflow at 4x4-block granularity (intra_mode 0)
// The band kernel above walks the picture in x + y order with one barrier per step: a macroblock is a unit, an Intra_4x4 macroblock is ten
// dependent sub-steps (blocks with bx + 2 by = s), so the critical path is 10 (mbw + mbh) sub-steps.  But macroblock x + 1 reads only the
// right column of macroblock x, whose block (3, by) is final after sub-step 3 + 2 by: its sub-step s may run once x has finished sub-step
// s + 3 -- a lag of four sub-steps, not ten.  And the row below reads only the bottom lines of the blocks (bx, 3), final after sub-steps
// 6 .. 9 (the top-right neighbour of a macroblock's block 5 is never used: the analysis leaves the two modes that read it out): a lag of
// eight.  Critical path 4 mbw + 8 mbh sub-steps (1080p: ~1030 against ~1880) -- if macroblocks overlap.  So:
//   * one workgroup per macroblock ROW, IR_LW luma waves taking the row's macroblocks in turn (x mod IR_LW: 2.5 macroblocks of a row are in
//     flight at a lag of four) and one chroma wave walking the row by itself (8x8 chroma is a third of an Intra_4x4 macroblock's work and
//     depends on nothing of luma): four waves, one per SIMD;
//   * inside the row everything is dataflow through LDS: a progress word per macroblock (x << 4 | sub-steps done) and its right column in a
//     ring of IR_RING slots; a wave waits for exactly the word it needs before the sub-step that needs it (hooks of intra_compute);
//   * between rows the bottom line of every 4x4 block column travels as an 8-byte granule {4 samples, tag}, one sc1 store the moment the
//     block is done, polled by the row below (MI355X_MICROARCH.md hand-off R2; the tag is the picture's epoch inverted, nothing is cleared);
//   * the record of a macroblock needs the luma and the chroma wave's flags: whoever finishes second writes it (one LDS word per macroblock in
//     a ring of eight; a wave that is eight macroblocks ahead of the other plane waits there).
// An Intra_16x16 macroblock needs its whole left column and top line, so it simply waits for "ten sub-steps done" of its left neighbour: the
// dataflow form degrades to the x + y order exactly where the standard demands it.  Every wait is bounded and reports through `err`.
#ifndef IR_LW
#define IR_LW 3
#endif
#ifndef IR_SLEEP
#define IR_SLEEP 1
#endif
#define IR_RING 4
#define IR_TR 8 /* macroblocks of top lines / source samples / decisions the movers keep in LDS */
#define IR_WAVES (IR_LW + 3)
#define IR_LDONE (1u << 30)
#define IR_CDONE (1u << 31)
struct ir_args { frame_ctx_t ctx; uint2 *gran; unsigned *err; unsigned *row_done; };
struct ir_shared {
    unsigned prog[IR_RING];                                   // (x << 4) | sub-steps done (10: the macroblock is complete), slot x % IR_RING
    unsigned rec[8];                                          // nzmask bits of one plane + IR_LDONE / IR_CDONE, slot x % 8
    __attribute__((aligned(4))) uint8_t right[IR_RING][16];   // right luma column of macroblock x
    // what the two mover waves bring in, rings of IR_TR macroblocks, and how far they and their readers are (all counts only grow)
    unsigned nly;                                             // 4 x + j: the luma top-line granules (x', 0..3) of every x' < x and (x, 0 .. j-1) are in topy
    unsigned ncu;                                             // macroblocks whose chroma top line is in topc
    unsigned nsrc;                                            // macroblocks whose source samples and decision are in srcy / srcc / dec
    unsigned ldone, cdone;                                    // macroblocks the luma waves / the chroma wave have completed
    __attribute__((aligned(16))) uint8_t topy[IR_TR][16], topc[IR_TR][16];
    __attribute__((aligned(16))) unsigned srcy[IR_TR][64];
    __attribute__((aligned(16))) uint2 srcc[IR_TR][32];
    __attribute__((aligned(16))) unsigned dec[IR_TR][8];
};
// LDS-only ordering: a wave's LDS operations are carried out in the order it issued them, so "data, then flag" needs nothing but the
// compiler to keep the order, and "flag, then data" nothing but the wait for the flag's value.  (A workgroup-scope acquire / release on a
// generic pointer also waits for every global store the wave has in flight -- the levels and reconstruction stores of the sub-step before:
// a memory round trip per sub-step on the dependency chain.)
DEV unsigned lds_ld_acq(const unsigned *p) {
    const unsigned v = *(const volatile __attribute__((address_space(3))) unsigned *)p;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    return v;
}
DEV void lds_st_rel(unsigned *p, unsigned v) {
    asm volatile("" ::: "memory");
    *(volatile __attribute__((address_space(3))) unsigned *)p = v;
}
// wave-uniform bounded wait for an LDS word to reach `need` (words only grow)
DEV void ir_wait_lds(const unsigned *p, unsigned need, unsigned *err, unsigned code) {
    int spins = 0;
    while ((int)(lds_ld_acq(p) - need) < 0) {
        __builtin_amdgcn_s_sleep(IR_SLEEP);
        if (++spins > 8 * DB_SPIN_MAX) { st_sc1(err, code); break; }
        if ((spins & 8191) == 0 && ld_sc1(err)) break;
    }
}
DEV void ir_put_record(const frame_ctx_t *ctx, int mbn, uint2 dec1, unsigned bits) {
    const bool use_i4 = ((dec1.x >> 16) & 255) != 0;
    mb_info_t mb;
    mb.mvx = 0; mb.mvy = 0; mb.mb_type = use_i4 ? 2 : 0; mb.i16_mode = use_i4 ? 0 : (uint8_t)(dec1.x & 255); mb.chroma_mode = (uint8_t)((dec1.x >> 8) & 255);
    mb.qp = (uint8_t)mb_qp_dev(ctx, mbn); mb.nzmask = bits & ~(IR_LDONE | IR_CDONE); mb.cost = dec1.y;
    st_mbinfo(&ctx->mbi[mbn], mb);
}
// deposit one plane's bits for macroblock x; the second plane to arrive writes the record and frees the slot (lane 0 of the wave)
DEV void ir_deposit(ir_shared *sh, const frame_ctx_t *ctx, int mbn, int x, uint2 dec1, unsigned bits, unsigned mine, unsigned other) {
    const unsigned old = atomicOr(&sh->rec[x & 7], bits | mine);
    if (old & other) { ir_put_record(ctx, mbn, dec1, old | bits); lds_st_rel(&sh->rec[x & 7], 0u); }
}
#ifdef IR_PROF /* debug builds: cycle counters per wave, left in ctx->isad (tests/devtools/irprof.py) */
#define IR_T0() const unsigned long long ir_t0 = __builtin_readcyclecounter()
#define IR_ACC(v) v += __builtin_readcyclecounter() - ir_t0
#else
#define IR_T0() do { } while (0)
#define IR_ACC(v) do { } while (0)
#endif
struct ir_luma_hook {
    static constexpr bool own_record = true;
    ir_shared *sh; uint8_t *T4; uint2 *gran_my; unsigned *err; unsigned tag; int x, lane, mbw; bool has_top, has_left, feeds;
    unsigned nz; bool ldc;
    unsigned long long c_wait_top = 0, c_wait_left = 0, c_after = 0;
    DEV void before(int s) {
        if (has_top && s <= 2) { // the top line arrives with the blocks that read it: (x, 0), (x, 1) and the corner for sub-step 0, then (x, 2), then (x, 3)
            { IR_T0(); ir_wait_lds(&sh->nly, 4u * (unsigned)x + (unsigned)(s + 2), err, 14u); IR_ACC(c_wait_top); }
            const uint8_t *line = sh->topy[x & (IR_TR - 1)];
            if (s == 0) {
                if (lane == 0) T4[3] = has_left ? sh->topy[(x - 1) & (IR_TR - 1)][15] : (uint8_t)0;
                else if (lane < 9) T4[3 + lane] = line[lane - 1];
            } else if (lane < 4) T4[8 + 4 * s + lane] = line[4 + 4 * s + lane];
        }
        if (has_left && !(s & 1) && s <= 6) { // block (0, s / 2) reads rows 2 s .. 2 s + 3 of the left neighbour's right column: final after its sub-step s + 3
            { IR_T0(); ir_wait_lds(&sh->prog[(x - 1) & (IR_RING - 1)], ((unsigned)(x - 1) << 4) | (unsigned)(s + 4), err, 18u); IR_ACC(c_wait_left); }
            if (lane < 4) {
                const uint8_t v = sh->right[(x - 1) & (IR_RING - 1)][2 * s + lane];
                T4[17 * 24 + 4 + 2 * s + lane] = v;        // transposed tile: the column left of the macroblock
                T4[(2 * s + lane + 1) * 24 + 3] = v;       // tile: the corner of the blocks (0, by > 0)
            }
        }
        WAVE_SYNC();
    }
    // straight from the lanes' registers: the right column of block (3, by) for macroblock x + 1, the bottom line of block (bx, 3) for the row below
    DEV void after(int s, int bx, int by, int px, int py, bool valid, int recp) {
        IR_T0();
        if (valid && bx == 3 && px == 3) sh->right[x & (IR_RING - 1)][4 * by + py] = (uint8_t)recp;
        if (feeds && s >= 6) {
            int v = recp << (8 * px);
            v |= quad_xor<1>(v);
            v |= quad_xor<2>(v);
            if (valid && by == 3 && py == 3 && px == 0) st64_sc1(gran_my + (size_t)x * 8 + bx, make_uint2((unsigned)v, tag));
        }
        if (lane == 0) lds_st_rel(&sh->prog[x & (IR_RING - 1)], ((unsigned)x << 4) | (unsigned)(s + 1));
        IR_ACC(c_after);
    }
    // Intra_8x8 (intra_compute<..., I8>): the left neighbour's right column half a macroblock at a time -- rows 0..7 are final once it has finished
    // sub-step 5 / its own block 1 (progress 6), rows 8..15 with the macroblock (10) -- into column -1 of the stride-32 tile
    DEV void before8(int b, uint8_t *R) {
        if (b == 1) { // block 1 predicts from the first eight samples of the macroblock above-right: its granules 0 and 1 (the row above does not wait for this row)
            const bool has_tr = has_top && x + 1 < mbw;
            if (has_tr) { IR_T0(); ir_wait_lds(&sh->nly, 4u * (unsigned)(x + 1) + 2u, err, 14u); IR_ACC(c_wait_top); }
            if (lane < 8) R[20 + lane] = has_tr ? sh->topy[(x + 1) & (IR_TR - 1)][lane] : (uint8_t)0;
        }
        if (has_left && !(b & 1)) {
            { IR_T0(); ir_wait_lds(&sh->prog[(x - 1) & (IR_RING - 1)], ((unsigned)(x - 1) << 4) | (b ? 10u : 6u), err, 18u); IR_ACC(c_wait_left); }
            if (lane < 8) R[(4 * b + lane + 1) * 32 + 3] = sh->right[(x - 1) & (IR_RING - 1)][4 * b + lane];
        }
    }
    // ... and this macroblock's right column behind blocks 1 and 3, "six sub-steps done" behind block 1 (what an Intra_4x4 or Intra_8x8 macroblock to
    // the right waits for before it reads rows 0..7), the bottom line of blocks 2 and 3 as two granules each for the row below; the caller publishes "complete"
    DEV void after8(int b, int bx, int by, int recp) {
        if (feeds && b >= 2) {
            int v = recp << (8 * (bx & 3));
            v |= quad_xor<1>(v);
            v |= quad_xor<2>(v);
            if (by == 7 && !(bx & 3)) st64_sc1(gran_my + (size_t)x * 8 + 2 * (b & 1) + (bx >> 2), make_uint2((unsigned)v, tag));
        }
        if (b & 1) {
            if (bx == 7) sh->right[x & (IR_RING - 1)][4 * (b - 1) + by] = (uint8_t)recp;
            if (b == 1 && lane == 0) lds_st_rel(&sh->prog[x & (IR_RING - 1)], ((unsigned)x << 4) | 6u);
        }
    }
    DEV void luma_done(unsigned nzb, bool dc) { nz = nzb; ldc = dc; }
    DEV void chroma_done(unsigned, unsigned) {}
};
struct ir_chroma_hook {
    static constexpr bool own_record = true;
    unsigned nz8, dc2;
    DEV void before(int) {}
    DEV void after(int, int, int, int, int, bool, int) {}
    DEV void luma_done(unsigned, bool) {}
    DEV void chroma_done(unsigned a, unsigned b) { nz8 = a; dc2 = b; }
};

template <bool IR_I8> // Intra_8x8 macroblocks may occur (ctx->i8).  A build of its own: with the branch compiled in, the Intra_4x4 chain of the default path is 9 % slower (register allocation)
__global__ __launch_bounds__(64 * IR_WAVES) void intra_rows_kernel(ir_args a) {
    __shared__ intra_lds LD[IR_LW + 1]; // private to each compute wave
    __shared__ ir_shared SH;
    __shared__ unsigned tabw[TAB_DWORDS];
    const dev_tables *T = (const dev_tables *)tabw;
    const frame_ctx_t *__restrict__ ctx = &a.ctx;
    const int mbw = ctx->mbw, mbh = ctx->mbh, my = blockIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const bool has_top = row_has_top(ctx, my), feeds = my < mbh - 1; // (a slice's first row reads nothing of the row above and does not wait for it: the slices are independent chains)
    for (int i = threadIdx.x; i < TAB_DWORDS; i += 64 * IR_WAVES) tabw[i] = ((const unsigned *)&g_tab)[i];
    if (threadIdx.x < IR_RING) SH.prog[threadIdx.x] = 0;
    if (threadIdx.x < 8) SH.rec[threadIdx.x] = 0;
    if (threadIdx.x == 0) { SH.nly = 0; SH.ncu = 0; SH.nsrc = 0; SH.ldone = 0; SH.cdone = 0; }
    const unsigned tag = ~ctx->epoch;
    const uint2 *gran_up = a.gran + (size_t)(my > 0 ? my - 1 : 0) * mbw * 8;
    uint2 *gran_my = a.gran + (size_t)my * mbw * 8;
    const int spy = (lane >> 2) & 3;
    __syncthreads();
    if (w < IR_LW) {
        // ================================================================ luma: macroblocks w, w + IR_LW, ...  Nothing here waits for memory:
        // source samples, decisions and the row above's bottom lines come through LDS (the movers), stores are fire-and-forget.
        intra_lds *L = &LD[w];
        ir_luma_hook hk;
        hk.sh = &SH; hk.T4 = L->T4; hk.gran_my = gran_my; hk.err = a.err; hk.tag = tag; hk.lane = lane; hk.mbw = mbw; hk.has_top = has_top; hk.feeds = feeds;
#ifdef IR_PROF
        unsigned long long c_mb4 = 0, c_mb16 = 0, c_src = 0, c_rec = 0, n4 = 0, n16 = 0;
        const unsigned long long c_start = __builtin_readcyclecounter();
#endif
        for (int x = w; x < mbw; x += IR_LW) {
            { IR_T0(); ir_wait_lds(&SH.nsrc, (unsigned)x + 1u, a.err, 20u); IR_ACC(c_src); }
            const unsigned *dw = SH.dec[x & (IR_TR - 1)];
            const uint4 dec0 = make_uint4(dw[0], dw[1], dw[2], dw[3]);
            const uint2 dec1 = make_uint2(dw[4], dw[5]);
            const uint2 srcc = make_uint2(SH.srcy[x & (IR_TR - 1)][lane], 0u);
            const int itype = (int)((dec1.x >> 16) & 255);
            const bool use_i4 = itype == 1, has_left = x > 0;
            hk.x = x; hk.has_left = has_left; hk.nz = 0; hk.ldc = false;
            if (!use_i4) { // Intra_16x16: the whole top line, the corner, the whole left column; Intra_8x8: the top line and the corner (the rest through its hooks)
                int tv = 0, lv = 0;
                if (has_left && itype != 2) ir_wait_lds(&SH.prog[(x - 1) & (IR_RING - 1)], ((unsigned)(x - 1) << 4) | 10u, a.err, 18u); // (Intra_8x8: the hook, per half)
                if (has_top) {
                    ir_wait_lds(&SH.nly, 4u * (unsigned)x + 4u, a.err, 14u);
                    if (lane >= 1 && lane < 17) tv = (int)SH.topy[x & (IR_TR - 1)][lane - 1];
                    if (lane == 0 && has_left) tv = (int)SH.topy[(x - 1) & (IR_TR - 1)][15];
                }
                if (has_left && itype != 2 && lane >= 1 && lane < 17) lv = (int)SH.right[(x - 1) & (IR_RING - 1)][lane - 1];
                if (lane == 0) lv = tv; // the corner belongs to both arrays (0 unless both neighbours exist)
                if (lane < 17) { L->top[0][lane] = tv; L->left[0][lane] = lv; }
                WAVE_SYNC();
            }
            {
                IR_T0();
                if (lane == 0) { // a luma wave eight macroblocks ahead of the chroma wave waits for its record slot
                    int spins = 0;
                    while (lds_ld_acq(&SH.rec[x & 7]) & IR_LDONE) { __builtin_amdgcn_s_sleep(2); if (++spins > 8 * DB_SPIN_MAX) { st_sc1(a.err, 19u); break; } }
                }
                IR_ACC(c_rec);
            }
            {
                IR_T0();
                intra_compute<true, false, ir_luma_hook, IR_I8>(ctx, T, L, x, my, 0, lane, dec0, dec1, &srcc, hk);
#ifdef IR_PROF
                if (use_i4) { IR_ACC(c_mb4); n4++; } else { IR_ACC(c_mb16); n16++; }
#endif
            }
            if (!use_i4) { // published at once: right column, bottom line, "complete"
                WAVE_SYNC();
                if (lane < 16) SH.right[x & (IR_RING - 1)][lane] = L->right_y[lane];
                if (feeds && lane < 4 && itype != 2) st64_sc1(gran_my + (size_t)x * 8 + lane, make_uint2(((const unsigned *)L->bot_y[x & 3])[lane], tag)); // (Intra_8x8: behind blocks 2 and 3)
                if (lane == 0) lds_st_rel(&SH.prog[x & (IR_RING - 1)], ((unsigned)x << 4) | 10u);
            }
            if (lane == 0) {
                ir_deposit(&SH, ctx, my * mbw + x, x, dec1, hk.nz | (hk.ldc ? NZ_LDC : 0u), IR_LDONE, IR_CDONE);
                atomicMax(&SH.ldone, (unsigned)x + 1u); // (macroblocks complete in order -- x cannot pass sub-step 6 before x - 1 is done -- but the waves reach this line in any order)
            }
        }
#ifdef IR_PROF
        if (lane == 0 && my < 4) {
            unsigned *o = (unsigned *)ctx->isad + (my * 4 + w) * 16;
            o[0] = (unsigned)(__builtin_readcyclecounter() - c_start); o[1] = (unsigned)c_mb4; o[2] = (unsigned)n4; o[3] = (unsigned)c_mb16; o[4] = (unsigned)n16;
            o[5] = (unsigned)hk.c_wait_top; o[6] = (unsigned)hk.c_wait_left; o[7] = (unsigned)hk.c_after; o[8] = (unsigned)c_src; o[9] = (unsigned)c_rec;
        }
#endif
    } else if (w == IR_LW) {
        // ================================================================ chroma: every macroblock of the row, left to right
        intra_lds *L = &LD[IR_LW];
        ir_chroma_hook hk;
        const bool tlane = lane < 18; // plane c = lane / 9, sample i = lane % 9 - 1 (-1: the corner)
        const int mc = lane / 9, mi = lane % 9 - 1;
#ifdef IR_PROF
        unsigned long long c_cmp = 0, c_src = 0, c_top = 0, c_rec = 0;
        const unsigned long long c_start = __builtin_readcyclecounter();
#endif
        for (int x = 0; x < mbw; x++) {
            { IR_T0(); ir_wait_lds(&SH.nsrc, (unsigned)x + 1u, a.err, 20u); IR_ACC(c_src); }
            const unsigned *dw = SH.dec[x & (IR_TR - 1)];
            const uint2 dec1 = make_uint2(dw[4], dw[5]);
            const uint2 srcc = SH.srcc[x & (IR_TR - 1)][lane & 31];
            const bool has_left = x > 0;
            int tv = 0, lv = 0;
            if (has_top) {
                { IR_T0(); ir_wait_lds(&SH.ncu, (unsigned)x + 1u, a.err, 14u); IR_ACC(c_top); }
                if (tlane && (mi >= 0 || has_left)) tv = mi >= 0 ? (int)SH.topc[x & (IR_TR - 1)][2 * mi + mc] : (int)SH.topc[(x - 1) & (IR_TR - 1)][14 + mc];
            }
            if (tlane && has_left && (mi >= 0 || has_top)) lv = mi >= 0 ? (int)L->right_c[mc][mi] : tv;
            WAVE_SYNC(); // (right_c of the previous macroblock has been read)
            if (tlane) { L->top[1 + mc][mi + 1] = tv; L->left[1 + mc][mi + 1] = lv; }
            {
                IR_T0();
                if (lane == 0) {
                    int spins = 0;
                    while (lds_ld_acq(&SH.rec[x & 7]) & IR_CDONE) { __builtin_amdgcn_s_sleep(2); if (++spins > 8 * DB_SPIN_MAX) { st_sc1(a.err, 19u); break; } }
                }
                WAVE_SYNC();
                IR_ACC(c_rec);
            }
            { IR_T0(); intra_compute<true, false, ir_chroma_hook>(ctx, T, L, x, my, 1, lane, make_uint4(0, 0, 0, 0), dec1, &srcc, hk); IR_ACC(c_cmp); }
            WAVE_SYNC();
            if (feeds && lane < 4) st64_sc1(gran_my + (size_t)x * 8 + 4 + lane, make_uint2(((const unsigned *)L->bot_c[x & 3])[lane], tag));
            unsigned bits = hk.nz8 << 16;
            if (hk.dc2 & 1) bits |= NZ_CBDC;
            if (hk.dc2 & 2) bits |= NZ_CRDC;
            if (lane == 0) {
                ir_deposit(&SH, ctx, my * mbw + x, x, dec1, bits, IR_CDONE, IR_LDONE);
                lds_st_rel(&SH.cdone, (unsigned)x + 1u);
            }
        }
#ifdef IR_PROF
        if (lane == 0 && my < 4) {
            unsigned *o = (unsigned *)ctx->isad + (my * 4 + 3) * 16;
            o[0] = (unsigned)(__builtin_readcyclecounter() - c_start); o[1] = (unsigned)c_cmp; o[2] = (unsigned)mbw; o[5] = (unsigned)c_top; o[8] = (unsigned)c_src; o[9] = (unsigned)c_rec;
        }
#endif
    } else if (w == IR_LW + 1) {
        // ================================================================ mover G: the bottom lines of the row above, granule by granule, into LDS.
        // Lanes 0..3 follow the luma granules of macroblock xl (they arrive in the order 0 1 2 3), lanes 4..7 the chroma line of
        // macroblock xc.  This wave stores nothing, so its waits for the polled loads are nothing but their latency.
        if (has_top) {
            int xl = 0, nj = 0, xc = 0, idle = 0;
            while (xl < mbw || xc < mbw) {
                const unsigned ld = lds_ld_acq(&SH.ldone), cd = lds_ld_acq(&SH.cdone);
                const bool go_l = xl < mbw && xl < (int)ld + IR_TR - 2, go_c = xc < mbw && xc < (int)cd + IR_TR - 2; // the ring slot's previous line (macroblock x - 8) is the corner of x - 7
                const bool mine = lane < 4 ? go_l : (lane < 8 && go_c);
                const int gx = lane < 4 ? (xl < mbw ? xl : mbw - 1) : (xc < mbw ? xc : mbw - 1);
                uint2 g = make_uint2(0u, ~tag);
                if (mine) g = ld64_sc1(gran_up + (size_t)gx * 8 + (lane & 7));
                const unsigned ok = (unsigned)__ballot(mine && g.y == tag);
                bool moved = false;
                int np = __builtin_ctz(~(ok & 15u)); // granules 0 .. np-1 of xl are there
                if (go_l && np > nj) {
                    if (lane >= nj && lane < np) *(unsigned *)&SH.topy[xl & (IR_TR - 1)][4 * lane] = g.x;
                    nj = np; moved = true;
                    if (nj == 4) { xl++; nj = 0; }
                    if (lane == 0) lds_st_rel(&SH.nly, 4u * (unsigned)xl + (unsigned)nj);
                }
                if (go_c && (ok & 0xF0u) == 0xF0u) {
                    if (lane >= 4 && lane < 8) *(unsigned *)&SH.topc[xc & (IR_TR - 1)][4 * (lane - 4)] = g.x;
                    xc++; moved = true;
                    if (lane == 0) lds_st_rel(&SH.ncu, (unsigned)xc);
                }
                if (moved) idle = 0;
                else {
                    __builtin_amdgcn_s_sleep(1);
                    if (++idle > DB_SPIN_MAX) { st_sc1(a.err, 14u); break; }
                    if ((idle & 1023) == 0 && ld_sc1(a.err)) break;
                }
            }
        }
    } else {
        // ================================================================ mover S: source samples and decisions, two macroblocks ahead of their landing
        int sy = my * 16 + 4 * (lane >> 4) + spy;
        sy = sy < ctx->vis_h ? sy : ctx->vis_h - 1;
        const uint8_t *srow = ctx->src_y + (size_t)sy * ctx->src_stride + 4 * (lane & 3);
        const int vh2 = ctx->vis_h >> 1;
        int cy = my * 8 + 4 * ((lane >> 4) & 1) + spy;
        cy = cy < vh2 ? cy : vh2 - 1;
        const uint8_t *crow = ctx->src_uv + (size_t)cy * ctx->src_stride + 8 * (lane & 1);
        const uint8_t *drow = ctx->idec + (size_t)my * mbw * IDEC_BYTES + 4 * (lane & 7);
        unsigned yA, yB, dA, dB;
        uint2 cA, cB;
        auto issue = [&](int x, unsigned &yv, uint2 &cv, unsigned &dv) __attribute__((always_inline)) {
            const int xc = x < mbw ? x : mbw - 1;
            yv = ldg32(srow + xc * 16); cv = ldg64(crow + xc * 16); dv = ldg32(drow + (size_t)xc * IDEC_BYTES);
        };
        auto land = [&](int x, unsigned yv, uint2 cv, unsigned dv) __attribute__((always_inline)) {
            // slot x % IR_TR held macroblock x - IR_TR: both planes must be through with it
            ir_wait_lds(&SH.ldone, (unsigned)(x - IR_TR + 1), a.err, 20u);
            ir_wait_lds(&SH.cdone, (unsigned)(x - IR_TR + 1), a.err, 20u);
            SH.srcy[x & (IR_TR - 1)][lane] = yv;
            if (lane < 32) SH.srcc[x & (IR_TR - 1)][lane] = cv;
            if (lane < 8) SH.dec[x & (IR_TR - 1)][lane] = dv;
            if (lane == 0) lds_st_rel(&SH.nsrc, (unsigned)x + 1u);
        };
        issue(0, yA, cA, dA);
        issue(1, yB, cB, dB);
        for (int x = 0; x < mbw; x += 2) {
            land(x, yA, cA, dA);
            issue(x + 2, yA, cA, dA);
            if (x + 1 < mbw) land(x + 1, yB, cB, dB);
            issue(x + 3, yB, cB, dB);
        }
    }
    // ---- this row's reconstruction and records are complete: tell the band deblocker, which may be waiting on another stream (release
    // pattern of MI355X_MICROARCH.md: every storing wave drains, the workgroup meets, one lane writes this XCD's L2 back, then the tagged flag)
    if (a.row_done) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            st_sc1(a.row_done + my, tag);
        }
    }
}
// d_gran: 8 granules per macroblock (four luma block columns, four words of the interleaved chroma line), one set per row
void k_launch_intra_rows(const frame_ctx_t *h_ctx, int mbh, uint2 *d_gran, unsigned *d_err, unsigned *d_row_done, hipStream_t s) {
    ir_args a;
    a.ctx = *h_ctx; a.gran = d_gran; a.err = d_err; a.row_done = d_row_done;
    if (h_ctx->i8) hipLaunchKernelGGL(intra_rows_kernel<true>, dim3(mbh), dim3(64 * IR_WAVES), 0, s, a);
    else hipLaunchKernelGGL(intra_rows_kernel<false>, dim3(mbh), dim3(64 * IR_WAVES), 0, s, a);
}

// =================================================================== launchers
int k_intra_diags(int mbw, int mbh) { return mbw + mbh - 1; }
void k_launch_intra_analyse(const frame_ctx_t *h_ctx, int mbw, int mbh, int gate_p, hipStream_t s) {
    if (gate_p) hipLaunchKernelGGL(intra_analyse_gated_kernel, dim3((mbw * mbh + 3) / 4), dim3(256), 0, s, *h_ctx);
    else hipLaunchKernelGGL(intra_analyse_kernel, dim3((mbw * mbh + 3) / 4), dim3(256), 0, s, *h_ctx);
}

// =================================================================== intra macroblocks of P pictures (body: intra_mb.hpp, intra_p_row)
__global__ __launch_bounds__(128) void intra_p_kernel(ip_args a) { intra_p_row(a, (int)blockIdx.x, nullptr, 0u); }
void k_launch_intra_p(const frame_ctx_t *h_ctx, int mbw, int mbh, unsigned *d_progress, uint8_t *d_strips, unsigned *d_err, hipStream_t s) {
    (void)mbw;
    ip_args a;
    a.ctx = *h_ctx; a.progress = d_progress; a.strips = d_strips; a.err = d_err;
    hipLaunchKernelGGL(intra_p_kernel, dim3(mbh), dim3(128), 0, s, a);
}
void k_launch_intra_diag(const frame_ctx_t *d_ctx, int mbw, int mbh, int diag, hipStream_t s) {
    int y_lo = diag - (mbw - 1) > 0 ? diag - (mbw - 1) : 0;
    int y_hi = diag < mbh - 1 ? diag : mbh - 1;
    if (y_hi < y_lo) return;
    hipLaunchKernelGGL(intra_kernel, dim3(y_hi - y_lo + 1), dim3(128), 0, s, d_ctx, diag);
}
