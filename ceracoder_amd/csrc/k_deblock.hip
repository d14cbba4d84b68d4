// k_deblock.hip -- in-loop deblocking filter (H.264 8.7): the persistent band kernel, and the per-diagonal form
// Hand-written HIP for gfx950 (CDNA4, wave64); part of libmi355enc (see kernels_common.hpp).
#include "intra_mb.hpp"

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
// ... with partitions (ctx->partitions): the vectors of the 8x8 quadrants the two 4x4 blocks lie in (vp / vq: x | y << 16 per quadrant, raster)
DEV int bs_of_q(const mb_info_t &mp, const unsigned *vp, int bxp, int byp, const mb_info_t &mq, const unsigned *vq, int bxq, int byq, bool mb_edge) {
    if (mp.mb_type != 1 || mq.mb_type != 1) return mb_edge ? 4 : 3;
    if (has_coef(mp, bxp, byp) || has_coef(mq, bxq, byq)) return 2;
    const unsigned a = vp[(byp >> 1) * 2 + (bxp >> 1)], b = vq[(byq >> 1) * 2 + (bxq >> 1)];
    if (iabs((int)(int16_t)(a & 0xFFFF) - (int)(int16_t)(b & 0xFFFF)) >= 4 || iabs((int)(int16_t)(a >> 16) - (int)(int16_t)(b >> 16)) >= 4) return 1;
    return 0;
}
// the quadrant vectors of macroblock mbn (record m): partition 0's from the record, the others from the luma-DC slot of its levels
DEV void ld_qmv(const frame_ctx_t *__restrict__ ctx, int mbn, const mb_info_t &m, unsigned *v) {
    const unsigned v0 = ((unsigned)(uint16_t)m.mvx) | ((unsigned)(uint16_t)m.mvy << 16);
    const int shape = m.mb_type == 1 ? (m.i16_mode & 3) : 0;
    v[0] = v[1] = v[2] = v[3] = v0;
    if (shape) {
        const int16_t *l = ctx->levels + (size_t)mbn * MB_LEVELS + L_LDC;
        const unsigned w0 = ldg32(l), w1 = ldg32(l + 2), w2 = ldg32(l + 4); // partitions 1, 2, 3
        if (shape == 1) { v[2] = v[3] = w0; }
        else if (shape == 2) { v[1] = v[3] = w0; }
        else { v[1] = w0; v[2] = w1; v[3] = w2; }
    }
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
            if ((e == 0 && !db_has_top(ctx, my)) || ((cur.nzmask & NZ_T8) && (e & 1))) continue;
            const mb_info_t &mp = e == 0 ? upp : cur;
            int bS = bs_of(mp, k >> 2, e == 0 ? 3 : e - 1, cur, k >> 2, e, e == 0);
            filter_line(T, &tl[(4 + 4 * e) * TLS + 4 + k], TLS, bS, mp.qp, cur.qp, false);
        }
    } else if (lane < 24) {
        const int k = lane - 16;
#pragma unroll
        for (int e = 0; e < 4; e += 2) {
            if (e == 0 && !db_has_top(ctx, my)) continue;
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
struct db_args { frame_ctx_t ctx; unsigned *err; int band0, nb_total; uint2 *gran; const unsigned *ip_progress; unsigned *partab; const unsigned *iband_done; int ib_rows; unsigned *band_done; unsigned *started; const unsigned *row_done; unsigned row_need; int nip; unsigned *ip_progress_w; uint8_t *ip_strips; unsigned *ip_done; unsigned *qpc; unsigned qpc_base; unsigned *part_cnt; }; // qpc (adaptive quantisation): the launch's first workgroup resolves the QP_Y chain row by row and counts the rows there (qp_chain_rows) // nip > 0: the launch's first nip workgroups are the picture's intra macroblock rows (intra_p_row) // ip_progress: see GATED; partab: see the prologue // a launch covers bands band0 .. band0 + gridDim.x/2 - 1

// =================================================================== deblocking, persistent: bands of rows in x + y order
// One launch per picture instead of one per wavefront.  Two observations shorten the dependency chain:
//  (1) Boundary strengths and the alpha/beta/tc0 triples depend only on the macroblock records,
//      so they are computed up front (the band kernel's prologue, db_record):
//      64 bytes per macroblock {bS nibbles V/H, six packed parameter pairs}.
//  (2) x + 2y is sufficient but not necessary.  Macroblock (x, y) only conflicts with its
//      top-right neighbour (x+1, y-1) on the 3x3 corner of (x, y-1) that the neighbour's left
//      edge (a VERTICAL edge, first thing it filters) and this macroblock's top edge (a
//      HORIZONTAL edge, filtered after all vertical ones) both touch.  If every row filters its
//      vertical edges, all rows meet at one barrier, and then every row filters its horizontal
//      edges, (x, y) and (x+1, y-1) can share a step: the order x + y with ONE barrier per step
//      reproduces the raster-order result (mbw + mbh - 1 steps instead of mbw + 2(mbh - 1)).
// A band waits only on the band above it, so the wait graph is acyclic; every spin is bounded and reports through `err`.
#define DBREC_BYTES 64
DEV unsigned pack_par_ab(const dev_tables *T, int idx) { return (unsigned)T->alpha[idx] | ((unsigned)T->beta[idx] << 8); }
DEV unsigned pack_par_tc(const dev_tables *T, int idx) { return (unsigned)T->tc0[idx][0] | ((unsigned)T->tc0[idx][1] << 8) | ((unsigned)T->tc0[idx][2] << 16); }
DEV edge_par par_of(unsigned ab, unsigned tc) { edge_par p; p.alpha = (int)(ab & 255); p.beta = (int)(ab >> 8); p.tc0 = tc; return p; }

// The record of one macroblock: words 0-1 bS of the vertical edges (nibble 4*edge + segment), 2-3 of the horizontal edges, then
// {alpha|beta<<8, tc0 bytes} for luma left / top / inner and chroma left / top / inner.  Edges that are not filtered (picture
// border, 8x8-transform inner edges) get bS 0.  Returns whether any edge of the macroblock has bS != 0.  (A flat kernel of its own
// in r01 and most of r02; now the band kernel's prologue: one launch and one launch gap less on the chain that sets the picture
// period.)
DEV bool db_record(const frame_ctx_t *__restrict__ ctx, const dev_tables *T, const int mx, const int my, unsigned *w) {
    const int mbw = ctx->mbw, i = my * mbw + mx;
    const mb_info_t cur = ld_mbinfo(&ctx->mbi[i]);
    const mb_info_t lft = ld_mbinfo(&ctx->mbi[mx > 0 ? i - 1 : i]);
    const mb_info_t upp = ld_mbinfo(&ctx->mbi[my > 0 ? i - mbw : i]);
    const bool t8 = (cur.nzmask & NZ_T8) != 0;
    const bool dbtop = db_has_top(ctx, my); // the top macroblock edge is filtered (8.7: not the picture's; with slice-local deblocking not a slice's)
    const bool parts = ctx->partitions != 0; // (wave-uniform)
    unsigned vc[4], vl[4], vu[4];
    if (parts) { ld_qmv(ctx, i, cur, vc); ld_qmv(ctx, mx > 0 ? i - 1 : i, lft, vl); ld_qmv(ctx, my > 0 ? i - mbw : i, upp, vu); }
    unsigned long long bv = 0, bh = 0;
#pragma unroll
    for (int e = 0; e < 4; e++)
#pragma unroll
        for (int sg = 0; sg < 4; sg++) {
            int v = 0, h = 0;
            if (!(t8 && (e & 1))) {
                if (!parts) {
                    if (!(e == 0 && mx == 0)) v = bs_of(e == 0 ? lft : cur, e == 0 ? 3 : e - 1, sg, cur, e, sg, e == 0);
                    if (!(e == 0 && !dbtop)) h = bs_of(e == 0 ? upp : cur, sg, e == 0 ? 3 : e - 1, cur, sg, e, e == 0);
                } else {
                    if (!(e == 0 && mx == 0)) v = bs_of_q(e == 0 ? lft : cur, e == 0 ? vl : vc, e == 0 ? 3 : e - 1, sg, cur, vc, e, sg, e == 0);
                    if (!(e == 0 && !dbtop)) h = bs_of_q(e == 0 ? upp : cur, e == 0 ? vu : vc, sg, e == 0 ? 3 : e - 1, cur, vc, sg, e, e == 0);
                }
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
    return (bv | bh) != 0;
}

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
        // both arms of every select are named values computed up front: written inside the ?: they stay a branch diamond
        // (exec-mask save/restore around a few instructions) on the dependency chain
        const int sp0a = (p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3, sp0b = (2 * p1 + p0 + q1 + 2) >> 2;
        const int sq0a = (p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3, sq0b = (2 * q1 + q0 + p1 + 2) >> 2;
        const int sp0 = sp ? sp0a : sp0b, sq0 = sq ? sq0a : sq0b;
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
DEV int bsel(int m, int a, int b) { return (a & m) | (b & ~m); } // v_bfi_b32: m all-ones -> a
// Inner edges (bS < 4) with the conditions kept as VALU masks (sign bit of `value - threshold`, spread by an arithmetic
// shift) instead of compares into SGPR pairs + s_and + v_cndmask: no VALU -> SALU -> VALU round trips on the chain.
DEV void edge_luma_inner(const edge_par &P, int &p2, int &p1, int &p0, int &q0, int &q1, int &q2, int bS) {
    const int alpha = P.alpha, beta = P.beta;
    const int mf = ((adiff(p0, q0) - alpha) & (adiff(p1, p0) - beta) & (adiff(q1, q0) - beta) & -bS) >> 31; // all true (bS is 0..3 here)
    const int map = (adiff(p2, p0) - beta) >> 31, maq = (adiff(q2, q0) - beta) >> 31;
    const int tc0 = (int)((P.tc0 >> (8 * ((bS - 1) & 3))) & 0xFF); // bS 0 reads a don't-care byte
    const int tc = tc0 - map - maq;
    const int dl = clip3(-tc, tc, (((q0 - p0) << 2) + (p1 - q1) + 4) >> 3);
    const int avg = (p0 + q0 + 1) >> 1;
    const int np1 = p1 + clip3(-tc0, tc0, (p2 + avg - (p1 << 1)) >> 1);
    const int nq1 = q1 + clip3(-tc0, tc0, (q2 + avg - (q1 << 1)) >> 1);
    const int np0 = clip255(p0 + dl), nq0 = clip255(q0 - dl);
    p0 = bsel(mf, np0, p0); q0 = bsel(mf, nq0, q0);
    p1 = bsel(mf & map, np1, p1); q1 = bsel(mf & maq, nq1, q1);
}
DEV void edge_chroma2(const edge_par &P, int p1, int &p0, int &q0, int q1, int bS) {
    const bool f = (bS != 0) & (adiff(p0, q0) < P.alpha) & (adiff(p1, p0) < P.beta) & (adiff(q1, q0) < P.beta);
    const int tc = (int)((P.tc0 >> (8 * ((bS - 1) & 3))) & 0xFF) + 1;
    const int dl = clip3(-tc, tc, (((q0 - p0) << 2) + (p1 - q1) + 4) >> 3);
    const bool s4 = bS == 4;
    const int sp0 = (2 * p1 + p0 + q1 + 2) >> 2, sq0 = (2 * q1 + q0 + p1 + 2) >> 2, wp0 = clip255(p0 + dl), wq0 = clip255(q0 - dl);
    const int np0 = s4 ? sp0 : wp0, nq0 = s4 ? sq0 : wq0; // named arms: see edge_luma2
    p0 = f ? np0 : p0; q0 = f ? nq0 : q0;
}

// ------------------------------------------------------------------ the band kernel: the four edges of a direction in parallel
// The four vertical (then the four horizontal) luma edges of a macroblock are almost independent: a normal-strength filter
// changes p1, p0, q0, q1 and reads p2 and q2; q2 of edge e is p1 of edge e + 1 (which edge e + 1 changes, but only after edge e
// has read it: everybody reads originals) and q1 of edge e is p2 of edge e + 1 -- the single true dependency.  q1' needs only
// the q side, p0 and the filter flag, none of which involve p2, so all four edges compute q1' at once, hand it to the next
// edge's lanes with one DPP move, and finish.  The strong filter (bS 4) exists only on the macroblock edge and also changes
// q2 = p1 of edge 1, which edge 1's flag reads: a wave that sees a bS 4 runs the macroblock edge as a pass of its own first.
// So the chain of a step is one or two edge filters per direction instead of four, on 64 lanes = 16 lines x 4 edges of ONE
// macroblock (lane = 4 * line + edge), and a workgroup is a band of DB_ROWS rows (luma and chroma in separate workgroups:
// they share nothing but the records).
//
// Three waves per macroblock row.  A first form of this kernel gave a row ONE wave that did everything; per-phase cycle
// counters showed its step to be ~25 % filters and ~75 % data movement (prefetch, landing, ring copies, final-sample reads,
// stores) on the very wave whose instruction stream IS the dependency chain.  Here F filters and touches nothing but LDS; M
// loads macroblocks two steps ahead, lands them in a ring of 8 tiles and turns the 64-byte record into one parameter word
// per lane and direction {bS, alpha, beta, tc0[bS]}; S reads finished lines out of the ring and stores them, and publishes the
// band's bottom strip.  With 4-row bands that is 12 waves, M and S sharing a SIMD with their row's F: they issue VMEM / LDS /
// SALU beside its VALU.  Nothing is copied between steps: the left strip of macroblock x is columns 12..15 of tile x-1, the
// strip above is rows 12..15 of the tile of the row above (for the first row of a band: a small ring its M wave fills from the
// band above's granules), so vertical edge 0 and horizontal edge 0 simply address the neighbouring tile.
// Strips between bands travel as 8-byte {samples, picture epoch} granules, one `sc1` store each, which the consumer polls
// directly (MI355X_MICROARCH.md, hand-off R2 / handoff-1to1): no drain, no separate counter, nothing to clear.
// 1080p P picture: 16 lanes per macroblock + 16-row bands + strips/drain/counter 0.34 ms -> one wave per row, edges in
// parallel 0.25 -> granules 0.20 -> three waves per row 0.13 (I pictures 0.45 -> 0.22).
DEV int quad_prev(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x90, 0xF, 0xF, false); } // quad_perm:[0,0,1,2]: value of lane - 1 of the quad
DEV int quad_next(int v) { return __builtin_amdgcn_update_dpp(0, v, 0xF9, 0xF, 0xF, false); } // quad_perm:[1,2,3,3]: value of lane + 1 of the quad
#define DBT_NB 8
struct dbt_luma { uint8_t t[DBT_NB][256]; uint8_t up[DBT_NB][64]; unsigned par[4][32]; };   // 3.1 KB per row
struct dbt_chroma { uint8_t t[DBT_NB][128]; uint8_t up[DBT_NB][32]; unsigned par[4][32]; }; // 1.8 KB per row
DEV uint4 lds128(const void *p) { return *(const uint4 *)p; }

// The four luma edges of one direction; lane = 4 * line + edge, `par` = this lane's {bS, alpha, beta, tc0}.  s[0..7] = p3 p2 p1 p0 q0 q1
// q2 q3 of this lane's edge (originals); on return s[2..5] (p1 p0 q0 q1) are final, for edge 0 also s[1] (p2), and s[6] is NOT (it
// is the next edge's p1).
template <bool ALL_INTRA>
DEV void edges4p_luma(const int e, const unsigned par, const bool any4, int *s) {
    int bS = (int)(par & 15u);
    const int alpha = (int)((par >> 8) & 255u), beta = (int)((par >> 16) & 255u), tc0 = (int)(par >> 24);
    if (ALL_INTRA || any4) { // the macroblock edge first (lanes of the other edges compute and discard)
        edge_par PE; PE.alpha = alpha; PE.beta = beta; PE.tc0 = (unsigned)tc0 * 0x010101u;
        edge_luma2<true>(PE, s[0], s[1], s[2], s[3], s[4], s[5], s[6], s[7], e == 0 ? bS : 0, true);
        const int f1 = quad_prev(s[5]), f2 = quad_prev(s[6]); // edge 0's q1', q2' are edge 1's p2, p1
        s[1] = e == 1 ? f1 : s[1]; s[2] = e == 1 ? f2 : s[2];
        bS = e == 0 ? 0 : bS;
    }
    const int p1 = s[2], p0 = s[3], q0 = s[4], q1 = s[5], q2 = s[6];
    const int mf = ((adiff(p0, q0) - alpha) & (adiff(p1, p0) - beta) & (adiff(q1, q0) - beta) & -bS) >> 31; // bS is 0..3 here
    const int maq = (adiff(q2, q0) - beta) >> 31;
    const int avg = (p0 + q0 + 1) >> 1;
    const int nq1 = q1 + clip3(-tc0, tc0, (q2 + avg - (q1 << 1)) >> 1);
    const int q1f = bsel(mf & maq, nq1, q1);
    const int fw = quad_prev(q1f);
    const int p2 = e == 0 ? s[1] : fw; // the previous edge's q1' (edge 0: the neighbouring macroblock's sample, final)
    const int map = (adiff(p2, p0) - beta) >> 31;
    const int tc = tc0 - map - maq;
    const int dl = clip3(-tc, tc, (((q0 - p0) << 2) + (p1 - q1) + 4) >> 3);
    const int np1 = p1 + clip3(-tc0, tc0, (p2 + avg - (p1 << 1)) >> 1);
    s[3] = bsel(mf, clip255(p0 + dl), p0); s[4] = bsel(mf, clip255(q0 - dl), q0);
    s[2] = bsel(mf & map, np1, p1); s[5] = q1f;
}
DEV void edge_chroma_p(const unsigned par, int p1, int &p0, int &q0, int q1) {
    const int bS = (int)(par & 15u), alpha = (int)((par >> 8) & 255u), beta = (int)((par >> 16) & 255u), tc = (int)(par >> 24) + 1;
    const bool f = (bS != 0) & (adiff(p0, q0) < alpha) & (adiff(p1, p0) < beta) & (adiff(q1, q0) < beta);
    const int dl = clip3(-tc, tc, (((q0 - p0) << 2) + (p1 - q1) + 4) >> 3);
    const bool s4 = bS == 4;
    const int sp0 = (2 * p1 + p0 + q1 + 2) >> 2, sq0 = (2 * q1 + q0 + p1 + 2) >> 2, wp0 = clip255(p0 + dl), wq0 = clip255(q0 - dl);
    const int np0 = s4 ? sp0 : wp0, nq0 = s4 ? sq0 : wq0;
    p0 = f ? np0 : p0; q0 = f ? nq0 : q0;
}
// {bS, alpha, beta, tc0[bS]} of one lane's edge: nib = the lane's bS nibble, ab / tcw = the edge class's packed parameters
DEV unsigned par_word(unsigned nib, unsigned ab, unsigned tcw) { return nib | (ab << 8) | (((tcw >> (8 * ((nib - 1) & 3))) & 255u) << 24); }

// GATED: intra_p_kernel of the same picture may still be running (on another stream).  Its progress word of a macroblock row
// says how many leading macroblocks of the row are final (it stores their samples sc1 and drains before it publishes); the
// mover loads macroblock x only when x + 1 is below that mark as well (or the row is complete), and takes an acquire fence whenever it has read a new mark (a 128-byte line
// holds eight macroblocks of a line: an earlier load may have cached bytes that were not final yet).  That is all the
// ordering there is to keep: intra prediction reads the line above and the column to the left of a macroblock from the
// picture, and this kernel writes a macroblock's lines 0..11 once its row is three macroblocks further, its bottom lines
// once the ROW BELOW is -- by which time the marks say that the intra macroblocks that read them are done.
template <bool CHROMA, bool ALL_INTRA, int ROWS, bool GATED>
// part: -1 the band whole; 0 / 1 (a.part_cnt set): the band's left / right part, cut at a column where the filter does nothing (see "the cut" below)
DEV void rows3_body(const db_args &a, const int band, const int nb, uint8_t *lds, const int part) {
    constexpr int rows_mb = CHROMA ? 8 : 16, strip = CHROMA ? 2 : 4, ring_n = CHROMA ? 8 : 16;
    constexpr int ROW_LDS = CHROMA ? (int)sizeof(dbt_chroma) : (int)sizeof(dbt_luma);
    constexpr int TILE = rows_mb * 16, UPB = strip * 16;
    const frame_ctx_t *__restrict__ ctx = &a.ctx;
    const int mbw = ctx->mbw, mbh = ctx->mbh, stride = ctx->stride;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int role = wid / ROWS, r = wid - role * ROWS; // 0: filter, 1: mover (loads), 2: storer
    const int my = band * ROWS + r;
    // Slice-local deblocking (ctx->slice_dbf == 2; slices are whole bands): a slice's first row has no top edge to filter and takes nothing from the band above
    // (dbtop false: like the picture's first row), its last row keeps its bottom lines to itself (last_row true: like the picture's last row) -- every slice
    // is a wavefront of its own, and the launch is as long as the longest of them.
    const bool row_ok = my < mbh, last_row = my == mbh - 1 || db_slice_last(ctx, my), dbtop = db_has_top(ctx, my);
    const bool fed = row_ok && r == 0 && dbtop;
    const bool feeds = row_ok && r == ROWS - 1 && !last_row;
    // ---- prologue: per macroblock of this band, the parameter word {bS, alpha, beta, tc0[bS]} of every (edge, segment) of either
    // direction -- 32 words for luma, 16 for chroma -- into a table in global memory (it stays in L2; the movers bring a
    // macroblock's words in with the same load instruction as its samples), and whether this band and its two neighbours have
    // any edge to filter (the neighbours evaluate that for themselves: no workgroup waits for another).  A band without work -- the
    // still part of a live picture -- leaves at once: its samples are final as they are, and its neighbours take its strips
    // straight from the picture.  (The movers used to derive the words from the 64-byte record, ~50 instructions per macroblock on
    // the SIMD whose issue rate sets the band's pace.)
    if (!ALL_INTRA && a.row_done) {
        // The fused P stage of the same picture may still be running (pmb_kernel<GATED>, on another stream; this launch sits directly behind
        // the previous picture's): it stores samples and records through to memory and then counts each macroblock for its row.  This
        // band reads the records of its own rows and of the two neighbouring bands' (their work flags), and the samples of its own: it
        // starts when those rows are complete -- the counts only grow, a.row_need is what they reach with this picture.  One row more at
        // the top: the work flag of the band above comes from db_record() over ITS macroblocks, and the record of a macroblock in its first
        // row reads the macroblock above it -- the last row of the band above THAT.  (Without it this band could take a stale record there
        // for an intra macroblock, expect strips from a band that -- seeing the final record -- had nothing to filter and left at once,
        // and wait for them until the bound: error word 12, once in several thousand pictures at low rates, where bands are often idle.)
        if (threadIdx.x < 64) {
            const int r = (band - 1) * ROWS - 1 + (int)threadIdx.x;
            const bool mine = (int)threadIdx.x < 3 * ROWS + 1 && r >= 0 && r < mbh;
            const unsigned *w = a.row_done + (mine ? r * MI355_PROG_STRIDE : 0);
            int spins = 0;
            while (__ballot(mine && (int)(ld_sc1(w) - a.row_need) < 0)) {
                __builtin_amdgcn_s_sleep(8);
                if (++spins > DB_SPIN_MAX) { st_sc1(a.err, 17u | ((unsigned)band << 8) | (CHROMA ? 0x8000u : 0u)); break; } // bounded; the host reports the picture as failed
                if ((spins & 1023) == 0 && ld_sc1(a.err)) break;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
    }
    if (ALL_INTRA && a.iband_done) {
        // The intra band kernel of the same picture may still be running (on another stream): this band's macroblocks -- and the row
        // above them, whose records the boundary strengths read and whose bottom lines this band's top edge changes -- are complete
        // once the intra bands that hold macroblock rows band * ROWS - 1 .. band * ROWS + ROWS - 1 (the intra kernel's bands have
        // a.ib_rows rows) have published the picture's tag.  One lane polls, takes the acquire, the workgroup meets
        // (MI355X_MICROARCH.md, consumer form).
        if (threadIdx.x == 0) {
            const unsigned tag = ~ctx->epoch;
            const int r_lo = band * ROWS > 0 ? band * ROWS - 1 : 0, r_hi = band * ROWS + ROWS - 1 < mbh ? band * ROWS + ROWS - 1 : mbh - 1;
            int spins = 0;
            for (int ib = r_lo / a.ib_rows; ib <= r_hi / a.ib_rows; ib++)
                while (ld_sc1(a.iband_done + ib) != tag) {
                    __builtin_amdgcn_s_sleep(8);
                    if (++spins > DB_SPIN_MAX) { st_sc1(a.err, 11u); break; } if ((spins & 1023) == 0 && ld_sc1(a.err)) { break; } // bounded; the host reports the picture as failed
                }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
    }
    if (a.qpc) { // adaptive quantisation: the QP_Y chain has passed this band's rows (the row above them was in an earlier batch or is in this one)
        if (threadIdx.x == 0) {
            const unsigned need = a.qpc_base + (unsigned)((band + 1) * ROWS < mbh ? (band + 1) * ROWS : mbh);
            int spins = 0;
            while ((int)(ld_sc1(a.qpc) - need) < 0) {
                __builtin_amdgcn_s_sleep(8);
                if (++spins > DB_SPIN_MAX) { st_sc1(a.err, 23u | ((unsigned)band << 8)); break; }
                if ((spins & 1023) == 0 && ld_sc1(a.err)) break;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
    }
#ifdef TL_PROF
    if (!CHROMA && threadIdx.x == 0) { const int sl = band == 0 ? 9 : band == nb / 2 ? 11 : band == nb - 1 ? 13 : -1; if (sl >= 0) *((volatile unsigned long long *)ctx->dbrec + (ctx->epoch & 63) * 16 + sl) = wall_clock64(); }
#endif
    constexpr int PARW = CHROMA ? 16 : 32;                                  // parameter words per macroblock
    unsigned *partab = a.partab + (CHROMA ? (size_t)nb * ROWS * mbw * 32 : 0) + (size_t)band * ROWS * mbw * PARW;
    unsigned *flagw = (unsigned *)(lds + ROWS * ROW_LDS);   // [3]: band - 1, band, band + 1
    unsigned *firsti = flagw + 4;                           // [ROWS] (GATED): the first intra macroblock of each row of this band
    unsigned *busyw = firsti + ROWS;                        // [2]: columns around the middle whose left macroblock edge has something to filter in some row of this band
    if (threadIdx.x < 3) flagw[threadIdx.x] = 0;
    if (threadIdx.x < 2) busyw[threadIdx.x] = 0;
    if (GATED && threadIdx.x < ROWS) firsti[threadIdx.x] = (unsigned)mbw;
    // The cut: every edge of a row reads what the edge before it wrote -- except across a vertical macroblock edge with bS = 0 in every row of the band; there the left and
    // the right part share no filtered sample (column x0 behaves like a picture's first column, column x1 - 1 like its last), so two workgroups walk them side by side,
    // each against the band above's strips of its own columns.  Both evaluate the same records and choose the same column: the one nearest the middle, within a quarter
    // of the row; without one the left part walks the whole row and the right part has nothing to do.
    const int cut_mid = mbw >> 1, cut_w = (mbw >> 2) < 31 ? (mbw >> 2) : 31;
    __syncthreads();
    {
        const dev_tables *T = &g_tab;
        const int per = ROWS * mbw;
        for (int j = threadIdx.x; j < per; j += 192 * ROWS) { // this band
            const int row = band * ROWS + j / mbw;
            if (row < mbh) {
                unsigned w[16], pw[PARW];
                const bool work = db_record(ctx, T, j % mbw, row, w);
                if (GATED && (ldg32(&ctx->mbi[(size_t)row * mbw + j % mbw].mb_type) & 255u) != 1u) atomicMin(&firsti[j / mbw], (unsigned)(j % mbw));
#pragma unroll
                for (int d = 0; d < 2; d++)
#pragma unroll
                    for (int ed = 0; ed < (CHROMA ? 2 : 4); ed++)
#pragma unroll
                        for (int sg = 0; sg < 4; sg++) {
                            unsigned nib, ab, tc;
                            if (!CHROMA) { // record words 0..3: bS, two edges per word; 4..9: {alpha|beta<<8, tc0 bytes} left, top, inner
                                nib = (w[2 * d + (ed >> 1)] >> (16 * (ed & 1) + 4 * sg)) & 15u;
                                ab = ed == 0 ? w[4 + 2 * d] : w[8]; tc = ed == 0 ? w[5 + 2 * d] : w[9];
                            } else {       // chroma edges are luma edges 0 and 2 (the low halves of the bS words); parameters at words 10..15: left, top, inner
                                nib = (w[2 * d + ed] >> (4 * sg)) & 15u;
                                ab = ed == 0 ? w[10 + 2 * d] : w[14]; tc = ed == 0 ? w[11 + 2 * d] : w[15];
                            }
                            pw[d * (PARW / 2) + ed * 4 + sg] = par_word(nib, ab, tc);
                        }
#pragma unroll
                for (int q = 0; q < PARW / 4; q++) stg128(partab + (size_t)j * PARW + 4 * q, make_uint4(pw[4 * q], pw[4 * q + 1], pw[4 * q + 2], pw[4 * q + 3]));
                if (work) flagw[1] = 1u;
                if (part >= 0) {
                    const int kc = j % mbw - (cut_mid - cut_w);
                    if (kc >= 0 && kc <= 2 * cut_w && ((pw[0] | pw[1] | pw[2] | pw[3]) & 15u) != 0) atomicOr(&busyw[kc >> 5], 1u << (kc & 31)); // (vertical, edge 0, the four segments)
                }
            }
        }
#ifdef TL_PROF
        if (!CHROMA && band == 0 && threadIdx.x == 0) *((volatile unsigned long long *)ctx->dbrec + (ctx->epoch & 63) * 16 + 14) = wall_clock64(); // band 0: its own records are on their way
#endif
        if (!ALL_INTRA) {
            // the neighbours: "is there an intra macroblock or a coded luma block" (=> an inner edge with bS >= 2) settles it nearly always, with one load per macroblock;
            // only a band of nothing but bare vectors needs the records' own test (vector differences across edges)
            for (int i = threadIdx.x; i < 2 * per; i += 192 * ROWS) {
                const int which = i < per ? 0 : 2, j = i < per ? i : i - per;
                const int row = (band - 1 + which) * ROWS + j / mbw;
                if (row >= 0 && row < mbh) {
                    const mb_info_t m = ld_mbinfo(&ctx->mbi[(size_t)row * mbw + j % mbw]);
                    if (m.mb_type != 1 || (m.nzmask & 0xFFFFu) != 0) flagw[which] = 1u; // (luma blocks only: chroma coefficients raise no boundary strength)
                }
            }
            __syncthreads();
            for (int which = 0; which < 3; which += 2)
                if (flagw[which] == 0) // (workgroup-uniform)
                    for (int j = threadIdx.x; j < per; j += 192 * ROWS) {
                        const int row = (band - 1 + which) * ROWS + j / mbw;
                        unsigned w[16];
                        if (row >= 0 && row < mbh && db_record(ctx, T, j % mbw, row, w)) flagw[which] = 1u;
                    }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the table's stores have left this CU before its movers load them
    __syncthreads();
    // This band's lines are final in memory: tell the next picture's P stage, which may be running already (pmb_kernel's gate).  Release pattern: every storing wave
    // drains, the workgroup meets, one lane writes this XCD's L2 back and publishes the epoch.  A band walked in two parts is done when both are: each part counts itself
    // (an agent-scope add BEHIND its own release), and the one that finds the count odd -- the other part's add, hence its release, came first -- publishes for both.
    auto publish = [&](const bool drained) {
        if (!a.band_done) return;
        if (!drained) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }
        if (part >= 0) {
            if (threadIdx.x == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                flagw[3] = __hip_atomic_fetch_add(a.part_cnt + 2 * band + (CHROMA ? 1 : 0), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 1u;
            }
            __syncthreads();
            if (flagw[3] == 0) return; // (workgroup-uniform) the other part is still at work: it will publish
        }
        if (threadIdx.x < DB_DONE_COPIES) { // (one copy per poller group: a few thousand waves reading one word would make its memory channel a hot spot)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            st_sc1(a.band_done + DB_DONE_STRIDE * threadIdx.x + 2 * band + (CHROMA ? 1 : 0), ctx->epoch);
        }
    };
    // (the cut of every band is left for the band below: {column, picture epoch}, one granule per band and plane behind the parts' counters)
    uint2 *cutg = part >= 0 ? (uint2 *)(a.part_cnt + 2 * nb) + 2 * band + (CHROMA ? 1 : 0) : nullptr;
    if (!ALL_INTRA && flagw[1] == 0) { // nothing written here: the samples were final at the previous kernel boundary
        if (cutg && threadIdx.x == 0) st64_sc1(cutg, make_uint2(0u, ctx->epoch)); // (nobody below waits for this band's strips: no constraint)
        publish(true);
        return;
    }
    int x0 = 0, x1 = mbw; // the columns this workgroup walks
    if (part >= 0) {
        int cut = mbw; // (workgroup-uniform: every thread reads the same words)
        // The band above's cut: this band's must not lie to the left of it (below).  It is there as soon as that band's prologue is over -- the two run side by side.
        int cut_up = 0;
        if (!ALL_INTRA && db_has_top(ctx, band * ROWS) && flagw[0] != 0) {
            if (threadIdx.x == 0) {
                uint2 g;
                int spins = 0;
                while ((g = ld64_sc1(cutg - 2)).y != ctx->epoch) {
                    __builtin_amdgcn_s_sleep(4);
                    if (++spins > DB_SPIN_MAX) { st_sc1(a.err, 24u | ((unsigned)band << 8) | (CHROMA ? 0x8000u : 0u)); g.x = 0; break; } // bounded; the host reports the picture as failed
                    if ((spins & 1023) == 0 && ld_sc1(a.err)) { g.x = 0; break; }
                }
                flagw[3] = g.x;
            }
            __syncthreads();
            cut_up = (int)flagw[3];
            __syncthreads(); // (flagw[3] is used again by publish)
        }
        if (!ALL_INTRA && mbw >= 16) {
            // The free column nearest to the middle that is not to the LEFT of the band above's cut: the right part of a band can only start where the band above has
            // published its strips, and a cut to the left of that band's leaves its first columns to that band's left part -- which reaches them last.  With cuts
            // that never step left the left parts follow each other and the right parts follow each other at the usual distance.
            int tgt = cut_w - (cut_w >> 3); // bit position in the window: a little left of the middle for a band that has nothing to keep to (the cuts below it can only step right)
            const int lo = (cut_up > 0 && cut_up < mbw) ? cut_up - (cut_mid - cut_w) : 0; // (the band above walked whole, or idle: nothing to keep to)
            tgt = tgt > lo ? tgt : lo;
            const unsigned long long busy = (unsigned long long)busyw[0] | ((unsigned long long)busyw[1] << 32);
            for (int d = 0; d <= 2 * cut_w && cut == mbw; d++) {
                if (tgt + d <= 2 * cut_w && !((busy >> (tgt + d)) & 1ull)) cut = cut_mid - cut_w + tgt + d;
                else if (tgt - d >= lo && tgt - d >= 0 && !((busy >> (tgt - d)) & 1ull)) cut = cut_mid - cut_w + tgt - d;
            }
        }
        cut = __builtin_amdgcn_readfirstlane(cut); // an SGPR: read out of LDS the column is a vector value to the compiler, and every test against it in the step loops below a vector compare with exec-mask control flow (measured: the steps twice as long)
        if (threadIdx.x == 0) st64_sc1(cutg, make_uint2((unsigned)cut, ctx->epoch)); // (both parts write the same granule)
        if (part == 0) x1 = cut; else x0 = cut;
        if (x0 >= x1) { publish(true); return; } // (the right part of a band without a cut)
    }
#ifdef TL_PROF
    if (!CHROMA && band == 0 && threadIdx.x == 0) *((volatile unsigned long long *)ctx->dbrec + (ctx->epoch & 63) * 16 + 15) = wall_clock64(); // band 0: the prologue is over
#endif
    const bool up_work = ALL_INTRA || flagw[0] != 0; // the band above publishes its strips
    const bool dn_work = ALL_INTRA || flagw[2] != 0; // ... and the band below reads ours
    const unsigned epoch = ctx->epoch;
    uint2 *gran_up = a.gran + (CHROMA ? (size_t)nb * mbw * 16 : 0) + (size_t)(band > 0 ? band - 1 : 0) * mbw * ring_n;
    uint2 *gran_my = a.gran + (CHROMA ? (size_t)nb * mbw * 16 : 0) + (size_t)band * mbw * ring_n;
    uint8_t *__restrict__ plane = CHROMA ? ctx->rec_uv : ctx->rec_y;
    const size_t row0 = (size_t)my * rows_mb;
    uint8_t *rowl = lds + r * ROW_LDS;
    uint8_t *tiles = rowl;                                        // [DBT_NB][TILE]
    uint8_t *ups = rowl + DBT_NB * TILE;                          // [DBT_NB][UPB]: the strip above, first row of a band only
    unsigned *pars = (unsigned *)(rowl + DBT_NB * (TILE + UPB));  // [4][PARW]: the parameter words of the macroblocks in flight
    uint8_t *tiles_up = lds + (r > 0 ? r - 1 : 0) * ROW_LDS;      // the row above's tiles
    const int keep = last_row ? rows_mb : rows_mb - strip;
    const int t_end = (x1 - x0) + ROWS + 1;
    // ---- per-role state
    const int k = lane >> 2, e = lane & 3;                        // filter lanes (luma): line / column k, edge e
    uint4 preA, preB; // mover: two macroblock loads in flight.  Deliberately not initialised: a phi with a constant at the loop header
    uint2 gA, gB;     // costs a register copy on the back edge, and a copy of a register with a load in flight waits for the load
    const bool rlane = lane < rows_mb, uplane = lane >= 16 && lane < 16 + strip, glane = lane >= 32 && lane < 32 + ring_n;
    // every lane of the mover loads, every step, from a clamped address (lanes without a role repeat a record quarter): a load
    // under a branch or a predicate is a conditional assignment to a loop-carried value, which costs a merge move -- and the wait
    // for the load right behind its issue
    const int my_c = row_ok ? my : mbh - 1;
    constexpr int PL = PARW / 4;                                  // lanes 16 .. 16+PL-1 bring the macroblock's parameter words (16 bytes each)
    const bool plane_l = lane >= 16 && lane < 16 + PL;
    const uint8_t *ld_ptr = plane_l ? (const uint8_t *)(partab + (size_t)r * mbw * PARW + 4 * (lane - 16))
                                    : plane + ((size_t)my_c * rows_mb + (lane & (rows_mb - 1))) * stride; // (lanes without a role repeat a line)
    const int ld_step = plane_l ? PARW * 4 : 16;
    uint8_t *st_ptr = lane < 16 ? plane + ((size_t)my_c * rows_mb + lane) * stride : plane + ((size_t)my_c * rows_mb - (my_c > 0 ? strip : 0) + (lane & 3)) * stride; // storer: this lane's line
    const int gj = glane ? lane - 32 : 0;
    const uint8_t *g_ptr = up_work ? (const uint8_t *)(gran_up + gj) : plane + ((size_t)my_c * rows_mb - (my_c > 0 ? strip : 0) + (gj >> 2)) * stride + 4 * (gj & 3);
    const int g_step = up_work ? ring_n * 8 : 16;
#ifdef DBT_PROF /* debug builds: cycles before the barrier, in it, after it, per role; row 1 of band 1 */
    unsigned long long pc[3] = {0, 0, 0}, tm0 = 0, tm1;
    unsigned nmiss = 0;
#define DBT_TICK(i) do { tm1 = __builtin_readcyclecounter(); pc[i] += tm1 - tm0; tm0 = tm1; } while (0)
#define DBT_T0() tm0 = __builtin_readcyclecounter()
#else
#define DBT_TICK(i) do { } while (0)
#define DBT_T0() do { } while (0)
#endif
    const int t_last = t_end | 1; // every role runs the same, even number of steps (the mover's loop is unrolled by two)
    // Each role runs its own loop (one barrier per step in each): compiled as one loop with a role switch inside, the three
    // roles share a register assignment at the back edge, and the merge moves there wait for the mover's loads in flight and for
    // the storer's stores.
    if (role == 0) {
        for (int t = -2; t <= t_last; t++) {
            DBT_T0();
            const int x = x0 + t - 1 - r;
            const bool act = row_ok && x >= x0 && x < x1;
            // =========================================================== F: vertical edges | barrier | horizontal edges
                uint8_t *tile = tiles + (x & (DBT_NB - 1)) * TILE;
                if (act) {
                    const unsigned par = pars[(x & 3) * PARW + (CHROMA ? (lane & 1) * 4 + (lane >> 3) : e * 4 + (k >> 2))]; // vertical: (edge, segment of this lane's line)
                    if (!CHROMA) {
                        uint8_t *tl = tiles + ((x - 1) & (DBT_NB - 1)) * TILE;
                        unsigned *pw = (unsigned *)(e == 0 ? tl + k * 16 + 12 : tile + k * 16 + 4 * e - 4), *qw = (unsigned *)(tile + k * 16 + 4 * e);
                        const unsigned w0 = *pw, w1 = *qw;
                        const unsigned long long work = __ballot((par & 15u) != 0);
                        if (ALL_INTRA || work) {
                            int s[8];
#pragma unroll
                            for (int i = 0; i < 4; i++) { s[i] = byte_of(w0, i); s[4 + i] = byte_of(w1, i); }
                            edges4p_luma<ALL_INTRA>(e, par, __ballot((par & 15u) == 4u) != 0, s);
                            const int nx = quad_next(s[2] | (s[3] << 8));
                            const unsigned hi = e == 3 ? (unsigned)(s[6] | (s[7] << 8)) : (unsigned)nx;
                            *qw = (unsigned)s[4] | ((unsigned)s[5] << 8) | (hi << 16);
                            if (e == 0 && x > x0) *pw = pack4(s[0], s[1], s[2], s[3]);
                        }
                    } else if (lane < 32) {
                        const int kk = lane >> 2, c = (lane >> 1) & 1, ee = lane & 1; // line, plane, edge (luma edges 0 and 2)
                        uint8_t *tl = tiles + ((x - 1) & (DBT_NB - 1)) * TILE;
                        uint8_t *pb = ee == 0 ? tl + kk * 16 + 12 + c : tile + kk * 16 + 4 + c, *qb = tile + kk * 16 + 8 * ee + c;
                        int p1 = pb[0], p0 = pb[2], q0 = qb[0], q1 = qb[2];
                        edge_chroma_p(par, p1, p0, q0, q1);
                        if (ee != 0 || x > x0) pb[2] = (uint8_t)p0;
                        qb[0] = (uint8_t)q0;
                    }
                }
                DBT_TICK(0);
                BAND_BARRIER();
                DBT_TICK(1);
                if (act) {
                    const unsigned par = pars[(x & 3) * PARW + PARW / 2 + (CHROMA ? (lane & 1) * 4 + (lane >> 3) : e * 4 + (k >> 2))]; // horizontal: (edge, segment of this lane's column)
                    uint8_t *upb = fed ? ups + (x & (DBT_NB - 1)) * UPB : tiles_up + (x & (DBT_NB - 1)) * TILE + (rows_mb - strip) * 16;
                    if (!CHROMA) {
                        uint8_t *pb = (e == 0 ? upb : tile + (4 * e - 4) * 16) + k, *qb = tile + 4 * e * 16 + k;
                        int s[8];
#pragma unroll
                        for (int i = 0; i < 4; i++) { s[i] = pb[i * 16]; s[4 + i] = qb[i * 16]; }
                        const unsigned long long work = __ballot((par & 15u) != 0);
                        if (ALL_INTRA || work) {
                            const bool any4 = __ballot((par & 15u) == 4u) != 0;
                            edges4p_luma<ALL_INTRA>(e, par, any4, s);
                            if (e != 0 || dbtop) { pb[2 * 16] = (uint8_t)s[2]; pb[3 * 16] = (uint8_t)s[3]; }
                            qb[0] = (uint8_t)s[4]; qb[16] = (uint8_t)s[5];
                            if ((ALL_INTRA || any4) && e == 0 && dbtop) pb[16] = (uint8_t)s[1];
                        }
                    } else if (lane < 32) {
                        const int kb = lane >> 1, ee = lane & 1; // byte column, edge
                        uint8_t *pb = (ee == 0 ? upb : tile + 2 * 16) + kb, *qb = tile + 4 * ee * 16 + kb;
                        int p1 = pb[0], p0 = pb[16], q0 = qb[0], q1 = qb[16];
                        edge_chroma_p(par, p1, p0, q0, q1);
                        if (ee != 0 || dbtop) pb[16] = (uint8_t)p0;
                        qb[0] = (uint8_t)q0;
                    }
                }
            DBT_TICK(2);
        }
    } else if (role == 1) {
        // `cur` is the load set of this step's parity (two macroblock loads are in flight: the loop is unrolled by two so that
        // both sets are plain registers -- a set chosen by `t & 1` lives in scratch, and its load gets waited for at once)
        // macroblocks of this row known to be final: everything left of the row's first intra macroblock is (the fused stage wrote it; what
        // intra_p_kernel publishes first says the same, but that kernel may not have started yet)
        int fin = GATED ? (int)firsti[r] : 0x7FFF;
        auto mstep = [&](const int t, uint4 &cur, uint2 &gpre) __attribute__((always_inline)) {
            DBT_T0();
            const int x = x0 + t - 1 - r;
            const bool act = row_ok && x >= x0 && x < x1;
            // =========================================================== M: land macroblock x+1 | barrier | load macroblock x+3 (and the strip above x+1)
                const int xm = x + 1, xl = x + 3;
                const bool lands = row_ok && xm >= x0 && xm < x1;
                if (lands) {
                    if (rlane) *(uint4 *)(tiles + (xm & (DBT_NB - 1)) * TILE + lane * 16) = cur;
                    if (plane_l) *(uint4 *)(pars + (xm & 3) * PARW + 4 * (lane - 16)) = cur; // ... and its parameter words
                }
                if (fed && act) { // the strip above macroblock x, asked for two steps ago: every granule must carry this picture's epoch
                    // the first test stands outside the retry loop: a loop that reloads `gpre` makes the compiler wait for every load in
                    // flight at its header (vmcnt(0): also the loads issued half a step ago), the common case needs only the one from two steps ago
                    unsigned *dst = (unsigned *)(ups + (x & (DBT_NB - 1)) * UPB) + gj;
                    // The prefetched word is stored unconditionally; a strip that had not been published yet when it was asked for is
                    // fetched again into variables of its own and stored over it.  (Any form in which the two values meet in one store --
                    // a select, or two stores the compiler can merge -- costs a copy of gpre.x on the loop's back edge, which waits for the
                    // sc1 load issued right there: a full cross-XCD round trip per two steps, on every mover.)
                    const bool late = up_work && __ballot(glane && gpre.y != epoch) != 0;
                    if (glane) *dst = gpre.x;
                    if (late) {
                        uint2 g2;
                        int spins = 0;
#ifdef DBT_PROF
                        nmiss++;
#endif
                        do {
                            __builtin_amdgcn_s_sleep(1);
                            g2 = ld64_sc1(gran_up + (size_t)x * ring_n + gj);
                            if (++spins > DB_SPIN_MAX) { st_sc1(a.err, 12u | ((unsigned)band << 8) | (CHROMA ? 0x8000u : 0u) | ((unsigned)x << 16) | ((unsigned)((__ballot(glane && g2.y == epoch) >> 32) & 0xFu) << 28)); break; } if ((spins & 1023) == 0 && ld_sc1(a.err)) { break; } // bounded; once tripped, nobody waits again
                        } while (__ballot(glane && g2.y != epoch));
                        if (glane) *dst = g2.x;
                    }
                }
                DBT_TICK(0);
                BAND_BARRIER();
                DBT_TICK(1);
                {
                    const int xg = x + 2, xgc = xg < 0 ? 0 : (xg < mbw ? xg : mbw - 1), xlc = xl < 0 ? 0 : (xl < mbw ? xl : mbw - 1);
                    // macroblock xl may be loaded (and, a step later, filtered) once it AND its right neighbour are final: filtering xl's horizontal
                    // edges changes its right column, and its top edge the sample above that column's top -- the left column and the corner an intra
                    // macroblock at xl + 1 predicts from.  (The first form of the test was `> xl`: safe only as long as intra_p_kernel has a head start.)
                    const int need = xl + 2 < mbw ? xl + 2 : mbw;
                    if (GATED && row_ok && xl >= x0 && xl < x1 && fin < need) {
                        const unsigned tag = (epoch & 0xFFFFFu) << 12; // IP_EPOCH of k_intra.hip
                        int spins = 0;
                        for (;;) {
                            const unsigned v = (unsigned)__builtin_amdgcn_readfirstlane((int)ld_sc1(a.ip_progress + my * MI355_PROG_STRIDE));
                            if ((v & ~0xFFFu) == tag && (int)(v & 0xFFFu) >= need) { fin = (int)(v & 0xFFFu); break; }
                            __builtin_amdgcn_s_sleep(2);
                            if (++spins > DB_SPIN_MAX) { st_sc1(a.err, 13u); fin = 0x7FFF; break; } if ((spins & 1023) == 0 && ld_sc1(a.err)) { fin = 0x7FFF; break; } // bounded; once tripped, nobody waits again
                        }
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    }
                    cur = ldg128(ld_ptr + (size_t)xlc * ld_step); // first: memory operations return in order, and the granule load below (sc1, another XCD's data) takes about two steps
                    gpre = ld64_sc1((const uint2 *)(g_ptr + (size_t)xgc * g_step)); // the strip above macroblock x+2, two steps ahead like the macroblocks: a cross-XCD round trip is longer than a step (first row of a band; elsewhere unused)
                }
            DBT_TICK(2);
        };
        for (int t = -2; t <= t_last; t += 2) { mstep(t, preA, gA); mstep(t + 1, preB, gB); }
    } else {
        for (int t = -2; t <= t_last; t++) {
            DBT_T0();
            const int x = x0 + t - 1 - r;
            // =========================================================== S: store macroblock x-2 | barrier | publish the strip of macroblock x-1
            // Lines 0 .. keep-1 of macroblock x-2 and the strip above it became final with the barrier of the step before (this row's
            // vertical phase of x-1 patched its columns 12..15; the horizontal phase of x-2 finished the strip above): one LDS read
            // and one 16-byte store per lane, per-lane addresses.
            const int xs2 = x - 2;
#ifdef DBG_DELAY_BAND /* adversarial-schedule build: the band's last four macroblocks of every row are written back long after its loop has ended (below) */
            if (row_ok && xs2 >= x0 && xs2 < x1 && !(band == DBG_DELAY_BAND && xs2 >= mbw - 4 && lane < keep)) {
#else
            if (row_ok && xs2 >= x0 && xs2 < x1) {
#endif
                const bool is_row = lane < keep, is_up = uplane && dbtop;
                const uint8_t *upb = fed ? ups + (xs2 & (DBT_NB - 1)) * UPB : tiles_up + (xs2 & (DBT_NB - 1)) * TILE + (rows_mb - strip) * 16;
                const uint8_t *src = is_row ? tiles + (xs2 & (DBT_NB - 1)) * TILE + lane * 16 : upb + (lane & 3) * 16;
                if (is_row || is_up) stg128(st_ptr + xs2 * 16, lds128(src));
            }
            DBT_TICK(0);
            BAND_BARRIER();
            DBT_TICK(1);
            const int xs = x - 1; // its bottom strip is final within this band now: the band below waits for it
            if (feeds && xs >= x0 && xs < x1 && glane) {
                const int j = lane - 32;
                const unsigned w = *(const unsigned *)(tiles + (xs & (DBT_NB - 1)) * TILE + (rows_mb - strip) * 16 + 4 * j);
                if (dn_work) st64_sc1(gran_my + (size_t)xs * ring_n + j, make_uint2(w, epoch));
                else stg32(plane + (row0 + rows_mb - strip + (j >> 2)) * stride + xs * 16 + 4 * (j & 3), w); // nobody below will store it
            }
            DBT_TICK(2);
        }
    }
    // ---- this band's lines are final in memory (see publish above)
    if (a.band_done) {
#ifdef DBG_DELAY_BAND /* adversarial-schedule build (tests/test_adversarial_gpu.py): this band's last macroblocks reach memory ~0.3 ms after its loop has
                         ended -- long after the bands below it, which only needed its strips, have published: a reader of the next picture that
                         checked the lowest band of its window alone (round 2's first form of pmb_kernel's gate) reads lines that are not there yet */
        if (band == DBG_DELAY_BAND) {
            for (int i = 0; i < 100; i++) __builtin_amdgcn_s_sleep(127); // 100 x 127 x 64 cycles
            if (role == 2 && row_ok && lane < keep)
                for (int xs2 = mbw - 4 > 0 ? mbw - 4 : 0; xs2 < mbw; xs2++) stg128(st_ptr + xs2 * 16, lds128(tiles + (xs2 & (DBT_NB - 1)) * TILE + lane * 16));
        }
#endif
        publish(false);
    }
    tl_last(ctx, 8);
#ifdef TL_PROF
    if (!CHROMA && threadIdx.x == 0) { const int sl = band == 0 ? 10 : band == nb / 2 ? 12 : -1; if (sl >= 0) *((volatile unsigned long long *)ctx->dbrec + (ctx->epoch & 63) * 16 + sl) = wall_clock64(); }
#endif
#ifdef DBT_PROF
    if (lane == 0 && band < 2) { // rows 0..3 of bands 0 and 1
        unsigned *o = (unsigned *)(ctx->dbrec) + (CHROMA ? 128 : 0) + 64 * band + 16 * r + 4 * role; // debug build only: overwrites the first records after use
        for (int i = 0; i < 3; i++) o[i] = (unsigned)pc[i];
        o[3] = (unsigned)(t_last + 3) | (nmiss << 16);
    }
#endif
}

// Adaptive quantisation: the QP_Y of a macroblock that sends no mb_qp_delta is that of the macroblock before it in decoding order (7.4.5), and the
// filter reads it -- a chain over the whole picture (oracle: orc_qp_chain_slices).  As a kernel of its own between a picture's records and its
// deblocking launch it put the picture's stages in stream order; here it rides in the deblocking launch as its FIRST workgroup (placed before any
// band), walks the rows in batches of one row per wave behind the same progress the bands wait for -- pmb_kernel's row counts, the intra rows'
// flags, or nothing (stream order) -- rewrites the qp byte of the macroblocks without a delta and counts the rows it has resolved; a band starts
// when the count covers its rows and the row above them.  Intra_16x16 macroblocks always send a delta; an Intra_4x4 macroblock of a P picture does
// not when it has no coefficients, which is known only after intra_p_kernel: with cfg.intra_in_p = 2 the host keeps the stages in order.
template <int NW>
DEV void qp_chain_rows(const db_args &a) {
    __shared__ int row_last[NW], row_in[NW];
    const frame_ctx_t *__restrict__ ctx = &a.ctx;
    const int mbw = ctx->mbw, mbh = ctx->mbh, lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned tag = ~ctx->epoch;
    int carry = ctx->qp; // QP_Y,PRED at the start of a slice: the slice's QP
    for (int r0 = 0; r0 < mbh; r0 += NW) {
        const int r = r0 + w;
        const bool mine = r < mbh;
        if (mine && lane == 0) { // this wave's row is final: every macroblock counted (P pictures) / its intra band flagged (I pictures)
            int spins = 0;
            if (!ctx->all_intra && a.row_done)
                while ((int)(ld_sc1(a.row_done + r * MI355_PROG_STRIDE) - a.row_need) < 0) {
                    __builtin_amdgcn_s_sleep(8);
                    if (++spins > DB_SPIN_MAX) { st_sc1(a.err, 22u | ((unsigned)r << 8)); break; }
                    if ((spins & 1023) == 0 && ld_sc1(a.err)) break;
                }
            if (ctx->all_intra && a.iband_done)
                while (ld_sc1(a.iband_done + r / a.ib_rows) != tag) {
                    __builtin_amdgcn_s_sleep(8);
                    if (++spins > DB_SPIN_MAX) { st_sc1(a.err, 22u | ((unsigned)r << 8)); break; }
                    if ((spins & 1023) == 0 && ld_sc1(a.err)) break;
                }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // pass 1: the last QP that was sent in this row (-1: none)
        int last = -1;
        if (mine)
            for (int x0 = 0; x0 < mbw; x0 += 64) {
                const int x = x0 + lane;
                int q = -1;
                if (x < mbw) { const uint4 rec = ldg128(&ctx->mbi[(size_t)r * mbw + x]); if ((rec.y & 255u) == 0u || (rec.z & 0x07FFFFFFu) != 0u) q = (int)(rec.y >> 24); }
                const unsigned long long m = __ballot(q >= 0);
                if (m) last = __shfl(q, 63 - __builtin_clzll(m), 64);
            }
        if (lane == 0) row_last[w] = mine ? last : -1;
        __syncthreads();
        // the carry into every row of the batch (a slice starts again from the slice's QP)
        if (threadIdx.x == 0) {
            int c = carry;
            for (int k = 0; k < NW; k++) {
                const int rr = r0 + k;
                if (ctx->slice_rows > 0 && rr % ctx->slice_rows == 0) c = ctx->qp;
                row_in[k] = c;
                if (row_last[k] >= 0) c = row_last[k];
            }
            row_last[0] = c; // the carry out of the batch
        }
        __syncthreads();
        // pass 2: every macroblock without a delta takes the QP of the last one before it that sent one
        if (mine) {
            int prev = row_in[w];
            for (int x0 = 0; x0 < mbw; x0 += 64) {
                const int x = x0 + lane;
                int q = -1;
                uint4 rec = make_uint4(0, 0, 0, 0);
                if (x < mbw) { rec = ldg128(&ctx->mbi[(size_t)r * mbw + x]); if ((rec.y & 255u) == 0u || (rec.z & 0x07FFFFFFu) != 0u) q = (int)(rec.y >> 24); }
                const unsigned long long m = __ballot(q >= 0);
                const unsigned long long below = m & ((2ull << lane) - 1ull); // senders at or before this lane
                const int src = below ? 63 - __builtin_clzll(below) : 0;
                const int got = __shfl(q, src, 64);
                const int res = below ? got : prev;
                if (x < mbw && q < 0) stg32((unsigned *)&ctx->mbi[(size_t)r * mbw + x] + 1, (rec.y & 0x00FFFFFFu) | ((unsigned)res << 24));
                if (m) prev = __shfl(q, 63 - __builtin_clzll(m), 64);
            }
        }
        carry = row_last[0];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            st_sc1(a.qpc, a.qpc_base + (unsigned)(r0 + NW < mbh ? r0 + NW : mbh));
        }
        __syncthreads();
    }
}

// FUSED_IP: the intra macroblocks of the same P picture ride in this launch -- its first a.nip workgroups run intra_p_row (two waves each, the
// other waves end at once), one macroblock row each, behind the same row counts of pmb_kernel<GATED, ROWS> the bands wait for.  As a kernel of
// its own intra_p_kernel could only follow pmb_kernel in stream order, i.e. after its LAST row -- and with two deblocking launches in flight the
// bands of a launch that starts early would sit at every row's first intra macroblock until then (device timeline: band 0 done 280 us after
// the launch started, 120 us of them waiting).  In one launch, leading the grid, the rows are on the chip before any band is (workgroups are
// placed in index order), whatever else fills it; they wait for nothing but pmb_kernel's rows and the row above.
template <int ROWS, bool ALL_INTRA, bool GATED, bool FUSED_IP>
__global__ __launch_bounds__(192 * ROWS) void deblock_rows3_kernel(db_args a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[]; // ROWS rows of tile rings, then three work flags and the rows' first intra macroblocks
    int bi = (int)blockIdx.x, nwg = (int)gridDim.x;
    if (!FUSED_IP && a.qpc) { // (adaptive quantisation: the QP_Y chain leads the launch)
        if (bi == 0) { qp_chain_rows<3 * ROWS>(a); return; }
        bi -= 1; nwg -= 1;
    }
    if (FUSED_IP) {
        if (bi < a.nip) {
            if (threadIdx.x >= 128) return;
            ip_args ia;
            ia.ctx = a.ctx; ia.progress = a.ip_progress_w; ia.strips = a.ip_strips; ia.err = a.err;
            intra_p_row(ia, bi, a.row_done, a.row_need);
            // records and levels of this row's intra macroblocks are read by the hand-over kernels of another stream while this launch still
            // runs: drain, meet, write this XCD's L2 back, then count the row as done (the host's wait kernel follows the count)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_fetch_add(a.ip_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            return;
        }
        bi -= a.nip; nwg -= a.nip;
    }
    if (bi == 0) tl_first(&a.ctx, 7);
    if (a.started && threadIdx.x == 0) __hip_atomic_fetch_add(a.started, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // this workgroup holds its place on a CU (wait_started_kernel)
    int part = -1;
    if (!ALL_INTRA && a.part_cnt) { part = bi & 1; bi >>= 1; nwg >>= 1; } // two workgroups per band: its left and its right part (rows3_body: the cut)
    const int nl = nwg >> 1;
    if (bi < nl) rows3_body<false, ALL_INTRA, ROWS, GATED>(a, a.band0 + bi, a.nb_total, lds, part);
    else rows3_body<true, ALL_INTRA, ROWS, GATED>(a, a.band0 + bi - nl, a.nb_total, lds, part);
}

// =================================================================== launchers
int k_deblock_diags(int mbw, int mbh) { return mbw + 2 * (mbh - 1); }
void k_launch_deblock_diag(const frame_ctx_t *d_ctx, int mbw, int mbh, int diag, hipStream_t s) {
    int y_lo = diag - (mbw - 1);
    y_lo = y_lo > 0 ? (y_lo + 1) >> 1 : 0;
    int y_hi = diag / 2 < mbh - 1 ? diag / 2 : mbh - 1;
    if (y_hi < y_lo) return;
    hipLaunchKernelGGL(deblock_kernel, dim3(y_hi - y_lo + 1), dim3(64), 0, s, d_ctx, diag);
}
#ifndef DB_CUT_MIN_MBW
#define DB_CUT_MIN_MBW 60 /* rows at least this long are walked in two parts (A/B, alternating processes: 2160p +6...9 %, 1080p +2 %, 720p +5 %, 640 x 368 -3 %: two workgroups' prologues cost more than twenty columns save) */
#endif
#ifndef DB_ROWS
#define DB_ROWS MI355_BAND_ROWS /* rows per band: three waves per row, one row per SIMD (5 rows = 15 waves is equal on P pictures, a third slower on I pictures) */
#endif
int k_deblock_bands16(int mbh) { return (mbh + DB_ROWS - 1) / DB_ROWS; }
size_t k_deblock_gran_bytes(int mbw, int mbh) { return (size_t)k_deblock_bands16(mbh) * mbw * 24 * sizeof(uint2); } // per band boundary and macroblock: 16 luma + 8 chroma granules
// The band kernel may be launched in several pieces (bands [band0, band1)): a band only ever waits for the band above it.
// d_ip_progress (may be null): intra_p_kernel of the same picture is still running; the movers follow its per-row progress words.
template <typename K>
static int launch_bands(K kernel, const db_args &a, int nbands, int mbw, hipStream_t s) {
    constexpr size_t lds = (size_t)DB_ROWS * sizeof(dbt_luma) + 16 + 4 * DB_ROWS + 8; // the rows' tile rings + three work flags (and the parts' hand-shake word) + the rows' first intra macroblocks + the cut's column mask: ~12.5 KB
    static_assert(lds <= 48 * 1024, "above 48 KB of dynamic LDS every instantiation launched here would need hipFuncAttributeMaxDynamicSharedMemorySize");
    const int wgs = (a.part_cnt ? 4 : 2) * nbands;
    hipLaunchKernelGGL(kernel, dim3(wgs + (a.nip > 0 ? a.nip : 0) + (a.qpc && a.nip <= 0 ? 1 : 0)), dim3(192 * DB_ROWS), lds, s, a);
    return wgs; // the workgroups that count themselves in a.started
}
// One wave that ends once `count` workgroups of band-deblocking launches have been placed since the encoder was opened (the count only
// grows; the comparison is wrap-safe).  On a stream in front of a kernel whose workgroups wait for the deblocker's flags, it keeps
// them from filling the chip before the deblocker is on it.
__global__ __launch_bounds__(64) void wait_started_kernel(const unsigned *started, unsigned count, unsigned *err) {
    if (threadIdx.x) return;
    int spins = 0;
    while ((int)(ld_sc1(started) - count) < 0) {
        __builtin_amdgcn_s_sleep(32);
        if (++spins > DB_SPIN_MAX) { st_sc1(err, 4u); break; } // bounded; the host reports the picture as failed
        if ((spins & 255) == 0 && ld_sc1(err)) break;
    }
}
void k_launch_wait_started(const unsigned *d_started, unsigned count, unsigned *d_err, hipStream_t s) { hipLaunchKernelGGL(wait_started_kernel, dim3(1), dim3(64), 0, s, d_started, count, d_err); }
size_t k_deblock_done_bytes(void) { return (size_t)DB_DONE_COPIES * DB_DONE_STRIDE * sizeof(unsigned); } // 2 words per band: up to 512 bands
size_t k_deblock_partab_bytes(int mbw, int mbh) { return (size_t)k_deblock_bands16(mbh) * DB_ROWS * mbw * 48 * sizeof(unsigned); } // 32 luma + 16 chroma words per macroblock
int k_launch_deblock_bands(const frame_ctx_t *h_ctx, int mbh, int band0, int band1, unsigned *d_err, uint2 *d_gran, unsigned *d_partab, const unsigned *d_ip_progress,
                           const unsigned *d_iband_done, int ib_rows, unsigned *d_band_done, unsigned *d_started, const unsigned *d_row_done, unsigned row_need,
                           uint8_t *d_ip_strips, unsigned *d_ip_done, unsigned *d_qpc, unsigned qpc_base, unsigned *d_part_cnt, hipStream_t s) {
    db_args a;
    a.part_cnt = nullptr;
    a.ctx = *h_ctx; a.err = d_err; a.band0 = band0; a.nb_total = k_deblock_bands16(mbh); a.gran = d_gran; a.ip_progress = d_ip_progress; a.partab = d_partab; a.iband_done = d_iband_done; a.ib_rows = ib_rows > 0 ? ib_rows : DB_ROWS; a.band_done = d_band_done; a.started = d_started; a.row_done = d_row_done; a.row_need = row_need;
    a.nip = 0; a.ip_progress_w = nullptr; a.ip_strips = nullptr; a.ip_done = nullptr;
    a.qpc = (h_ctx->qp_off && band0 == 0) ? d_qpc : nullptr; a.qpc_base = qpc_base; // (a launch of all the picture's bands)
    if (band1 <= band0) return 0;
    if (h_ctx->all_intra) return launch_bands(deblock_rows3_kernel<DB_ROWS, true, false, false>, a, band1 - band0, h_ctx->mbw, s); // IDR pictures: every edge has work
#ifndef DBG_DELAY_BAND /* (the adversarial-schedule build delays a band's last columns by their position in the whole row) */
    if (band0 == 0 && band1 == a.nb_total && h_ctx->mbw >= DB_CUT_MIN_MBW) a.part_cnt = d_part_cnt; // P pictures, the whole picture in one launch: every band as two workgroups (rows3_body: the cut)
#endif
    if (d_ip_progress && d_ip_done && d_row_done && band0 == 0) { // the picture's intra macroblock rows lead the launch (pmb_kernel<GATED, ROWS> of the same picture still runs)
        a.nip = mbh; a.ip_progress_w = const_cast<unsigned *>(d_ip_progress); a.ip_strips = d_ip_strips; a.ip_done = d_ip_done;
        return launch_bands(deblock_rows3_kernel<DB_ROWS, false, true, true>, a, band1 - band0, h_ctx->mbw, s);
    }
    if (d_ip_progress) return launch_bands(deblock_rows3_kernel<DB_ROWS, false, true, false>, a, band1 - band0, h_ctx->mbw, s); // beside intra_p_kernel
    return launch_bands(deblock_rows3_kernel<DB_ROWS, false, false, false>, a, band1 - band0, h_ctx->mbw, s);
}
