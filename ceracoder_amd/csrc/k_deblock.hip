// k_deblock.hip -- in-loop deblocking filter (H.264 8.7): prep + persistent 16-row bands, and the per-diagonal form
// Hand-written HIP for gfx950 (CDNA4, wave64); part of libmi355enc (see kernels_common.hpp).
#include "kernels_common.hpp"

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
struct db_args { frame_ctx_t ctx; unsigned *progress; unsigned *err; int band0, nb_total; uint2 *gran; }; // a launch covers bands band0 .. band0 + gridDim.x/2 - 1

// =================================================================== deblocking, persistent: bands of rows in x + y order
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
// A band waits only on the band above it, so the wait graph is acyclic; every spin is bounded and reports through `err`.
#define DBREC_BYTES 64
DEV unsigned pack_par_ab(const dev_tables *T, int idx) { return (unsigned)T->alpha[idx] | ((unsigned)T->beta[idx] << 8); }
DEV unsigned pack_par_tc(const dev_tables *T, int idx) { return (unsigned)T->tc0[idx][0] | ((unsigned)T->tc0[idx][1] << 8) | ((unsigned)T->tc0[idx][2] << 16); }
DEV edge_par par_of(unsigned ab, unsigned tc) { edge_par p; p.alpha = (int)(ab & 255); p.beta = (int)(ab >> 8); p.tc0 = tc; return p; }

// One thread per macroblock: words 0-1 bS of the vertical edges (nibble 4*edge + segment), 2-3 of
// the horizontal edges, then {alpha|beta<<8, tc0 bytes} for luma left / top / inner and chroma
// left / top / inner.  Edges that are not filtered (picture border, 8x8-transform inner edges) get bS 0.
// Covers macroblocks [mb0, mb1); also zeroes two ranges of band progress counters (which ones: see the launch sites -- a
// counter must be zero before any kernel that polls it can start, so a launch never clears counters its own picture's bands
// are about to use unless everything else has been joined).
// `flags`: one word per band of this picture's set, raised if any macroblock of the band has an edge with bS != 0 -- a band
// without one has nothing to filter, and its workgroups publish "done" and leave at once (static parts of live pictures).
__global__ __launch_bounds__(256) void deblock_prep_kernel(const frame_ctx_t cv, unsigned *__restrict__ clr_a, int n_a, unsigned *__restrict__ clr_b, int n_b,
                                                           unsigned *__restrict__ clr_c, int n_c, unsigned *__restrict__ flags, int mb0, int mb1, int band_rows) {
    const frame_ctx_t *__restrict__ ctx = &cv;
    const int j = blockIdx.x * 256 + threadIdx.x, i = mb0 + j;
    if (j < n_a) clr_a[j] = 0;
    if (j < n_b) clr_b[j] = 0;
    if (j < n_c) clr_c[j] = 0;
    const int mbw = ctx->mbw;
    if (i >= mb1) return;
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
    { // one atomic per band that has work among the 64 consecutive macroblocks of this wave
        const int band = my / band_rows;
        const bool work = (bv | bh) != 0;
        unsigned long long rem = __ballot(work);
        while (rem) {
            const int l = __builtin_ctzll(rem);
            const int b = __builtin_amdgcn_readlane(band, l);
            if ((int)(threadIdx.x & 63) == l) atomicOr(&flags[b], 1u);
            rem &= ~__ballot(work && band == b);
        }
    }
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

// ------------------------------------------------------------------ the band kernel: one wave per macroblock row
// The four vertical (then
// the four horizontal) edges of a macroblock are almost independent: a normal-strength filter changes p1, p0, q0, q1 and
// reads p2 and q2; q2 of edge e is p1 of edge e + 1 (which edge e + 1 changes, but only after edge e has read it: everybody
// reads originals) and q1 of edge e is p2 of edge e + 1 -- the single true dependency.  q1' needs only the q side, p0 and the
// filter flag, none of which involve p2, so all four edges compute q1' at once, hand it to the next edge's lanes with one DPP
// move, and finish.  The strong filter (bS 4) exists only on the macroblock edge and also changes q2 = p1 of edge 1, which
// edge 1's flag reads: a wave that sees a bS 4 runs the macroblock edge as a pass of its own first.  So the chain of a step is
// one or two edge filters per direction instead of four, on 64 lanes = 16 lines x 4 edges of ONE macroblock, and a
// workgroup is a band of DB_ROWS rows with one wave each (luma and chroma in separate workgroups: they share nothing but
// the records).  The form this replaced -- 16 lanes per macroblock, a wave serving four rows, 16-row bands, strips + drain +
// progress counter between bands -- took 0.34 ms per 1080p P picture against 0.20 ms.
#define RFL2(v) ((unsigned)__builtin_amdgcn_readfirstlane((int)(v)))
#define DBR_TS 20 /* tile row: bytes 0..3 = the four samples left of the macroblock (chroma: two per plane), 4..19 = its 16 bytes */
struct dbr_luma { uint8_t t[20 * DBR_TS]; unsigned ring[4][16]; };   // rows -4..15
struct dbr_chroma { uint8_t t[10 * DBR_TS]; unsigned ring[4][8]; };  // rows -2..7
DEV int quad_prev(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x90, 0xF, 0xF, false); } // quad_perm:[0,0,1,2]: value of lane - 1 of the quad
DEV int quad_next(int v) { return __builtin_amdgcn_update_dpp(0, v, 0xF9, 0xF, 0xF, false); } // quad_perm:[1,2,3,3]: value of lane + 1 of the quad

// The four luma edges of one direction; lane = 4 * line + edge.  s[0..7] = p3 p2 p1 p0 q0 q1 q2 q3 of this lane's edge
// (originals); on return s[2..5] (p1 p0 q0 q1) are final, for edge 0 also s[1] (p2), and s[6] is NOT (it is the next edge's p1).
// bS: this lane's boundary strength; PE / PI: parameters of the macroblock edge / the inner edges.
template <bool ALL_INTRA>
DEV void edges4_luma(const edge_par &PE, const edge_par &PI, int e, int bS, bool any4, int *s) {
    if (ALL_INTRA || any4) { // the macroblock edge first (lanes of the other edges compute and discard)
        const int b0 = e == 0 ? bS : 0;
        edge_luma2<true>(PE, s[0], s[1], s[2], s[3], s[4], s[5], s[6], s[7], b0, true);
        const int f1 = quad_prev(s[5]), f2 = quad_prev(s[6]); // edge 0's q1', q2' are edge 1's p2, p1
        s[1] = e == 1 ? f1 : s[1]; s[2] = e == 1 ? f2 : s[2];
        bS = e == 0 ? 0 : bS;
    }
    const int alpha = (!(ALL_INTRA || any4) && e == 0) ? PE.alpha : PI.alpha, beta = (!(ALL_INTRA || any4) && e == 0) ? PE.beta : PI.beta;
    const unsigned tcw = (!(ALL_INTRA || any4) && e == 0) ? PE.tc0 : PI.tc0;
    const int p1 = s[2], p0 = s[3], q0 = s[4], q1 = s[5], q2 = s[6];
    const int mf = ((adiff(p0, q0) - alpha) & (adiff(p1, p0) - beta) & (adiff(q1, q0) - beta) & -bS) >> 31; // bS is 0..3 here
    const int maq = (adiff(q2, q0) - beta) >> 31;
    const int tc0 = (int)((tcw >> (8 * ((bS - 1) & 3))) & 0xFF); // bS 0 reads a don't-care byte
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

template <bool CHROMA, bool ALL_INTRA, int ROWS>
DEV void rows_body(const db_args &a, const unsigned *__restrict__ recs, const int band, const int nb, uint8_t *lds) {
    constexpr int rows_mb = CHROMA ? 8 : 16, strip = CHROMA ? 2 : 4, ring_n = CHROMA ? 8 : 16, TS = DBR_TS;
    constexpr int ROW_LDS = CHROMA ? (int)sizeof(dbr_chroma) : (int)sizeof(dbr_luma);
    constexpr int T_BYTES = (CHROMA ? 10 : 20) * TS;
    const frame_ctx_t *__restrict__ ctx = &a.ctx;
    const int mbw = ctx->mbw, mbh = ctx->mbh, stride = ctx->stride;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63; // the wave index as an SGPR: everything derived from it
    const int my = band * ROWS + wave;                                                              // (row, step, activity) is scalar control flow
    const bool row_ok = my < mbh, last_row = my == mbh - 1;
    const bool fed = row_ok && wave == 0 && band > 0;
    const bool feeds = row_ok && wave == ROWS - 1 && !last_row;
    // no edge of this band has work (deblock_prep_kernel's flags): its samples are final as they are, and the band below, which
    // reads the same flags, takes this band's bottom strip straight from the picture
    if (!ALL_INTRA && a.progress[2 * nb + band] == 0) return;
    const bool up_work = ALL_INTRA || (band > 0 && a.progress[2 * nb + band - 1] != 0);     // the band above publishes its strips
    const bool dn_work = ALL_INTRA || (band + 1 < nb && a.progress[2 * nb + band + 1] != 0); // ... and the band below reads ours
    // The band kernel may be launched in several pieces (bands [band0, band1)).  Strips between bands travel as 8-byte {samples, picture epoch} granules, one `sc1` store each, which the consumer polls
    // directly (MI355X_MICROARCH.md, hand-off R2 / handoff-1to1): no drain, no separate counter, one round trip.
    const unsigned epoch = ctx->epoch;
    uint2 *gran_up = a.gran + (CHROMA ? (size_t)nb * mbw * 16 : 0) + (size_t)(band > 0 ? band - 1 : 0) * mbw * ring_n;
    uint2 *gran_my = a.gran + (CHROMA ? (size_t)nb * mbw * 16 : 0) + (size_t)band * mbw * ring_n;
    uint8_t *__restrict__ plane = CHROMA ? ctx->rec_uv : ctx->rec_y;
    const size_t row0 = (size_t)my * rows_mb;
    uint8_t *tile = lds + wave * ROW_LDS;
    unsigned *ring = (unsigned *)(tile + T_BYTES);
    const unsigned *ring_up = (const unsigned *)(lds + (wave > 0 ? wave - 1 : 0) * ROW_LDS + T_BYTES);
    const int keep = last_row ? rows_mb : rows_mb - strip; // rows stored by this row itself; the strip below goes through the ring
    // lane roles.  Filters: line k = lane >> 2 (luma) with edge e = lane & 3.  Data movement: word (lane & 3) of line (lane >> 2).
    const int k = lane >> 2, e = lane & 3;
    const int mrow = lane >> 2, mword = lane & 3;       // one word per lane covers 16 lines x 16 bytes
    const bool mlane = CHROMA ? lane < 32 : true;       // chroma: 8 lines
    const bool slane = lane < strip * 4;                // one word per lane covers a strip
    unsigned own = 0, stripv = 0, recv = 0;
    // the 64-byte record of a macroblock is wave-uniform (one macroblock per wave): loaded one word per lane a step ahead, moved to SGPRs by v_readlane when it has landed
    const int rec_row = __builtin_amdgcn_readfirstlane(row_ok ? my : 0) * mbw;
    uint4 rc0 = make_uint4(0, 0, 0, 0), rc1 = rc0, rc2 = rc0;
    // per-lane addresses of this row's lines, less the macroblock's x offset
    const uint8_t *ld_base = plane + (row0 + mrow) * stride + 4 * mword;
    uint8_t *sc_base = plane + (row0 - (my > 0 ? strip : 0) + mrow) * stride + 4 * mword;
    uint2 gnext = make_uint2(0, 0);
    const int nsteps = mbw + ROWS + 2;
#ifdef DBR_PROF /* debug builds only: per-phase cycle sums of wave 1 of band 1 (every tick drains lgkmcnt: perturbs) */
    unsigned long long pc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tm0 = 0, tm1;
#define DBR_TICK(i) do { tm1 = __builtin_readcyclecounter(); pc[i] += tm1 - tm0; tm0 = tm1; } while (0)
#else
#define DBR_TICK(i) do { } while (0)
#endif
    for (int t = 0; t < nsteps; t++) {
        const int x = t - 1 - wave, xn = x + 1;
        const bool act = row_ok && x >= 0 && x < mbw;
        const bool pf = row_ok && xn >= 0 && xn < mbw;
        const bool pub = feeds && x >= 1 && x <= mbw;         // strip of macroblock x-1 becomes final in this step's vertical phase
        const int x0b = x * 16;
#ifdef DBR_PROF
        tm0 = __builtin_readcyclecounter();
#endif
        // ---- A. prefetch macroblock x+1 (its samples, its record, the strip of the band above)
        if (pf) {
            if (fed && slane) {
                if (up_work) gnext = ld64_sc1(gran_up + (size_t)xn * ring_n + lane);
                else { gnext.x = ldg32(plane + (row0 - strip + mrow) * stride + xn * 16 + 4 * mword); gnext.y = epoch; }
            }
#ifndef DBX_NOLOAD
            if (mlane) own = ldg32(ld_base + xn * 16);
#endif
            if (lane < 16) recv = ldg32(recs + (size_t)(rec_row + xn) * (DBREC_BYTES / 4) + lane);
        }
        DBR_TICK(0);
        // the record is wave-uniform now (one macroblock per wave): boundary strengths in SGPRs
        unsigned bvl = 0, bvh = 0, bhl = 0, bhh = 0;
        if (act) { bvl = rc0.x; bvh = rc0.y; bhl = rc0.z; bhh = rc0.w; }
        const edge_par PL = par_of(rc1.x, rc1.y), PT = par_of(rc1.z, rc1.w), PI = par_of(rc2.x, rc2.y);
        DBR_TICK(1);
        // ---- B. vertical edges
#ifdef DBX_NOFILT
        if (0) {
#else
        if (ALL_INTRA ? act : (bvl | bvh) != 0) {
#endif
            if (!CHROMA) {
                unsigned *tw = (unsigned *)&tile[(k + 4) * TS + 4 * e];
                const unsigned w0 = tw[0], w1 = tw[1];
                int s[8];
#pragma unroll
                for (int i = 0; i < 4; i++) { s[i] = byte_of(w0, i); s[4 + i] = byte_of(w1, i); }
                const unsigned bw = e < 2 ? bvl : bvh;
                const int bS = (int)((bw >> (16 * (e & 1) + 4 * (k >> 2))) & 15);
                edges4_luma<ALL_INTRA>(PL, PI, e, bS, (bvl & 0x4444u) != 0, s);
                // word of columns 4e .. 4e+3: own q0 q1, then the next edge's p1 p0 (edge 3: own q2 q3, which no edge of this macroblock changes)
                const int nx = quad_next(s[2] | (s[3] << 8));
                const unsigned hi = e == 3 ? (unsigned)(s[6] | (s[7] << 8)) : (unsigned)nx;
                tw[1] = (unsigned)s[4] | ((unsigned)s[5] << 8) | (hi << 16);
                if (e == 0) {
                    const unsigned l0 = pack4(s[0], s[1], s[2], s[3]);
                    tw[0] = l0;
                    if (k >= 12 && x > 0 && !last_row) ring[((x - 1) & 3) * 16 + (k - 12) * 4 + 3] = l0; // columns 12..15 of the previous macroblock's strip
                }
            } else if (lane < 32) {
                const int kk = lane >> 2, c = (lane >> 1) & 1, ee = lane & 1; // line, plane, edge (luma edges 0 and 2)
                uint8_t *b = &tile[(kk + 2) * TS + c + 8 * ee];               // samples of a plane sit 2 bytes apart; p1 p0 q0 q1
                int p1 = b[0], p0 = b[2], q0 = b[4], q1 = b[6];
                const int bS = (int)(((ee ? bvh : bvl) >> (4 * (kk >> 1))) & 15);
                const edge_par P = ee ? PI : PL;
                edge_chroma2(P, p1, p0, q0, q1, bS);
                b[2] = (uint8_t)p0; b[4] = (uint8_t)q0;
            }
            if (CHROMA) {
                WAVE_SYNC();
                if (lane >= 6 && lane < 8 && x > 0 && !last_row) ring[((x - 1) & 3) * 8 + (lane - 6) * 4 + 3] = *(const unsigned *)&tile[(lane + 2) * TS];
            }
        }
        // ---- the strip of macroblock x-1 is final now: hand it to the band below
        if (pub) {
            WAVE_SYNC();
            if (slane) {
                const unsigned w = ring[((x - 1) & 3) * ring_n + lane];
                if (dn_work) st64_sc1(gran_my + (size_t)(x - 1) * ring_n + lane, make_uint2(w, epoch));
                else stg32(plane + (row0 + rows_mb - strip + mrow) * stride + (x - 1) * 16 + 4 * mword, w); // nobody below will store it
            }
        }
        DBR_TICK(2);
        // ---- C. the one barrier of the step: every vertical edge of this step precedes every horizontal edge
        BAND_BARRIER();
        DBR_TICK(3);
        unsigned fin = 0, fsb = 0, fsc = 0;
        if (act) {
            // ---- D. horizontal edges
            if (my > 0 && slane) *(unsigned *)&tile[mrow * TS + 4 + 4 * mword] = fed ? stripv : ring_up[(x & 3) * ring_n + lane];
            WAVE_SYNC();
#ifdef DBX_NOFILT
            if (0) {
#else
            if (ALL_INTRA || (bhl | bhh) != 0) {
#endif
                if (!CHROMA) {
                    uint8_t *col = &tile[(4 * e) * TS + 4 + k]; // column k, rows 4e-4 .. 4e+3
                    int s[8];
#pragma unroll
                    for (int i = 0; i < 8; i++) s[i] = col[i * TS];
                    const unsigned bw = e < 2 ? bhl : bhh;
                    const int bS = (int)((bw >> (16 * (e & 1) + 4 * (k >> 2))) & 15);
                    const bool any4 = (bhl & 0x4444u) != 0;
                    edges4_luma<ALL_INTRA>(PT, PI, e, bS, any4, s);
                    col[2 * TS] = (uint8_t)s[2]; col[3 * TS] = (uint8_t)s[3]; col[4 * TS] = (uint8_t)s[4]; col[5 * TS] = (uint8_t)s[5];
                    if ((ALL_INTRA || any4) && e == 0) col[1 * TS] = (uint8_t)s[1];
                } else if (lane < 32) {
                    const int kb = lane >> 1, ee = lane & 1; // byte column, edge
                    uint8_t *col = &tile[(4 * ee) * TS + 4 + kb];
                    int p1 = col[0], p0 = col[TS], q0 = col[2 * TS], q1 = col[3 * TS];
                    const int bS = (int)(((ee ? bhh : bhl) >> (4 * (kb >> 2))) & 15);
                    const edge_par P = ee ? PI : PT;
                    edge_chroma2(P, p1, p0, q0, q1, bS);
                    col[TS] = (uint8_t)p0; col[2 * TS] = (uint8_t)q0;
                }
            }
            WAVE_SYNC();
            DBR_TICK(4);
            // ---- E. bottom strip -> ring (read by the row below after the next barrier); final samples into registers
            if (!last_row && slane) ring[(x & 3) * ring_n + lane] = *(const unsigned *)&tile[(rows_mb + mrow) * TS + 4 + 4 * mword];
            if (mlane) {
                const unsigned *rp = (const unsigned *)&tile[(mrow + strip) * TS];
                fin = rp[mword]; // line mrow, byte columns 4*mword-4 .. 4*mword-1 (the left strip is final now)
                fsb = rp[4];     // ... and columns 12..15, which only the last macroblock of a row stores itself
            }
            if (my > 0 && slane) fsc = *(const unsigned *)&tile[mrow * TS + 4 + 4 * mword]; // the strip of the row above is final after this top edge
            WAVE_SYNC();
            if (mlane && mword == 0) *(unsigned *)&tile[(mrow + strip) * TS] = fsb; // right strip becomes the next macroblock's left strip
        }
        DBR_TICK(5);
        // ---- land the prefetch (issued a whole step ago) before this step's stores queue up behind it
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("" ::"v"(own), "v"(recv), "v"(gnext.x), "v"(gnext.y));
        DBR_TICK(6);
        if (pf) {
            if (fed) { // every granule of the strip must carry this picture's epoch; in steady state the first load already does
                int spins = 0;
                while (__ballot(slane && gnext.y != epoch)) {
                    __builtin_amdgcn_s_sleep(1);
                    if (slane) gnext = ld64_sc1(gran_up + (size_t)xn * ring_n + lane);
                    if (++spins > DB_SPIN_MAX || ((spins & 1023) == 0 && ld_sc1(a.err))) { st_sc1(a.err, 1u); break; } // bounded; once tripped, nobody waits again
                }
                stripv = gnext.x;
            }
            WAVE_SYNC();
            if (mlane) *(unsigned *)&tile[(mrow + strip) * TS + 4 + 4 * mword] = own;
        }
        DBR_TICK(7);
        // ---- F. stores (nobody inside this launch reads them back)
#ifndef DBX_NOSTORE
        if (act) {
            if (mlane && mrow < keep && (x > 0 || mword > 0)) stg32((uint8_t *)ld_base + x0b - 4, fin);
            if (x == mbw - 1 && mlane && mrow < keep && mword == 3) stg32((uint8_t *)ld_base + x0b, fsb); // no right neighbour will patch columns 12..15
            if (my > 0 && slane) stg32(sc_base + x0b, fsc);
        }
#endif
        DBR_TICK(8);
        {
#define RL(i) ((unsigned)__builtin_amdgcn_readlane((int)recv, (i)))
            constexpr int o = CHROMA ? 10 : 4; // luma: left, top, inner at words 4..9; chroma: at words 10..15
            rc0 = make_uint4(RL(0), RL(1), RL(2), RL(3));
            rc1 = make_uint4(RL(o), RL(o + 1), RL(o + 2), RL(o + 3));
            rc2 = make_uint4(RL(o + 4), RL(o + 5), 0, 0);
        }
    }
#ifdef DBR_PROF
    if (lane == 0 && wave == 1 && band == 1) {
        unsigned *o = (unsigned *)(ctx->dbrec) + (CHROMA ? 16 : 0); // debug build only: overwrites the first records after use
        for (int i = 0; i < 9; i++) o[i] = (unsigned)pc[i];
        o[9] = (unsigned)nsteps;
    }
#endif
}

template <int ROWS, bool ALL_INTRA>
__global__ __launch_bounds__(64 * ROWS) void deblock_rows_kernel(db_args a, const unsigned *__restrict__ recs) { // recs = a.ctx.dbrec, as a noalias argument: scalar loads
    __shared__ __attribute__((aligned(16))) uint8_t lds[ROWS * sizeof(dbr_luma)];
    const int nl = gridDim.x >> 1;
    if ((int)blockIdx.x < nl) rows_body<false, ALL_INTRA, ROWS>(a, recs, a.band0 + blockIdx.x, a.nb_total, lds);
    else rows_body<true, ALL_INTRA, ROWS>(a, recs, a.band0 + blockIdx.x - nl, a.nb_total, lds);
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
#define DB_ROWS 4 /* rows per band = waves per workgroup, one per SIMD; 8 is as fast on P pictures and 13 % slower on I pictures */
int k_deblock_bands16(int mbh) { return (mbh + DB_ROWS - 1) / DB_ROWS; }
size_t k_deblock_gran_bytes(int mbw, int mbh) { return (size_t)k_deblock_bands16(mbh) * mbw * 24 * sizeof(uint2); } // per band boundary and macroblock: 16 luma + 8 chroma granules
// `d_progress` holds 2 * bands counters (luma, chroma) followed by the sticky error word at d_err.  The prep kernel clears
// the counters; the band kernel may be launched in several pieces (bands [band0, band1)): a band only ever waits for the
// band above it, so pieces may run concurrently on different streams as long as the upper piece is submitted first.
void k_launch_deblock_prep(const frame_ctx_t *h_ctx, int mbw, int row0, int row1, unsigned *clr_a, int n_a, unsigned *clr_b, int n_b, unsigned *clr_c, int n_c,
                           unsigned *flags, hipStream_t s) {
    int m = (row1 - row0) * mbw;
    if (n_a > m) m = n_a;
    if (n_b > m) m = n_b;
    if (n_c > m) m = n_c;
    if (m > 0) hipLaunchKernelGGL(deblock_prep_kernel, dim3((m + 255) / 256), dim3(256), 0, s, *h_ctx, clr_a, n_a, clr_b, n_b, clr_c, n_c, flags, row0 * mbw, row1 * mbw, DB_ROWS);
}
void k_launch_deblock_bands(const frame_ctx_t *h_ctx, int mbh, int band0, int band1, unsigned *d_progress, unsigned *d_err, uint2 *d_gran, hipStream_t s) {
    db_args a;
    a.ctx = *h_ctx; a.progress = d_progress; a.err = d_err; a.band0 = band0; a.nb_total = k_deblock_bands16(mbh); a.gran = d_gran;
    if (band1 <= band0) return;
    const dim3 g(2 * (band1 - band0));
    if (h_ctx->all_intra) hipLaunchKernelGGL((deblock_rows_kernel<DB_ROWS, true>), g, dim3(64 * DB_ROWS), 0, s, a, (const unsigned *)h_ctx->dbrec); // IDR pictures: every edge has work
    else hipLaunchKernelGGL((deblock_rows_kernel<DB_ROWS, false>), g, dim3(64 * DB_ROWS), 0, s, a, (const unsigned *)h_ctx->dbrec);
}
