/*
 * h264_host.c -- host side of the encoder: Annex-B writer, parameter sets, CAVLC slice
 * coding.  north_star keeps entropy coding on the host thread; the device hands over one
 * 16-byte record and 408 int16 levels per macroblock (mi355enc_dev.h).
 *
 * Produces what the reference's graph expects from its encoder element: `video/x-h264,
 * stream-format=byte-stream, alignment=au`, SPS+PPS in band before every IDR so that
 * `h264parse config-interval=-1` (/root/reference/pipeline/generic/x264_superfast_camlink:6)
 * passes it through.  Clause numbers refer to ITU-T H.264.
 */
#include "h264_host.h"

#include <emmintrin.h>
#include <pthread.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#include "h264_vlc_tables.h"

/* Table 9-4 (inter column), coded_block_pattern -> codeNum, ChromaArrayType 1 */
static const uint8_t cbp_inter_code[48] = {0,  2,  3,  7,  4,  8,  17, 13, 5,  18, 9,  14, 10, 15, 16, 11,
                                           1,  32, 33, 36, 34, 37, 44, 40, 35, 45, 38, 41, 39, 42, 43, 19,
                                           6,  24, 25, 20, 26, 21, 46, 28, 27, 47, 22, 29, 23, 30, 31, 12};
static const uint8_t blk_to_raster[16] = {0, 1, 4, 5, 2, 3, 6, 7, 8, 9, 12, 13, 10, 11, 14, 15}; /* self-inverse */
/* Table 9-4 (intra column) */
static const uint8_t cbp_intra_code[48] = {3,  29, 30, 17, 31, 18, 37, 8,  32, 38, 19, 9,  20, 10, 11, 2,
                                           16, 33, 34, 21, 35, 22, 39, 4,  36, 40, 23, 5,  24, 6,  7,  1,
                                           41, 42, 43, 25, 44, 26, 46, 12, 45, 47, 27, 13, 28, 14, 15, 0};

/* ------------------------------------------------------------------ bit sink */
typedef struct {
    uint8_t *p, *end;
    uint64_t acc; /* bits are collected at the low end */
    int n;        /* valid bits in acc (< 32 after every put) */
    int overflow;
} bits_t;

static inline void bits_init(bits_t *b, uint8_t *buf, size_t cap) { b->p = buf; b->end = buf + cap; b->acc = 0; b->n = 0; b->overflow = 0; }
static inline void bits_put(bits_t *b, int len, uint32_t v) { /* len 0..32, v < 2^len */
    b->acc = (b->acc << len) | v;
    b->n += len;
    if (b->n >= 32) {
        b->n -= 32;
        uint32_t w = (uint32_t)(b->acc >> b->n);
        if (b->end - b->p >= 4) { b->p[0] = (uint8_t)(w >> 24); b->p[1] = (uint8_t)(w >> 16); b->p[2] = (uint8_t)(w >> 8); b->p[3] = (uint8_t)w; b->p += 4; }
        else b->overflow = 1;
    }
}
static inline void bits_ue(bits_t *b, uint32_t v) {
    uint32_t k = v + 1;
    int len = 31 - __builtin_clz(k);
    bits_put(b, len, 0);
    bits_put(b, len + 1, k);
}
static inline void bits_se(bits_t *b, int v) { bits_ue(b, v > 0 ? (uint32_t)(2 * v - 1) : (uint32_t)(-2 * v)); }
static size_t bits_finish(bits_t *b, uint8_t *base) { /* rbsp_trailing_bits + byte flush */
    bits_put(b, 1, 1);
    if (b->n & 7) bits_put(b, 8 - (b->n & 7), 0);
    while (b->n > 0) {
        b->n -= 8;
        if (b->p < b->end) *b->p++ = (uint8_t)(b->acc >> b->n);
        else b->overflow = 1;
    }
    return (size_t)(b->p - base);
}

/* start code + header + payload with emulation prevention (7.4.1.1); 0 when out of room */
static size_t emit_nal(uint8_t *out, size_t cap, int ref_idc, int type, const uint8_t *rbsp, size_t n) {
    if (cap < 5) return 0;
    uint8_t *o = out, *oe = out + cap;
    *o++ = 0; *o++ = 0; *o++ = 0; *o++ = 1;
    *o++ = (uint8_t)((ref_idc << 5) | type);
    int zeros = 0;
    for (size_t i = 0; i < n; i++) {
        uint8_t v = rbsp[i];
        if (zeros >= 2 && v <= 3) {
            if (o >= oe) return 0;
            *o++ = 3;
            zeros = 0;
        }
        if (o >= oe) return 0;
        *o++ = v;
        zeros = v ? 0 : zeros + 1;
    }
    return (size_t)(o - out);
}

/* ------------------------------------------------------------------ parameter sets */
static int pick_level(int mbw, int mbh, int fps_num, int fps_den) { /* Table A-1 */
    static const int t[][3] = {{10, 1485, 99},      {11, 3000, 396},     {12, 6000, 396},    {13, 11880, 396},
                               {20, 11880, 396},    {21, 19800, 792},    {22, 20250, 1620},  {30, 40500, 1620},
                               {31, 108000, 3600},  {32, 216000, 5120},  {40, 245760, 8192}, {42, 522240, 8704},
                               {50, 589824, 22080}, {51, 983040, 36864}, {52, 2073600, 36864}};
    long long fs = (long long)mbw * mbh, rate = (fs * fps_num + fps_den - 1) / fps_den;
    for (unsigned i = 0; i < sizeof t / sizeof t[0]; i++)
        if (fs <= t[i][2] && rate <= t[i][1] && mbw * mbw <= 8 * t[i][2] && mbh * mbh <= 8 * t[i][2]) return t[i][0];
    return 52;
}

size_t h264_write_headers(uint8_t *out, size_t cap, int width, int height, int fps_num, int fps_den, int t8) {
    uint8_t rb[160];
    bits_t b;
    const int mbw = (width + 15) / 16, mbh = (height + 15) / 16;
    /* 7.3.2.1.1 seq_parameter_set_data: Constrained Baseline */
    bits_init(&b, rb, sizeof rb);
    bits_put(&b, 8, t8 ? 100u : 66u);  /* High when the 8x8 transform is on, else Constrained Baseline */
    bits_put(&b, 8, t8 ? 0x00u : 0xC0u);
    bits_put(&b, 8, (uint32_t)pick_level(mbw, mbh, fps_num, fps_den));
    bits_ue(&b, 0);
    if (t8) { bits_ue(&b, 1); bits_ue(&b, 0); bits_ue(&b, 0); bits_put(&b, 2, 0); } /* 4:2:0, 8 bit, no bypass, flat scaling */
    bits_ue(&b, 4); /* log2_max_frame_num_minus4 */
    bits_ue(&b, 2); /* pic_order_cnt_type 2: output order = decoding order */
    bits_ue(&b, 1); /* max_num_ref_frames */
    bits_put(&b, 1, 0);
    bits_ue(&b, (uint32_t)(mbw - 1));
    bits_ue(&b, (uint32_t)(mbh - 1));
    bits_put(&b, 1, 1); /* frame_mbs_only_flag */
    bits_put(&b, 1, 1); /* direct_8x8_inference_flag */
    const int cr = (mbw * 16 - width) / 2, cb = (mbh * 16 - height) / 2;
    if (cr || cb) { bits_put(&b, 1, 1); bits_ue(&b, 0); bits_ue(&b, (uint32_t)cr); bits_ue(&b, 0); bits_ue(&b, (uint32_t)cb); }
    else bits_put(&b, 1, 0);
    bits_put(&b, 1, 1);  /* vui_parameters_present_flag */
    bits_put(&b, 4, 0);  /* aspect_ratio, overscan, video_signal_type, chroma_loc: absent */
    bits_put(&b, 1, 1);  /* timing_info_present_flag */
    bits_put(&b, 16, (uint32_t)fps_den >> 16); bits_put(&b, 16, (uint32_t)fps_den & 0xFFFF);
    bits_put(&b, 16, (uint32_t)(2 * fps_num) >> 16); bits_put(&b, 16, (uint32_t)(2 * fps_num) & 0xFFFF);
    bits_put(&b, 1, 1);  /* fixed_frame_rate_flag */
    bits_put(&b, 3, 0);  /* nal_hrd, vcl_hrd, pic_struct: absent */
    bits_put(&b, 1, 1);  /* bitstream_restriction_flag */
    bits_put(&b, 1, 1);  /* motion_vectors_over_pic_boundaries_flag */
    bits_ue(&b, 0); bits_ue(&b, 0); bits_ue(&b, 10); bits_ue(&b, 10);
    bits_ue(&b, 0); /* max_num_reorder_frames */
    bits_ue(&b, 1); /* max_dec_frame_buffering */
    size_t n = bits_finish(&b, rb);
    size_t a = emit_nal(out, cap, 3, 7, rb, n);
    if (!a || b.overflow) return 0;
    /* 7.3.2.2 pic_parameter_set_rbsp */
    bits_init(&b, rb, sizeof rb);
    bits_ue(&b, 0); bits_ue(&b, 0);
    bits_put(&b, 1, 0); /* CAVLC */
    bits_put(&b, 1, 0);
    bits_ue(&b, 0); bits_ue(&b, 0); bits_ue(&b, 0);
    bits_put(&b, 3, 0); /* weighted_pred_flag, weighted_bipred_idc */
    bits_se(&b, 0); bits_se(&b, 0); bits_se(&b, 0);
    bits_put(&b, 1, 1); /* deblocking_filter_control_present_flag */
    bits_put(&b, 2, 0); /* constrained_intra_pred_flag, redundant_pic_cnt_present_flag */
    if (t8) { bits_put(&b, 2, 2); bits_se(&b, 0); } /* transform_8x8_mode_flag = 1, pic_scaling_matrix_present_flag = 0, second_chroma_qp_index_offset */
    n = bits_finish(&b, rb);
    size_t c = emit_nal(out + a, cap - a, 3, 8, rb, n);
    if (!c || b.overflow) return 0;
    return a + c;
}

/* ------------------------------------------------------------------ residual block (9.2) */
/* coef: 16 int16 in scan order, 16-byte aligned group; `skip_first` drops coef[0]
 * (Intra16x16 AC / chroma AC: maxNumCoeff 15).  Returns TotalCoeff. */
static inline int put_block16(bits_t *b, const int16_t *coef, int skip_first, int nC) {
    __m128i z = _mm_setzero_si128();
    __m128i lo = _mm_loadu_si128((const __m128i *)coef), hi = _mm_loadu_si128((const __m128i *)(coef + 8));
    unsigned zm = (unsigned)_mm_movemask_epi8(_mm_packs_epi16(_mm_cmpeq_epi16(lo, z), _mm_cmpeq_epi16(hi, z)));
    unsigned nzm = ~zm & 0xFFFFu; /* bit k: coef[k] != 0 */
    if (skip_first) nzm >>= 1, coef++;
    const int maxnum = skip_first ? 15 : 16;
    const int total = __builtin_popcount(nzm);
    const int cls = nC < 2 ? 0 : nC < 4 ? 1 : nC < 8 ? 2 : 3;
    if (!total) { bits_put(b, vlc_coeff_token[cls][0][0].len, vlc_coeff_token[cls][0][0].bits); return 0; }
    int idx[16], n = 0;
    for (unsigned m = nzm; m; m &= m - 1) idx[n++] = __builtin_ctz(m);
    int t1 = 0;
    for (int i = n - 1; i >= 0 && t1 < 3; i--) {
        int v = coef[idx[i]];
        if (v == 1 || v == -1) t1++;
        else break;
    }
    bits_put(b, vlc_coeff_token[cls][total][t1].len, vlc_coeff_token[cls][total][t1].bits);
    uint32_t signs = 0;
    for (int i = 0; i < t1; i++) signs = (signs << 1) | (coef[idx[n - 1 - i]] < 0);
    bits_put(b, t1, signs);
    int sl = (total > 10 && t1 < 3) ? 1 : 0;
    for (int i = n - 1 - t1; i >= 0; i--) {
        int lv = coef[idx[i]], av = lv < 0 ? -lv : lv;
        int code = 2 * av - 2 + (lv < 0);
        if (i == n - 1 - t1 && t1 < 3) code -= 2;
        if (sl == 0) {
            if (code < 14) bits_put(b, code + 1, 1);
            else if (code < 30) { bits_put(b, 15, 1); bits_put(b, 4, (uint32_t)(code - 14)); }
            else { bits_put(b, 16, 1); bits_put(b, 12, (uint32_t)(code - 30)); }
            sl = 1;
        } else if (code < (15 << sl)) {
            bits_put(b, (code >> sl) + 1, 1);
            bits_put(b, sl, (uint32_t)code & ((1u << sl) - 1));
        } else { bits_put(b, 16, 1); bits_put(b, 12, (uint32_t)(code - (15 << sl))); }
        if (av > (3 << (sl - 1)) && sl < 6) sl++;
    }
    int zl = idx[n - 1] + 1 - total;
    if (total < maxnum) bits_put(b, vlc_total_zeros[total - 1][zl].len, vlc_total_zeros[total - 1][zl].bits);
    for (int i = n - 1; i > 0 && zl > 0; i--) {
        int run = idx[i] - idx[i - 1] - 1;
        const vlc_t *v = &vlc_run_before[(zl > 7 ? 7 : zl) - 1][run];
        bits_put(b, v->len, v->bits);
        zl -= run;
    }
    return total;
}
static inline void put_chroma_dc(bits_t *b, const int16_t *coef) { /* 4 coefficients, nC = -1 */
    int idx[4], n = 0;
    for (int i = 0; i < 4; i++) if (coef[i]) idx[n++] = i;
    int t1 = 0;
    for (int i = n - 1; i >= 0 && t1 < 3; i--) {
        if (coef[idx[i]] == 1 || coef[idx[i]] == -1) t1++;
        else break;
    }
    bits_put(b, vlc_coeff_token_cdc[n][t1].len, vlc_coeff_token_cdc[n][t1].bits);
    if (!n) return;
    for (int i = 0; i < t1; i++) bits_put(b, 1, coef[idx[n - 1 - i]] < 0);
    int sl = 0;
    for (int i = n - 1 - t1; i >= 0; i--) {
        int lv = coef[idx[i]], av = lv < 0 ? -lv : lv;
        int code = 2 * av - 2 + (lv < 0);
        if (i == n - 1 - t1 && t1 < 3) code -= 2;
        if (sl == 0) {
            if (code < 14) bits_put(b, code + 1, 1);
            else if (code < 30) { bits_put(b, 15, 1); bits_put(b, 4, (uint32_t)(code - 14)); }
            else { bits_put(b, 16, 1); bits_put(b, 12, (uint32_t)(code - 30)); }
            sl = 1;
        } else if (code < (15 << sl)) {
            bits_put(b, (code >> sl) + 1, 1);
            bits_put(b, sl, (uint32_t)code & ((1u << sl) - 1));
        } else { bits_put(b, 16, 1); bits_put(b, 12, (uint32_t)(code - (15 << sl))); }
        if (av > (3 << (sl - 1)) && sl < 6) sl++;
    }
    int zl = idx[n - 1] + 1 - n;
    if (n < 4) bits_put(b, vlc_total_zeros_cdc[n - 1][zl].len, vlc_total_zeros_cdc[n - 1][zl].bits);
    for (int i = n - 1; i > 0 && zl > 0; i--) {
        int run = idx[i] - idx[i - 1] - 1;
        const vlc_t *v = &vlc_run_before[zl - 1][run];
        bits_put(b, v->len, v->bits);
        zl -= run;
    }
}

/* Test hook (include/mi355enc.h, mi355enc_host_cavlc_block): one residual block through the coder above. */
int h264_cavlc_block_bits(const int16_t *coef, int maxnum, int nC, uint8_t *out, size_t cap) {
    int16_t tmp[16] __attribute__((aligned(16))) = {0};
    bits_t b;
    if (cap < 64 || (maxnum != 16 && maxnum != 15 && maxnum != 4)) return -1;
    memset(out, 0, cap);
    bits_init(&b, out, cap);
    if (maxnum == 4) { memcpy(tmp, coef, 8); put_chroma_dc(&b, tmp); }
    else if (maxnum == 15) { memcpy(tmp + 1, coef, 30); put_block16(&b, tmp, 1, nC); }
    else { memcpy(tmp, coef, 32); put_block16(&b, tmp, 0, nC); }
    int n = (int)(8 * (b.p - out)) + b.n;
    if (b.n & 7) bits_put(&b, 8 - (b.n & 7), 0);
    while (b.n > 0) { b.n -= 8; *b.p++ = (uint8_t)(b.acc >> b.n); }
    return b.overflow ? -1 : n;
}

/* ------------------------------------------------------------------ motion vector prediction */
static inline int med3(int a, int b, int c) {
    int lo = a < b ? a : b, hi = a < b ? b : a;
    return c < lo ? lo : (c > hi ? hi : c);
}
/* Partitions (mb_type 1; i16_mode = shape 0 16x16, 1 16x8, 2 8x16, 3 8x8): geometry (x0, y0, w, h) and count */
static const int8_t part_geo[4][4][4] = {{{0, 0, 16, 16}}, {{0, 0, 16, 8}, {0, 8, 16, 8}}, {{0, 0, 8, 16}, {8, 0, 8, 16}}, {{0, 0, 8, 8}, {8, 0, 8, 8}, {0, 8, 8, 8}, {8, 8, 8, 8}}};
static const int8_t part_n[4] = {1, 2, 2, 4};
static inline int mb_shape(const mb_info_t *m) { return m->mb_type == 1 ? (m->i16_mode & 3) : 0; }
/* the quadrant vectors of an inter macroblock from its record and the block in the luma-DC slot of its levels (vectors of partitions 1 .. 3) */
static inline void set_qmv(int16_t *q, const mb_info_t *m, const int16_t *ldc) {
    const int sh = mb_shape(m);
    for (int k = 0; k < 4; k++) {
        const int idx = sh == 0 ? 0 : sh == 1 ? (k >> 1) : sh == 2 ? (k & 1) : k;
        q[2 * k] = idx ? ldc[2 * (idx - 1)] : m->mvx; q[2 * k + 1] = idx ? ldc[2 * (idx - 1) + 1] : m->mvy;
    }
}
/* 6.4.11.7 / 8.4.1.3.2: the 8x8 block covering luma sample (X, Y) as a neighbour of a partition of macroblock (mx, my), whose own quadrants in `done`
 * carry cur[]; a macroblock is available when it comes earlier in raster order and belongs to the same slice (top: the row above does -- slices are whole rows) */
static inline void nb_blk(const int16_t *qmv, const mb_info_t *mbi, int mbw, int mx, int my, int top, unsigned done, const int16_t *cur, int X, int Y, int *avail, int *ref, int *vx, int *vy) {
    *avail = 0; *ref = -1; *vx = *vy = 0;
    if (X < 0 || Y < 0 || X >= mbw * 16) return;
    const int nx = X >> 4, ny = Y >> 4, q = ((Y & 15) >> 3) * 2 + ((X & 15) >> 3);
    if (nx == mx && ny == my) { if ((done >> q) & 1) { *avail = 1; *ref = 0; *vx = cur[2 * q]; *vy = cur[2 * q + 1]; } return; }
    if (!(ny < my || (ny == my && nx < mx))) return;
    if (ny < my && !top) return;
    *avail = 1;
    if (mbi[(size_t)ny * mbw + nx].mb_type == 1) { const int16_t *v = qmv + ((size_t)ny * mbw + nx) * 8 + 2 * q; *ref = 0; *vx = v[0]; *vy = v[1]; }
}
/* 8.4.1.3 for partition idx of shape `shape` at (x0, y0) of width wd, refIdx 0 everywhere; skip: 8.4.1.1's inference for P_Skip */
static inline void predict_part(const int16_t *qmv, const mb_info_t *mbi, int mbw, int mx, int my, int top, unsigned done, const int16_t *cur, int shape, int idx, int x0, int y0, int wd, int skip,
                                int *px, int *py) {
    const int X = mx * 16 + x0, Y = my * 16 + y0;
    int aA, rA, ax, ay, aB, rB, bx, by, aC, rC, cx, cy;
    nb_blk(qmv, mbi, mbw, mx, my, top, done, cur, X - 1, Y, &aA, &rA, &ax, &ay);
    nb_blk(qmv, mbi, mbw, mx, my, top, done, cur, X, Y - 1, &aB, &rB, &bx, &by);
    nb_blk(qmv, mbi, mbw, mx, my, top, done, cur, X + wd, Y - 1, &aC, &rC, &cx, &cy);
    if (!aC) nb_blk(qmv, mbi, mbw, mx, my, top, done, cur, X - 1, Y - 1, &aC, &rC, &cx, &cy);
    *px = 0; *py = 0;
    if (skip && (!aA || !aB || (rA == 0 && !ax && !ay) || (rB == 0 && !bx && !by))) return;
    if (shape == 1 && idx == 0 && rB == 0) { *px = bx; *py = by; return; }
    if (shape == 1 && idx == 1 && rA == 0) { *px = ax; *py = ay; return; }
    if (shape == 2 && idx == 0 && rA == 0) { *px = ax; *py = ay; return; }
    if (shape == 2 && idx == 1 && rC == 0) { *px = cx; *py = cy; return; }
    if (!aB && !aC && aA) { rB = rC = rA; bx = cx = ax; by = cy = ay; }
    const int hits = (rA == 0) + (rB == 0) + (rC == 0);
    if (hits == 1) { *px = rA == 0 ? ax : rB == 0 ? bx : cx; *py = rA == 0 ? ay : rB == 0 ? by : cy; return; }
    *px = med3(ax, bx, cx);
    *py = med3(ay, by, cy);
}

/* ------------------------------------------------------------------ slice */
struct h264_writer {
    int mbw, mbh, t8;
    int slice_rows; /* I pictures: a new slice (own NAL unit; the row above its first row not available, 6.4.8) every so many macroblock rows; 0: one slice */
    int pslice_rows; /* ... and P pictures (r04): there the cut also breaks the 8.4.1.3 vector predictors and the P_Skip inference */
    int dbf_idc;     /* disable_deblocking_filter_idc of every slice header: 0 (the filter runs across slice boundaries) or 2 (it stops at them) */
    uint8_t *rbsp; size_t rbsp_cap;
    uint8_t *tc_l; /* TotalCoeff per luma 4x4 in raster order, 16 per macroblock */
    uint8_t *tc_c; /* per chroma AC block: Cb 0-3, Cr 4-7 */
    uint8_t *i4m;  /* Intra4x4PredMode per luma4x4BlkIdx, 16 per macroblock (valid where mb_type == 2) */
    int16_t *qmv;  /* the vector of each 8x8 quadrant of the inter macroblocks coded (or read, fill_ctx_row) so far, 8 per macroblock: what the predictors of
                      partitioned macroblocks (mb_type 1 with a shape in i16_mode: 1 16x8, 2 8x16, 3 8x8) and of their neighbours are derived from */
    struct cavlc_pool *pool; /* row-parallel coding (h264_writer_set_threads); NULL = everything on the calling thread */
};

h264_writer_t *h264_writer_new(int mbw, int mbh, int t8) {
    h264_writer_t *w = (h264_writer_t *)calloc(1, sizeof *w);
    if (!w) return NULL;
    w->mbw = mbw; w->mbh = mbh; w->t8 = t8;
    w->rbsp_cap = (size_t)mbw * mbh * 1024 + 1024;
    w->rbsp = (uint8_t *)malloc(w->rbsp_cap);
    w->tc_l = (uint8_t *)malloc((size_t)mbw * mbh * 16);
    w->tc_c = (uint8_t *)malloc((size_t)mbw * mbh * 8);
    w->i4m = (uint8_t *)malloc((size_t)mbw * mbh * 16);
    w->qmv = (int16_t *)calloc((size_t)mbw * mbh * 8, sizeof(int16_t));
    if (!w->rbsp || !w->tc_l || !w->tc_c || !w->i4m || !w->qmv) { h264_writer_free(w); return NULL; }
    return w;
}
void h264_writer_set_slice_rows(h264_writer_t *w, int rows) { if (w) w->slice_rows = rows > 0 ? rows : 0; }
void h264_writer_set_p_slices(h264_writer_t *w, int rows, int dbf_idc) { if (w) { w->pslice_rows = rows > 0 ? rows : 0; w->dbf_idc = dbf_idc == 2 ? 2 : 0; } }
static inline int slice_rows_of(const h264_writer_t *w, int is_idr) { const int r = is_idr ? w->slice_rows : w->pslice_rows; return r > 0 && r < w->mbh ? r : 0; }
static void cavlc_pool_free(struct cavlc_pool *p);
void h264_writer_free(h264_writer_t *w) {
    if (!w) return;
    if (w->pool) cavlc_pool_free(w->pool);
    free(w->rbsp); free(w->tc_l); free(w->tc_c); free(w->i4m); free(w->qmv); free(w);
}
size_t h264_max_au_bytes(int mbw, int mbh) { return (size_t)mbw * mbh * 1536 + 4096; }

static inline int ctx_luma(const h264_writer_t *w, int mbn, int mx, int top, int bx, int by) { /* top: the row above is available (same slice) */
    const uint8_t *t = w->tc_l + (size_t)mbn * 16;
    int na = -1, nb = -1;
    if (bx) na = t[by * 4 + bx - 1];
    else if (mx) na = t[-16 + by * 4 + 3];
    if (by) nb = t[(by - 1) * 4 + bx];
    else if (top) nb = t[-16 * w->mbw + 12 + bx];
    return (na >= 0 && nb >= 0) ? (na + nb + 1) >> 1 : (na >= 0 ? na : (nb >= 0 ? nb : 0));
}
static inline int ctx_chroma(const h264_writer_t *w, int mbn, int mx, int top, int c, int bx, int by) {
    const uint8_t *t = w->tc_c + (size_t)mbn * 8 + 4 * c;
    int na = -1, nb = -1;
    if (bx) na = t[by * 2];
    else if (mx) na = t[-8 + by * 2 + 1];
    if (by) nb = t[bx];
    else if (top) nb = t[-8 * w->mbw + 2 + bx];
    return (na >= 0 && nb >= 0) ? (na + nb + 1) >> 1 : (na >= 0 ? na : (nb >= 0 ? nb : 0));
}

/* One implementation for both level layouts.  dense: 408 int16 per macroblock (mi355enc_dev.h).  packed: the stream
 * written by levels_pack_kernel -- per macroblock, 32-byte blocks [I4x4 modes if mb_type == 2][I16x16 DC if NZ_LDC]
 * [luma block b for each set bit b][chroma DC if NZ_CBDC|NZ_CRDC][chroma AC block i for each set bit 16+i]; blocks that
 * are absent are all-zero by construction and read from k_zero_block. */
static const int16_t k_zero_block[16] __attribute__((aligned(32))) = {0};
static void slice_header(bits_t *bp, int first_mb, int is_idr, int frame_num, int idr_pic_id, int slice_qp, int dbf_idc) { /* 7.3.3 */
    bits_ue(bp, (uint32_t)first_mb);
    bits_ue(bp, is_idr ? 7 : 5);
    bits_ue(bp, 0);
    bits_put(bp, 8, (uint32_t)frame_num & 0xFF);
    if (is_idr) bits_ue(bp, (uint32_t)idr_pic_id);
    if (!is_idr) bits_put(bp, 2, 0);  /* num_ref_idx_active_override_flag, ref_pic_list_modification_flag_l0 */
    if (is_idr) bits_put(bp, 2, 0);   /* no_output_of_prior_pics_flag, long_term_reference_flag */
    else bits_put(bp, 1, 0);          /* adaptive_ref_pic_marking_mode_flag */
    bits_se(bp, slice_qp - 26);
    bits_ue(bp, (uint32_t)dbf_idc);   /* disable_deblocking_filter_idc */
    bits_se(bp, 0); bits_se(bp, 0);
}
/* What a range of macroblock rows leaves open at its two ends (P slices): the mb_skip_run before its first coded
 * macroblock is NOT written when `defer_first_run` is set -- the caller that concatenates ranges writes
 * ue(skips pending from earlier ranges + lead_skip) -- and the run after its last coded macroblock is returned. */
typedef struct { int has_coded, lead_skip, trail_skip; } rows_result_t;

/* 7.3.4 slice_data for macroblock rows [row0, row1).  The neighbour state (tc_l, tc_c, i4m of row row0-1) must be in
 * place.  `packed` points at the first block of row0 in the packed stream (or is NULL: dense levels). */
static rows_result_t code_rows(h264_writer_t *w, bits_t *bp, int row0, int row1, int is_idr, int slice_qp, const mb_info_t *mbi,
                               const int16_t *levels, const int16_t *packed, int defer_first_run) {
    const int mbw = w->mbw;
    bits_t b = *bp;
    rows_result_t res = {0, 0, 0};
    memset(w->tc_l + (size_t)row0 * mbw * 16, 0, (size_t)(row1 - row0) * mbw * 16);
    memset(w->tc_c + (size_t)row0 * mbw * 8, 0, (size_t)(row1 - row0) * mbw * 8);
    /* QP_Y,PRED of the range's first macroblock (7.4.5): the QP_Y of the last macroblock before it that sent an mb_qp_delta (Intra_16x16, or any
     * coded block), the slice's if there is none.  (One QP per picture: the slice's everywhere; adaptive quantisation: whatever that macroblock had.) */
    const int srows = slice_rows_of(w, is_idr); /* (a range never straddles two slices) */
    int skip = 0, prev_qp = slice_qp;
    for (int i = row0 * mbw - 1; i >= (srows ? row0 / srows * srows * mbw : 0); i--)
        if (mbi[i].mb_type == 0 || (mbi[i].nzmask & 0x07FFFFFFu) != 0) { prev_qp = mbi[i].qp; break; }
    for (int my = row0, mbn = row0 * mbw; my < row1; my++) {
        const int top = srows ? my % srows != 0 : my > 0; /* the row above belongs to this slice */
        for (int mx = 0; mx < mbw; mx++, mbn++) {
            const mb_info_t *m = mbi + mbn;
            const uint32_t nz = m->nzmask;
            const int intra = m->mb_type != 1, i16 = m->mb_type == 0;
            const int16_t *p_modes, *p_ldc, *p_cdc, *p_luma[16], *p_cac[8];
            if (packed) {
                const int has_slot = m->mb_type == 2 || mb_shape(m) != 0; /* the luma-DC slot travels for Intra_4x4 modes and for the vectors of partitions 1 .. 3 */
                p_modes = has_slot ? packed : k_zero_block; if (has_slot) packed += 16;
                p_ldc = (nz & NZ_LDC) ? packed : k_zero_block; if (nz & NZ_LDC) packed += 16;
                for (int i = 0; i < 16; i++) { if ((nz >> i) & 1) { p_luma[i] = packed; packed += 16; } else p_luma[i] = k_zero_block; }
                p_cdc = (nz & (NZ_CBDC | NZ_CRDC)) ? packed : k_zero_block; if (nz & (NZ_CBDC | NZ_CRDC)) packed += 16;
                for (int i = 0; i < 8; i++) { if ((nz >> (16 + i)) & 1) { p_cac[i] = packed; packed += 16; } else p_cac[i] = k_zero_block; }
            } else {
                const int16_t *lv = levels + (size_t)mbn * MB_LEVELS;
                p_modes = p_ldc = lv + L_LDC; p_cdc = lv + L_CDC;
                for (int i = 0; i < 16; i++) p_luma[i] = lv + L_LUMA + i * 16;
                for (int i = 0; i < 8; i++) p_cac[i] = lv + L_CAC + i * 16;
            }
            uint8_t *im = w->i4m + (size_t)mbn * 16;
            if (m->mb_type == 2) for (int i = 0; i < 16; i++) im[i] = (uint8_t)((nz & NZ_T8) ? p_modes[i >> 2] : p_modes[i]); /* (Intra_8x8: four modes, each at its block's four blkIdx) */
            int cbp_l = 0;
            if (i16) cbp_l = (nz & 0xFFFF) ? 15 : 0;
            else cbp_l = ((nz & 0x000F) ? 1 : 0) | ((nz & 0x00F0) ? 2 : 0) | ((nz & 0x0F00) ? 4 : 0) | ((nz & 0xF000) ? 8 : 0);
            const int cbp_c = (nz & 0x00FF0000u) ? 2 : ((nz & (NZ_CBDC | NZ_CRDC)) ? 1 : 0);
            if (!intra) {
                const int shape = mb_shape(m);
                int16_t *qv = w->qmv + (size_t)mbn * 8;
                set_qmv(qv, m, p_modes);
                if (!cbp_l && !cbp_c && !shape) { /* 8.4.1.1: P_Skip when the vector equals the inferred one */
                    int sx, sy;
                    predict_part(w->qmv, mbi, mbw, mx, my, top, 0, qv, 0, 0, 0, 0, 16, 1, &sx, &sy);
                    if (m->mvx == sx && m->mvy == sy) { skip++; continue; }
                }
                if (defer_first_run && !res.has_coded) res.lead_skip = skip; else bits_ue(&b, (uint32_t)skip);
                skip = 0; res.has_coded = 1;
                bits_ue(&b, (uint32_t)shape); /* P_L0_16x16, P_L0_L0_16x8, P_L0_L0_8x16, P_8x8 */
                if (shape == 3) for (int i = 0; i < 4; i++) bits_ue(&b, 0); /* sub_mb_type: P_L0_8x8 */
                unsigned done = 0;
                for (int i = 0; i < part_n[shape]; i++) { /* mvd_l0 of the partitions in order (quarter-sample units; ref_idx is not sent: one reference) */
                    const int8_t *g = part_geo[shape][i];
                    const int q0 = (g[1] >> 3) * 2 + (g[0] >> 3);
                    int px, py;
                    predict_part(w->qmv, mbi, mbw, mx, my, top, done, qv, shape, i, g[0], g[1], g[2], 0, &px, &py);
                    bits_se(&b, qv[2 * q0] - px);
                    bits_se(&b, qv[2 * q0 + 1] - py);
                    for (int q = 0; q < 4; q++) { const int qx = (q & 1) * 8, qy = (q >> 1) * 8; if (qx >= g[0] && qx < g[0] + g[2] && qy >= g[1] && qy < g[1] + g[3]) done |= 1u << q; }
                }
                bits_ue(&b, cbp_inter_code[cbp_c * 16 + cbp_l]);
                if (w->t8 && cbp_l) bits_put(&b, 1, (nz & NZ_T8) ? 1u : 0u); /* transform_size_8x8_flag */
            } else {
                if (!is_idr) { if (defer_first_run && !res.has_coded) res.lead_skip = skip; else bits_ue(&b, (uint32_t)skip); skip = 0; }
                res.has_coded = 1;
                if (i16) {
                    int t = 1 + m->i16_mode + 4 * cbp_c + (cbp_l ? 12 : 0); /* Table 7-11 */
                    bits_ue(&b, (uint32_t)(is_idr ? t : t + 5));
                    bits_ue(&b, m->chroma_mode);
                } else { /* I_NxN: sixteen Intra_4x4 modes (8.3.1.1) or four Intra_8x8 modes (8.3.2.1), each predicted from the blocks left and above */
                    const int i8 = (nz & NZ_T8) != 0;
                    bits_ue(&b, is_idr ? 0u : 5u);
                    if (w->t8) bits_put(&b, 1, (uint32_t)i8); /* transform_size_8x8_flag: Intra_8x8 or Intra_4x4 */
                    if (i8) { /* im[] holds an 8x8 block's mode at all four of its 4x4 blkIdx, so that Intra_4x4 neighbours read it the way 8.3.1.1 says */
                        for (int b8 = 0; b8 < 4; b8++) {
                            int ma = -1, mb_ = -1;
                            if (b8 & 1) ma = im[4 * (b8 - 1)];
                            else if (mx) ma = m[-1].mb_type == 2 ? im[-16 + 4 * (b8 + 1) + 1] : 2;       /* the left macroblock's 4x4 block 4 n + 1 */
                            if (b8 >> 1) mb_ = im[4 * (b8 - 2)];
                            else if (top) mb_ = m[-mbw].mb_type == 2 ? im[-(ptrdiff_t)mbw * 16 + 4 * (b8 + 2) + 2] : 2; /* the macroblock above: 4 n + 2 */
                            const int pm = (ma < 0 || mb_ < 0) ? 2 : (ma < mb_ ? ma : mb_), mode = im[4 * b8];
                            if (mode == pm) bits_put(&b, 1, 1);
                            else bits_put(&b, 4, (uint32_t)(mode < pm ? mode : mode - 1));
                        }
                    } else
                    for (int blk = 0; blk < 16; blk++) {
                        const int r = blk_to_raster[blk], bx = r & 3, by = r >> 2;
                        int ma = -1, mb_ = -1;
                        if (bx) ma = im[blk_to_raster[by * 4 + bx - 1]];
                        else if (mx) ma = m[-1].mb_type == 2 ? im[-16 + blk_to_raster[by * 4 + 3]] : 2;
                        if (by) mb_ = im[blk_to_raster[(by - 1) * 4 + bx]];
                        else if (top) mb_ = m[-mbw].mb_type == 2 ? im[-(ptrdiff_t)mbw * 16 + blk_to_raster[12 + bx]] : 2;
                        const int pm = (ma < 0 || mb_ < 0) ? 2 : (ma < mb_ ? ma : mb_), mode = im[blk];
                        if (mode == pm) bits_put(&b, 1, 1);
                        else bits_put(&b, 4, (uint32_t)(mode < pm ? mode : mode - 1)); /* flag 0 + 3-bit rem */
                    }
                    bits_ue(&b, m->chroma_mode);
                    bits_ue(&b, cbp_intra_code[cbp_c * 16 + cbp_l]);
                }
            }
            if (i16 || cbp_l || cbp_c) { bits_se(&b, (int)m->qp - prev_qp); prev_qp = m->qp; }
            uint8_t *tl = w->tc_l + (size_t)mbn * 16;
            if (i16) put_block16(&b, p_ldc, 0, ctx_luma(w, mbn, mx, top, 0, 0));
            if (cbp_l)
                for (int blk = 0; blk < 16; blk++) {
                    if (!(cbp_l & (1 << (blk >> 2)))) continue;
                    const int r = blk_to_raster[blk], bx = r & 3, by = r >> 2;
                    const int nC = ctx_luma(w, mbn, mx, top, bx, by);
                    if ((nz >> blk) & 1) tl[r] = (uint8_t)put_block16(&b, p_luma[blk], i16, nC);
                    else { const int cls = nC < 2 ? 0 : nC < 4 ? 1 : nC < 8 ? 2 : 3; bits_put(&b, vlc_coeff_token[cls][0][0].len, vlc_coeff_token[cls][0][0].bits); }
                }
            if (cbp_c) {
                put_chroma_dc(&b, p_cdc);
                put_chroma_dc(&b, p_cdc + 4);
                if (cbp_c == 2)
                    for (int c = 0; c < 2; c++)
                        for (int blk = 0; blk < 4; blk++) {
                            const int nC = ctx_chroma(w, mbn, mx, top, c, blk & 1, blk >> 1);
                            w->tc_c[(size_t)mbn * 8 + 4 * c + blk] = (uint8_t)put_block16(&b, p_cac[4 * c + blk], 1, nC);
                        }
            }
        }
    }
    if (!res.has_coded) res.lead_skip = skip; else res.trail_skip = skip;
    *bp = b;
    return res;
}

/* Neighbour state of one macroblock row from the packed stream, without coding it: TotalCoeff of every coded luma /
 * chroma-AC block and the Intra_4x4 modes -- what the row below reads through ctx_luma / ctx_chroma / the mode predictor. */
static void fill_ctx_row(h264_writer_t *w, int row, const mb_info_t *mbi, const int16_t *packed) {
    const int mbw = w->mbw;
    const __m128i z = _mm_setzero_si128();
    for (int mx = 0, mbn = row * mbw; mx < mbw; mx++, mbn++) {
        const mb_info_t *m = mbi + mbn;
        const uint32_t nz = m->nzmask;
        uint8_t *tl = w->tc_l + (size_t)mbn * 16, *tc = w->tc_c + (size_t)mbn * 8, *im = w->i4m + (size_t)mbn * 16;
        memset(tl, 0, 16); memset(tc, 0, 8);
        if (m->mb_type == 2) { for (int i = 0; i < 16; i++) im[i] = (uint8_t)((nz & NZ_T8) ? packed[i >> 2] : packed[i]); packed += 16; }
        else if (m->mb_type == 1) { const int slot = mb_shape(m) != 0; set_qmv(w->qmv + (size_t)mbn * 8, m, slot ? packed : k_zero_block); if (slot) packed += 16; }
        if (nz & NZ_LDC) packed += 16;
        const int ac_only = m->mb_type == 0; /* Intra16x16: coefficient 0 travels in the DC block */
        for (int i = 0; i < 16; i++)
            if ((nz >> i) & 1) {
                __m128i lo = _mm_loadu_si128((const __m128i *)packed), hi = _mm_loadu_si128((const __m128i *)(packed + 8));
                unsigned nzm = ~(unsigned)_mm_movemask_epi8(_mm_packs_epi16(_mm_cmpeq_epi16(lo, z), _mm_cmpeq_epi16(hi, z))) & 0xFFFFu;
                if (ac_only) nzm >>= 1;
                tl[blk_to_raster[i]] = (uint8_t)__builtin_popcount(nzm);
                packed += 16;
            }
        if (nz & (NZ_CBDC | NZ_CRDC)) packed += 16;
        for (int i = 0; i < 8; i++)
            if ((nz >> (16 + i)) & 1) {
                __m128i lo = _mm_loadu_si128((const __m128i *)packed), hi = _mm_loadu_si128((const __m128i *)(packed + 8));
                unsigned nzm = ~(unsigned)_mm_movemask_epi8(_mm_packs_epi16(_mm_cmpeq_epi16(lo, z), _mm_cmpeq_epi16(hi, z))) & 0xFFFFu;
                tc[i] = (uint8_t)__builtin_popcount(nzm >> 1);
                packed += 16;
            }
    }
}

/* the packed stream behind `n` macroblocks starting at `m` (their blocks are counted from the records) */
static const int16_t *skip_packed_rows(const mb_info_t *m, int n, const int16_t *packed) {
    for (int i = 0; i < n; i++) {
        const uint32_t nz = m[i].nzmask;
        int blocks = (m[i].mb_type == 2 || mb_shape(&m[i]) != 0) + ((nz & NZ_LDC) != 0) + __builtin_popcount(nz & 0xFFFFu) + ((nz & (NZ_CBDC | NZ_CRDC)) != 0) + __builtin_popcount((nz >> 16) & 0xFFu);
        packed += 16 * blocks;
    }
    return packed;
}
static size_t write_slice_impl(h264_writer_t *w, uint8_t *out, size_t cap, int is_idr, int frame_num, int idr_pic_id,
                               int slice_qp, const mb_info_t *mbi, const int16_t *levels, const int16_t *packed) {
    const int srows = slice_rows_of(w, is_idr) ? slice_rows_of(w, is_idr) : w->mbh;
    size_t total = 0;
    for (int row0 = 0; row0 < w->mbh; row0 += srows) { /* one NAL unit per slice */
        const int row1 = row0 + srows < w->mbh ? row0 + srows : w->mbh;
        bits_t b;
        bits_init(&b, w->rbsp, w->rbsp_cap);
        slice_header(&b, row0 * w->mbw, is_idr, frame_num, idr_pic_id, slice_qp, w->dbf_idc);
        if (packed && row0 > 0) packed = skip_packed_rows(mbi + (size_t)(row0 - srows) * w->mbw, srows * w->mbw, packed);
        rows_result_t r = code_rows(w, &b, row0, row1, is_idr, slice_qp, mbi, levels, packed, 0);
        const int tail = r.has_coded ? r.trail_skip : r.lead_skip;
        if (!is_idr && tail) bits_ue(&b, (uint32_t)tail);
        size_t n = bits_finish(&b, w->rbsp);
        if (b.overflow) return 0;
        n = emit_nal(out + total, cap - total, is_idr ? 3 : 2, is_idr ? 5 : 1, w->rbsp, n);
        if (!n) return 0;
        total += n;
    }
    return total;
}

/* ------------------------------------------------------------------ row-parallel CAVLC
 * The slice stays ONE slice: ranges of macroblock rows are coded concurrently into private bit buffers and concatenated
 * bit-exactly.  That works because nothing in CAVLC depends on the bits before a macroblock, only on neighbour DATA:
 *   - nC contexts and Intra_4x4 mode prediction read the row above -> each worker first derives that row's TotalCoeff /
 *     modes from the packed stream (fill_ctx_row), it does not wait for the worker that codes it;
 *   - motion-vector prediction and P_Skip inference read records only;
 *   - mb_qp_delta is 0 everywhere (one QP per picture);
 *   - mb_skip_run crosses range boundaries -> the first run of a range is left to the stitcher.
 * Each worker owns a private h264_writer (context arrays + bit buffer): no shared mutable state. */
typedef struct {              /* one chunk of consecutive macroblock rows */
    int row0, row1;
    const uint8_t *base;       /* where its bits start (inside the buffer of the thread that coded it) */
    bits_t bits;               /* state at its end */
    rows_result_t res;
} cavlc_job_t;
/* The chunks of a picture (more of them than threads) are claimed dynamically: a worker that wakes up late simply finds
 * fewer chunks left, so the caller never waits for a sleeping thread -- only for chunks somebody has started.  With a
 * static split the slowest wake-up of a 16.7 ms-idle worker set the latency of every live picture. */
struct cavlc_pool {
    int n;                     /* threads including the calling one */
    pthread_t *th;
    h264_writer_t **wr;        /* private writer (context arrays + bit buffer) per thread */
    uint8_t **cursor;          /* per thread: where its next chunk's bits go */
    cavlc_job_t *job; int job_cap;
    pthread_mutex_t mu;
    pthread_cond_t cv_go, cv_done;
    unsigned long long generation;
    int nchunk, next, done, stop;
    /* per call */
    int is_idr, slice_qp;
    const mb_info_t *mbi;
    const int16_t *packed;
    const uint32_t *row_off;
};
static void cavlc_run_chunk(struct cavlc_pool *p, int k, int c) {
    cavlc_job_t *j = &p->job[c];
    h264_writer_t *w = p->wr[k];
    j->base = p->cursor[k];
    bits_init(&j->bits, p->cursor[k], (size_t)(w->rbsp + w->rbsp_cap - p->cursor[k]));
    if (j->row0 > 0) fill_ctx_row(w, j->row0 - 1, p->mbi, p->packed + (size_t)p->row_off[j->row0 - 1] * 16);
    j->res = code_rows(w, &j->bits, j->row0, j->row1, p->is_idr, p->slice_qp, p->mbi, NULL, p->packed + (size_t)p->row_off[j->row0] * 16, 1);
    p->cursor[k] = j->bits.p + 8; /* past the partial word the chunk's state still holds */
    if (p->cursor[k] > w->rbsp + w->rbsp_cap) p->cursor[k] = w->rbsp + w->rbsp_cap;
}
static void cavlc_work(struct cavlc_pool *p, int k) { /* claim chunks until none is left */
    for (;;) {
        pthread_mutex_lock(&p->mu);
        const int c = p->next < p->nchunk ? p->next++ : -1;
        pthread_mutex_unlock(&p->mu);
        if (c < 0) return;
        cavlc_run_chunk(p, k, c);
        pthread_mutex_lock(&p->mu);
        if (++p->done == p->nchunk) pthread_cond_signal(&p->cv_done);
        pthread_mutex_unlock(&p->mu);
    }
}
typedef struct { struct cavlc_pool *p; int k; } cavlc_arg_t;
static void *cavlc_thread(void *arg) {
    cavlc_arg_t a = *(cavlc_arg_t *)arg;
    free(arg);
    unsigned long long seen = 0;
    for (;;) {
        pthread_mutex_lock(&a.p->mu);
        while (!a.p->stop && a.p->generation == seen) pthread_cond_wait(&a.p->cv_go, &a.p->mu);
        if (a.p->stop) { pthread_mutex_unlock(&a.p->mu); return NULL; }
        seen = a.p->generation;
        pthread_mutex_unlock(&a.p->mu);
        cavlc_work(a.p, a.k);
    }
}
static void cavlc_pool_free(struct cavlc_pool *p) {
    if (!p) return;
    pthread_mutex_lock(&p->mu); p->stop = 1; pthread_cond_broadcast(&p->cv_go); pthread_mutex_unlock(&p->mu);
    for (int k = 1; k < p->n; k++) if (p->th && p->th[k]) pthread_join(p->th[k], NULL);
    for (int k = 0; k < p->n; k++) if (p->wr && p->wr[k]) { p->wr[k]->pool = NULL; h264_writer_free(p->wr[k]); }
    pthread_mutex_destroy(&p->mu); pthread_cond_destroy(&p->cv_go); pthread_cond_destroy(&p->cv_done);
    free(p->th); free(p->wr); free(p->cursor); free(p->job); free(p);
}
int h264_writer_set_threads(h264_writer_t *w, int threads) {
    if (!w) return -1;
    if (w->pool) { cavlc_pool_free(w->pool); w->pool = NULL; }
    if (threads > w->mbh) threads = w->mbh;
    if (threads <= 1) return 0;
    struct cavlc_pool *p = (struct cavlc_pool *)calloc(1, sizeof *p);
    if (!p) return -1;
    p->n = threads;
    p->job_cap = 4 * threads < w->mbh ? 4 * threads : w->mbh;
    p->th = (pthread_t *)calloc((size_t)threads, sizeof *p->th);
    p->wr = (h264_writer_t **)calloc((size_t)threads, sizeof *p->wr);
    p->cursor = (uint8_t **)calloc((size_t)threads, sizeof *p->cursor);
    p->job = (cavlc_job_t *)calloc((size_t)p->job_cap, sizeof *p->job);
    pthread_mutex_init(&p->mu, NULL); pthread_cond_init(&p->cv_go, NULL); pthread_cond_init(&p->cv_done, NULL);
    int ok = p->th && p->wr && p->cursor && p->job;
    for (int k = 0; ok && k < threads; k++) { p->wr[k] = h264_writer_new(w->mbw, w->mbh, w->t8); ok = p->wr[k] != NULL; }
    for (int k = 1; ok && k < threads; k++) {
        cavlc_arg_t *a = (cavlc_arg_t *)malloc(sizeof *a);
        if (!a) { ok = 0; break; }
        a->p = p; a->k = k;
        if (pthread_create(&p->th[k], NULL, cavlc_thread, a)) { free(a); p->th[k] = 0; ok = 0; }
    }
    if (!ok) { cavlc_pool_free(p); return -1; }
    w->pool = p;
    return 0;
}
static void bits_append(bits_t *dst, const bits_t *src, const uint8_t *base) { /* everything `src` has collected so far */
    const uint8_t *q = base;
    for (; q + 4 <= src->p; q += 4) bits_put(dst, 32, ((uint32_t)q[0] << 24) | ((uint32_t)q[1] << 16) | ((uint32_t)q[2] << 8) | q[3]);
    for (; q < src->p; q++) bits_put(dst, 8, *q);
    if (src->n) bits_put(dst, src->n, (uint32_t)(src->acc & ((1ull << src->n) - 1)));
    if (src->overflow) dst->overflow = 1;
}
size_t h264_write_slice_packed_rows(h264_writer_t *w, uint8_t *out, size_t cap, int is_idr, int frame_num, int idr_pic_id,
                                    int slice_qp, const mb_info_t *mbi, const int16_t *packed, const uint32_t *row_off) {
    struct cavlc_pool *p = w->pool;
    if (!p || !row_off) return write_slice_impl(w, out, cap, is_idr, frame_num, idr_pic_id, slice_qp, mbi, NULL, packed);
    p->is_idr = is_idr; p->slice_qp = slice_qp; p->mbi = mbi; p->packed = packed; p->row_off = row_off;
    /* chunks of about four rows, at least one per thread and at most four: small enough to balance rows of unequal cost and to
     * leave late threads nothing to hold up, large enough that deriving the context of the row above stays a small share.  A chunk
     * never straddles two slices: every slice is cut into its own chunks. */
    const int srows = slice_rows_of(w, is_idr) ? slice_rows_of(w, is_idr) : w->mbh;
    const int nslice = (w->mbh + srows - 1) / srows;
    int want = w->mbh / 4;
    if (want < p->n) want = p->n;
    if (want > p->job_cap) want = p->job_cap;
    int per = (want + nslice - 1) / nslice; /* chunks per slice */
    if (per < 1) per = 1;
    while (per > 1 && per * nslice > p->job_cap) per--;
    if (per * nslice > p->job_cap) return write_slice_impl(w, out, cap, is_idr, frame_num, idr_pic_id, slice_qp, mbi, NULL, packed);
    int nchunk = 0;
    for (int sl = 0; sl < nslice; sl++) {
        const int r0 = sl * srows, r1 = r0 + srows < w->mbh ? r0 + srows : w->mbh, rows = r1 - r0, k = per < rows ? per : rows;
        for (int c = 0; c < k; c++) { p->job[nchunk].row0 = r0 + (int)((long long)rows * c / k); p->job[nchunk].row1 = r0 + (int)((long long)rows * (c + 1) / k); nchunk++; }
    }
    for (int k = 0; k < p->n; k++) { p->cursor[k] = p->wr[k]->rbsp; p->wr[k]->slice_rows = w->slice_rows; p->wr[k]->pslice_rows = w->pslice_rows; p->wr[k]->dbf_idc = w->dbf_idc; }
    pthread_mutex_lock(&p->mu);
    p->nchunk = nchunk; p->next = 0; p->done = 0; p->generation++;
    pthread_cond_broadcast(&p->cv_go);
    pthread_mutex_unlock(&p->mu);
    cavlc_work(p, 0);
    pthread_mutex_lock(&p->mu);
    while (p->done < p->nchunk) pthread_cond_wait(&p->cv_done, &p->mu);
    pthread_mutex_unlock(&p->mu);
    size_t total = 0;
    for (int c = 0; c < nchunk;) { /* one NAL unit per slice: its header, then its chunks bit-exactly one after the other */
        const int first_row = p->job[c].row0, end_row = first_row + srows < w->mbh ? first_row + srows : w->mbh;
        bits_t b;
        bits_init(&b, w->rbsp, w->rbsp_cap);
        slice_header(&b, first_row * w->mbw, is_idr, frame_num, idr_pic_id, slice_qp, w->dbf_idc);
        int pending = 0;
        for (; c < nchunk && p->job[c].row0 < end_row; c++) {
            const cavlc_job_t *j = &p->job[c];
            if (j->res.has_coded) {
                if (!is_idr) bits_ue(&b, (uint32_t)(pending + j->res.lead_skip));
                bits_append(&b, &j->bits, j->base);
                pending = j->res.trail_skip;
            } else pending += j->res.lead_skip;
        }
        if (!is_idr && pending) bits_ue(&b, (uint32_t)pending);
        size_t n = bits_finish(&b, w->rbsp);
        if (b.overflow) return 0;
        n = emit_nal(out + total, cap - total, is_idr ? 3 : 2, is_idr ? 5 : 1, w->rbsp, n);
        if (!n) return 0;
        total += n;
    }
    return total;
}

size_t h264_pack_levels(int mbw, int mbh, const mb_info_t *mbi, const int16_t *levels, int16_t *packed, uint32_t *row_off) {
    const size_t nmb = (size_t)mbw * mbh;
    size_t nblk = 0;
    for (size_t mb = 0; mb < nmb; mb++) {
        const uint32_t nz = mbi[mb].nzmask;
        const int16_t *lv = levels + mb * MB_LEVELS;
        if (mb % (size_t)mbw == 0) row_off[mb / (size_t)mbw] = (uint32_t)nblk;
        if (mbi[mb].mb_type == 2 || mb_shape(&mbi[mb]) != 0) memcpy(packed + 16 * nblk++, lv + L_LDC, 32); /* Intra_4x4 modes / the vectors of partitions 1 .. 3 */
        if (nz & NZ_LDC) memcpy(packed + 16 * nblk++, lv + L_LDC, 32);
        for (int i = 0; i < 16; i++) if ((nz >> i) & 1) memcpy(packed + 16 * nblk++, lv + L_LUMA + 16 * i, 32);
        if (nz & (NZ_CBDC | NZ_CRDC)) memcpy(packed + 16 * nblk++, lv + L_CDC, 32);
        for (int i = 0; i < 8; i++) if ((nz >> (16 + i)) & 1) memcpy(packed + 16 * nblk++, lv + L_CAC + 16 * i, 32);
    }
    return nblk;
}

size_t h264_write_slice(h264_writer_t *w, uint8_t *out, size_t cap, int is_idr, int frame_num, int idr_pic_id,
                        int slice_qp, const mb_info_t *mbi, const int16_t *levels) {
    return write_slice_impl(w, out, cap, is_idr, frame_num, idr_pic_id, slice_qp, mbi, levels, NULL);
}
size_t h264_write_slice_packed(h264_writer_t *w, uint8_t *out, size_t cap, int is_idr, int frame_num, int idr_pic_id,
                               int slice_qp, const mb_info_t *mbi, const int16_t *packed) {
    return write_slice_impl(w, out, cap, is_idr, frame_num, idr_pic_id, slice_qp, mbi, NULL, packed);
}
