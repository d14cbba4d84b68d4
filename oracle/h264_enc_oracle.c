/*
 * oracle/h264_enc_oracle.c -- TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see h264_oracle.h).
 *
 * Scalar CPU restatement of the encoder path that replaces the reference's
 * `x264enc speed-preset=2 key-int-max=60 name=venc_kbps` pipeline token
 * (/root/reference/pipeline/generic/x264_superfast_camlink:5; instantiated by
 * /root/reference/src/io/pipeline_loader.c:59; bitrate driven through
 * /root/reference/src/gst/encoder_control.c:53).  The reference tree contains no codec
 * arithmetic, so every function cites the clause of ITU-T H.264 (04/2017 numbering) it
 * restates, or says "encoder choice" where the standard leaves the encoder free.
 *
 * Stream shape: Constrained Baseline, CAVLC, one slice per picture, IDR every `gop`
 * pictures, P pictures with one reference, macroblock types I16x16 (I pictures) and
 * P_L0_16x16 / P_Skip (P pictures), integer-pel motion, in-loop deblocking on.
 */
#include "h264_oracle.h"
#include "h264_tables_enc.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CLIP3(lo, hi, v) ((v) < (lo) ? (lo) : ((v) > (hi) ? (hi) : (v)))
static inline int clip1(int v) { return CLIP3(0, 255, v); }
static inline int iabs(int v) { return v < 0 ? -v : v; }
#define MAX_LEVEL 2047 /* keeps level_prefix <= 15 (Baseline, 9.2.2.1) */

/* ================================================================== bit writer */
typedef struct {
    uint8_t *buf;
    size_t cap, pos;
    uint64_t acc;
    int nbits;
    int overflow;
} bw_t;

static void bw_init(bw_t *b, uint8_t *buf, size_t cap) {
    b->buf = buf; b->cap = cap; b->pos = 0; b->acc = 0; b->nbits = 0; b->overflow = 0;
}
static void bw_put(bw_t *b, int n, uint32_t v) {
    if (n == 0) return;
    b->acc = (b->acc << n) | (v & (n == 32 ? 0xFFFFFFFFu : ((1u << n) - 1)));
    b->nbits += n;
    while (b->nbits >= 8) {
        b->nbits -= 8;
        if (b->pos < b->cap) b->buf[b->pos++] = (uint8_t)(b->acc >> b->nbits);
        else b->overflow = 1;
    }
}
/* 9.1 Exp-Golomb */
int orc_ue_bits(uint32_t v, uint32_t *code) {
    uint32_t k = v + 1;
    int len = 0;
    while ((k >> len) > 1) len++;
    if (code) *code = k;
    return 2 * len + 1;
}
static void bw_ue(bw_t *b, uint32_t v) {
    uint32_t code;
    int n = orc_ue_bits(v, &code);
    if (n > 32) { bw_put(b, n - 32, 0); n = 32; }
    bw_put(b, n, code);
}
static void bw_se(bw_t *b, int v) { bw_ue(b, v > 0 ? (uint32_t)(2 * v - 1) : (uint32_t)(-2 * v)); }
static void bw_trailing(bw_t *b) { /* 7.3.2.11 rbsp_trailing_bits */
    bw_put(b, 1, 1);
    if (b->nbits) bw_put(b, 8 - b->nbits, 0);
}

/* 7.4.1.1 emulation prevention: 00 00 {00,01,02,03} -> 00 00 03 xx */
size_t orc_nal_escape(const uint8_t *rbsp, size_t n, uint8_t *out, size_t cap) {
    size_t o = 0;
    int zeros = 0;
    for (size_t i = 0; i < n; i++) {
        if (zeros >= 2 && rbsp[i] <= 3) {
            if (o < cap) out[o] = 3;
            o++;
            zeros = 0;
        }
        if (o < cap) out[o] = rbsp[i];
        o++;
        zeros = rbsp[i] == 0 ? zeros + 1 : 0;
    }
    return o;
}
/* Annex B start code + NAL header + escaped payload.  Returns bytes written (0 = no room). */
static size_t write_nal(uint8_t *out, size_t cap, int ref_idc, int type, const uint8_t *rbsp, size_t n) {
    if (cap < 5) return 0;
    out[0] = 0; out[1] = 0; out[2] = 0; out[3] = 1;
    out[4] = (uint8_t)((ref_idc << 5) | type);
    size_t e = orc_nal_escape(rbsp, n, out + 5, cap - 5);
    if (e > cap - 5) return 0;
    return 5 + e;
}

/* ================================================================== VLC tables */
typedef struct { uint8_t len; uint16_t bits; } vlc_t;
static vlc_t t_coeff_token[4][17][4];
static vlc_t t_coeff_token_cdc[5][4];
static vlc_t t_total_zeros[15][16];
static vlc_t t_total_zeros_cdc[3][4];
static vlc_t t_run_before[7][15];
static int tables_ready;

static vlc_t parse_vlc(const char *s) {
    vlc_t v = {0, 0};
    if (!s) return v;
    for (; *s; s++) { v.bits = (uint16_t)((v.bits << 1) | (*s == '1')); v.len++; }
    return v;
}
static void init_tables(void) {
    if (tables_ready) return;
    for (int t = 0; t < 3; t++)
        for (int c = 0; c < 17; c++)
            for (int o = 0; o < 4; o++) t_coeff_token[t][c][o] = parse_vlc(k_coeff_token_str[t][c][o]);
    /* nC >= 8: 6-bit FLC (Table 9-5 last column) */
    for (int c = 0; c < 17; c++)
        for (int o = 0; o < 4; o++) {
            vlc_t v = {0, 0};
            if (c == 0 && o == 0) { v.len = 6; v.bits = 3; }
            else if (c > 0 && o <= c && o < 4) { v.len = 6; v.bits = (uint16_t)(((c - 1) << 2) | o); }
            t_coeff_token[3][c][o] = v;
        }
    for (int c = 0; c < 5; c++)
        for (int o = 0; o < 4; o++) t_coeff_token_cdc[c][o] = parse_vlc(k_coeff_token_cdc_str[c][o]);
    for (int i = 0; i < 15; i++)
        for (int z = 0; z < 16; z++) t_total_zeros[i][z] = parse_vlc(k_total_zeros_str[i][z]);
    for (int i = 0; i < 3; i++)
        for (int z = 0; z < 4; z++) t_total_zeros_cdc[i][z] = parse_vlc(k_total_zeros_cdc_str[i][z]);
    for (int i = 0; i < 7; i++)
        for (int r = 0; r < 15; r++) t_run_before[i][r] = parse_vlc(k_run_before_str[i][r]);
    tables_ready = 1;
}

/* FNV-1a checksums of the tables, for the KAT in tests/ */
static uint32_t fnv(uint32_t h, const void *p, size_t n) {
    const uint8_t *b = (const uint8_t *)p;
    for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 16777619u; }
    return h;
}
uint32_t orc_table_checksum(int which) {
    init_tables();
    uint32_t h = 2166136261u;
    switch (which) {
    case 0: for (int t = 0; t < 4; t++) for (int c = 0; c < 17; c++) for (int o = 0; o < 4; o++) {
                h = fnv(h, &t_coeff_token[t][c][o].len, 1); h = fnv(h, &t_coeff_token[t][c][o].bits, 2); } break;
    case 1: for (int c = 0; c < 5; c++) for (int o = 0; o < 4; o++) {
                h = fnv(h, &t_coeff_token_cdc[c][o].len, 1); h = fnv(h, &t_coeff_token_cdc[c][o].bits, 2); } break;
    case 2: for (int i = 0; i < 15; i++) for (int z = 0; z < 16; z++) {
                h = fnv(h, &t_total_zeros[i][z].len, 1); h = fnv(h, &t_total_zeros[i][z].bits, 2); } break;
    case 3: for (int i = 0; i < 3; i++) for (int z = 0; z < 4; z++) {
                h = fnv(h, &t_total_zeros_cdc[i][z].len, 1); h = fnv(h, &t_total_zeros_cdc[i][z].bits, 2); } break;
    case 4: for (int i = 0; i < 7; i++) for (int r = 0; r < 15; r++) {
                h = fnv(h, &t_run_before[i][r].len, 1); h = fnv(h, &t_run_before[i][r].bits, 2); } break;
    case 5: h = fnv(h, k_cbp_to_codenum_intra, 48); h = fnv(h, k_cbp_to_codenum_inter, 48); break;
    case 6: h = fnv(h, k_zigzag4, 16); h = fnv(h, k_blk_x, 16); h = fnv(h, k_blk_y, 16); break;
    case 7: h = fnv(h, k_dequant_v, sizeof k_dequant_v); h = fnv(h, k_quant_mf, sizeof k_quant_mf); break;
    case 8: h = fnv(h, k_chroma_qp, 52); break;
    case 9: h = fnv(h, k_alpha, 52); h = fnv(h, k_beta, 52); h = fnv(h, k_tc0, sizeof k_tc0); break;
    case 10: h = fnv(h, k_me_lambda, 52); break;
    default: return 0;
    }
    return h;
}
/* raw access for the cross-check against the decoder's transcription */
int orc_enc_vlc(int table, int a, int b, int c, int *len, int *bits) {
    init_tables();
    vlc_t v = {0, 0};
    switch (table) {
    case 0: if (a < 4 && b < 17 && c < 4) v = t_coeff_token[a][b][c]; break;
    case 1: if (b < 5 && c < 4) v = t_coeff_token_cdc[b][c]; break;
    case 2: if (b < 15 && c < 16) v = t_total_zeros[b][c]; break;
    case 3: if (b < 3 && c < 4) v = t_total_zeros_cdc[b][c]; break;
    case 4: if (b < 7 && c < 15) v = t_run_before[b][c]; break;
    default: break;
    }
    *len = v.len; *bits = v.bits;
    return v.len != 0;
}
int orc_enc_cbp_codenum(int intra, int cbp) { return intra ? k_cbp_to_codenum_intra[cbp] : k_cbp_to_codenum_inter[cbp]; }
int orc_me_lambda(int qp) { return k_me_lambda[CLIP3(0, 51, qp)]; }

/* ================================================================== transforms */
static inline int pos_class(int pos) { /* pos = y*4+x */
    int x = pos & 3, y = pos >> 2;
    if (!(x & 1) && !(y & 1)) return 0;
    if ((x & 1) && (y & 1)) return 1;
    return 2;
}
/* forward 4x4 core transform Y = Cf X Cf^T (encoder side of 8.5.12) */
void orc_fdct4(const int16_t in[16], int16_t out[16]) {
    int tmp[16];
    for (int i = 0; i < 4; i++) {
        int a = in[i * 4 + 0], b = in[i * 4 + 1], c = in[i * 4 + 2], d = in[i * 4 + 3];
        int s03 = a + d, d03 = a - d, s12 = b + c, d12 = b - c;
        tmp[i * 4 + 0] = s03 + s12;
        tmp[i * 4 + 1] = 2 * d03 + d12;
        tmp[i * 4 + 2] = s03 - s12;
        tmp[i * 4 + 3] = d03 - 2 * d12;
    }
    for (int j = 0; j < 4; j++) {
        int a = tmp[0 * 4 + j], b = tmp[1 * 4 + j], c = tmp[2 * 4 + j], d = tmp[3 * 4 + j];
        int s03 = a + d, d03 = a - d, s12 = b + c, d12 = b - c;
        out[0 * 4 + j] = (int16_t)(s03 + s12);
        out[1 * 4 + j] = (int16_t)(2 * d03 + d12);
        out[2 * 4 + j] = (int16_t)(s03 - s12);
        out[3 * 4 + j] = (int16_t)(d03 - 2 * d12);
    }
}
/* 8.5.12.2 inverse 4x4 transform (rows, then columns), (x+32)>>6, add to prediction.
 * `step` is the byte distance between horizontally adjacent samples (2 for NV12 chroma). */
static void idct4_add_step(const int32_t d[16], uint8_t *dst, int stride, int step) {
    int f[16];
    for (int i = 0; i < 4; i++) {
        int e0 = d[i * 4 + 0] + d[i * 4 + 2];
        int e1 = d[i * 4 + 0] - d[i * 4 + 2];
        int e2 = (d[i * 4 + 1] >> 1) - d[i * 4 + 3];
        int e3 = d[i * 4 + 1] + (d[i * 4 + 3] >> 1);
        f[i * 4 + 0] = e0 + e3; f[i * 4 + 1] = e1 + e2;
        f[i * 4 + 2] = e1 - e2; f[i * 4 + 3] = e0 - e3;
    }
    for (int j = 0; j < 4; j++) {
        int g0 = f[0 * 4 + j] + f[2 * 4 + j];
        int g1 = f[0 * 4 + j] - f[2 * 4 + j];
        int g2 = (f[1 * 4 + j] >> 1) - f[3 * 4 + j];
        int g3 = f[1 * 4 + j] + (f[3 * 4 + j] >> 1);
        int h[4] = {g0 + g3, g1 + g2, g1 - g2, g0 - g3};
        for (int i = 0; i < 4; i++) {
            uint8_t *p = dst + (size_t)i * stride + j * step;
            *p = (uint8_t)clip1(*p + ((h[i] + 32) >> 6));
        }
    }
}
void orc_idct4_add(const int32_t d[16], uint8_t *dst, int stride) { idct4_add_step(d, dst, stride, 1); }
/* encoder choice: dead-zone quantiser, f = 2^qbits/3 intra, /6 inter; |level| <= MAX_LEVEL */
int orc_quant4(int coef, int qp, int pos, int intra) {
    int qbits = 15 + qp / 6;
    int f = (1 << qbits) / (intra ? 3 : 6);
    int mf = k_quant_mf[qp % 6][pos_class(pos)];
    int l = (int)(((int64_t)iabs(coef) * mf + f) >> qbits);
    if (l > MAX_LEVEL) l = MAX_LEVEL;
    return coef < 0 ? -l : l;
}
static int quant_dc(int coef, int qp, int intra) { /* luma-DC / chroma-DC: one more bit */
    int qbits = 16 + qp / 6;
    int f = (1 << qbits) / (intra ? 3 : 6);
    int mf = k_quant_mf[qp % 6][0];
    int l = (int)(((int64_t)iabs(coef) * mf + f) >> qbits);
    if (l > MAX_LEVEL) l = MAX_LEVEL;
    return coef < 0 ? -l : l;
}
/* 8.5.12.1 with flat scaling lists: d = (c * LevelScale4x4) scaled by qP/6; flat lists make
 * LevelScale = 16*v, and the <<(qP/6-4) / rounded >> forms both reduce to (c*v) << (qP/6). */
int orc_dequant4(int level, int qp, int pos) {
    return (level * k_dequant_v[qp % 6][pos_class(pos)]) << (qp / 6);
}

/* ================================================================== 8x8 transform (High profile) */
static int g_orc_t8 = 0; /* process-wide: transform_8x8_mode (High profile stream, 8x8 transform for P macroblocks) */
void orc_set_transform8x8(int on) { g_orc_t8 = on; }
int orc_get_transform8x8(void) { return g_orc_t8; }
/* Slices: g_slice_rows > 0 cuts the picture being coded into slices of that many macroblock rows (7.3.2.8: each its own NAL unit).  Inside the
 * encoder the availability of the row above changes (6.4.8: a macroblock of another slice is not available for intra prediction, for the
 * Intra_4x4 mode predictor, for nC, for QP_Y,PRED, for the 8.4.1.3 motion vector predictors and the 8.4.1.1 P_Skip inference), and the slices of
 * a picture become independent chains for the intra wavefront.  I pictures: since r03.  P pictures (r04; orc_enc_set_p_slices): the same cut,
 * vector prediction included -- what x264enc's threads do behind /root/reference/pipeline/generic/x264_superfast_camlink:5.
 * g_slice_dbf = disable_deblocking_filter_idc of the picture's slices: 0 the deblocking filter runs across slice boundaries, 2 (8.7:
 * filterTopMbEdgeFlag = 0 where the macroblock above belongs to another slice) it stops at them -- then the slices are independent chains for
 * the deblocking wavefront too, which is the point on the device (the picture period IS the deblocking launch). */
static int g_slice_rows = 0, g_slice_dbf = 0;
void orc_set_slice_rows(int rows) { g_slice_rows = rows > 0 ? rows : 0; }
int orc_get_slice_rows(void) { return g_slice_rows; }
void orc_set_slice_deblock(int idc) { g_slice_dbf = idc == 2 ? 2 : 0; }
int orc_get_slice_deblock(void) { return g_slice_dbf; }
int orc_auto_intra_slices(int mbh) { int n = mbh / 17; return n < 1 ? 1 : n > 8 ? 8 : n; } /* about 17 rows each (1080p: 4), at most 8 */
/* rows per slice for n slices of a picture of mbh rows (0: one slice); with slice-local deblocking a multiple of four rows, the height of the
 * device deblocker's bands (a band never spans two slices) */
int orc_slice_rows_for(int mbh, int n, int local_deblock) {
    if (n <= 1) return 0;
    int rows = (mbh + n - 1) / n;
    if (local_deblock) rows = (rows + 3) & ~3;
    return rows >= mbh ? 0 : rows;
}
static int top_ok(int my) { return my > 0 && !(g_slice_rows > 0 && my % g_slice_rows == 0); }
/* 8.5.6 8x8 zig-zag (frame) scan: scan position -> raster index y*8+x */
static const uint8_t k_zigzag8[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                      41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                      30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
/* 8.5.9 normAdjust8x8 v(m, class) */
static const uint8_t k_dequant8_v[6][6] = {{20, 18, 32, 19, 25, 24}, {22, 19, 35, 21, 28, 26}, {26, 23, 42, 24, 33, 31},
                                           {28, 25, 45, 26, 35, 33}, {32, 28, 51, 30, 40, 38}, {36, 32, 58, 34, 46, 43}};
/* encoder multipliers: round(2^24 / (v * g(class))), g = squared norms of the 8x8 basis {64, 81.5625, 25, 72.25, 40, 45.15625} */
static const uint16_t k_quant8_mf[6][6] = {{13107, 11428, 20972, 12222, 16777, 15481}, {11916, 10826, 19174, 11058, 14980, 14290},
                                           {10082, 8943, 15978, 9675, 12710, 11985},   {9362, 8228, 14913, 8931, 11984, 11259},
                                           {8192, 7346, 13159, 7740, 10486, 9777},     {7282, 6428, 11570, 6830, 9118, 8640}};
static inline int pos_class8(int pos) { /* pos = y*8+x (symmetric in x,y) */
    int i = pos >> 3, j = pos & 7;
    if (i % 4 == 0 && j % 4 == 0) return 0;
    if ((i & 1) && (j & 1)) return 1;
    if (i % 4 == 2 && j % 4 == 2) return 2;
    if ((i % 4 == 0 && (j & 1)) || ((i & 1) && j % 4 == 0)) return 3;
    if ((i % 4 == 0 && j % 4 == 2) || (i % 4 == 2 && j % 4 == 0)) return 4;
    return 5;
}
static void fdct8_1d(const int *s, int st, int *d, int dt) { /* encoder-side 8-point transform matching 8.5.13's inverse */
    int s07 = s[0] + s[7 * st], s16 = s[st] + s[6 * st], s25 = s[2 * st] + s[5 * st], s34 = s[3 * st] + s[4 * st];
    int a0 = s07 + s34, a1 = s16 + s25, a2 = s07 - s34, a3 = s16 - s25;
    int d07 = s[0] - s[7 * st], d16 = s[st] - s[6 * st], d25 = s[2 * st] - s[5 * st], d34 = s[3 * st] - s[4 * st];
    int a4 = d16 + d25 + (d07 + (d07 >> 1)), a5 = d07 - d34 - (d25 + (d25 >> 1));
    int a6 = d07 + d34 - (d16 + (d16 >> 1)), a7 = d16 - d25 + (d34 + (d34 >> 1));
    d[0] = a0 + a1; d[dt] = a4 + (a7 >> 2); d[2 * dt] = a2 + (a3 >> 1); d[3 * dt] = a5 + (a6 >> 2);
    d[4 * dt] = a0 - a1; d[5 * dt] = a6 - (a5 >> 2); d[6 * dt] = (a2 >> 1) - a3; d[7 * dt] = (a4 >> 2) - a7;
}
static void idct8_1d(const int *s, int st, int *d, int dt) { /* 8.5.13 one-dimensional inverse */
    int a0 = s[0] + s[4 * st], a2 = s[0] - s[4 * st], a4 = (s[2 * st] >> 1) - s[6 * st], a6 = (s[6 * st] >> 1) + s[2 * st];
    int b0 = a0 + a6, b2 = a2 + a4, b4 = a2 - a4, b6 = a0 - a6;
    int a1 = -s[3 * st] + s[5 * st] - s[7 * st] - (s[7 * st] >> 1), a3 = s[st] + s[7 * st] - s[3 * st] - (s[3 * st] >> 1);
    int a5 = -s[st] + s[7 * st] + s[5 * st] + (s[5 * st] >> 1), a7 = s[3 * st] + s[5 * st] + s[st] + (s[st] >> 1);
    int b1 = (a7 >> 2) + a1, b3 = a3 + (a5 >> 2), b5 = (a3 >> 2) - a5, b7 = a7 - (a1 >> 2);
    d[0] = b0 + b7; d[dt] = b2 + b5; d[2 * dt] = b4 + b3; d[3 * dt] = b6 + b1;
    d[4 * dt] = b6 - b1; d[5 * dt] = b4 - b3; d[6 * dt] = b2 - b5; d[7 * dt] = b0 - b7;
}
void orc_fdct8(const int in[64], int out[64]) { /* rows, then columns */
    int t[64];
    for (int i = 0; i < 8; i++) fdct8_1d(in + 8 * i, 1, t + 8 * i, 1);
    for (int j = 0; j < 8; j++) fdct8_1d(t + j, 8, out + j, 8);
}
void orc_idct8(const int in[64], int out[64]) { /* 8.5.13: each row, then each column; caller rounds with (x + 32) >> 6 */
    int t[64];
    for (int i = 0; i < 8; i++) idct8_1d(in + 8 * i, 1, t + 8 * i, 1);
    for (int j = 0; j < 8; j++) idct8_1d(t + j, 8, out + j, 8);
}
int orc_quant8(int coef, int qp, int pos, int intra) {
    int qbits = 16 + qp / 6, f = (1 << qbits) / (intra ? 3 : 6);
    int l = (int)(((int64_t)iabs(coef) * k_quant8_mf[qp % 6][pos_class8(pos)] + f) >> qbits);
    if (l > MAX_LEVEL) l = MAX_LEVEL;
    return coef < 0 ? -l : l;
}
int orc_dequant8(int level, int qp, int pos) { /* 8.5.13 scaling with flat lists: LevelScale8x8 = 16 * normAdjust8x8 */
    int ls = 16 * k_dequant8_v[qp % 6][pos_class8(pos)];
    return qp >= 36 ? (level * ls) << (qp / 6 - 6) : (level * ls + (1 << (5 - qp / 6))) >> (6 - qp / 6);
}
int orc_zigzag8(int k) { return k_zigzag8[k]; }
/* One 8x8 luma block of an inter macroblock: transform, quantise, reconstruct in place over the prediction.
 * Levels are stored de-interleaved the way CAVLC transmits them (7.3.5.3.2): 4x4 "block" 4*i8+j holds
 * scan positions 4k+j, k = 0..15.  Returns the 4-bit mask of non-zero sub-blocks. */
static int tq8_block_i(const uint8_t *src, uint8_t *rec, int stride, int qp, int intra, int16_t *lev4 /* 4 x 16 */) {
    int res[64], co[64], dq[64], out[64], mask = 0;
    for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) res[y * 8 + x] = src[(size_t)y * stride + x] - rec[(size_t)y * stride + x];
    orc_fdct8(res, co);
    for (int k = 0; k < 64; k++) {
        int pos = k_zigzag8[k], l = orc_quant8(co[pos], qp, pos, intra);
        lev4[(k & 3) * 16 + (k >> 2)] = (int16_t)l;
        if (l) mask |= 1 << (k & 3);
        dq[pos] = orc_dequant8(l, qp, pos);
    }
    if (mask) {
        orc_idct8(dq, out);
        for (int y = 0; y < 8; y++)
            for (int x = 0; x < 8; x++) rec[(size_t)y * stride + x] = (uint8_t)clip1(rec[(size_t)y * stride + x] + ((out[y * 8 + x] + 32) >> 6));
    }
    return mask;
}

static int tq8_block(const uint8_t *src, uint8_t *rec, int stride, int qp, int16_t *lev4) { return tq8_block_i(src, rec, stride, qp, 0, lev4); }

/* residual of one 4x4 block -> levels in zig-zag order; returns 1 if any level (from `first`) != 0 */
static int tq_block(const int16_t res[16], int qp, int intra, int first, int16_t lev_zz[16], int16_t *dc_out) {
    int16_t co[16];
    orc_fdct4(res, co);
    if (dc_out) *dc_out = co[0];
    int nz = 0;
    for (int k = 0; k < 16; k++) {
        if (k < first) { lev_zz[k] = 0; continue; }
        int pos = k_zigzag4[k];
        int l = orc_quant4(co[pos], qp, pos, intra);
        lev_zz[k] = (int16_t)l;
        nz |= l != 0;
    }
    return nz;
}
/* levels (zig-zag) -> scaled coefficients d[] in raster order; `dc` overrides d[0] when have_dc */
static void dq_block(const int16_t lev_zz[16], int qp, int first, int have_dc, int dc, int32_t d[16]) {
    for (int k = 0; k < 16; k++) {
        int pos = k_zigzag4[k];
        d[pos] = k < first ? 0 : orc_dequant4(lev_zz[k], qp, pos);
    }
    if (have_dc) d[0] = dc;
}

/* ================================================================== motion search */
static inline int se_bits(int v) { return orc_ue_bits(v > 0 ? (uint32_t)(2 * v - 1) : (uint32_t)(-2 * v), NULL); } /* bits of se(v), 9.1 */
static inline int ref_at(const uint8_t *p, int stride, int W, int H, int x, int y) { /* 8.4.2.2: the reference picture extended by coordinate clamping */
    return p[(size_t)CLIP3(0, H - 1, y) * stride + CLIP3(0, W - 1, x)];
}
/* Encoder choice (the standard does not constrain motion search).
 *
 * orc_me_frame: the SAD of EVERY candidate (dx,dy) in [-range,range]^2 of every macroblock -- vectors may leave the picture
 * (the SPS says motion_vectors_over_pic_boundaries_flag = 1), the reference being extended by coordinate clamping as 8.4.2.2
 * prescribes -- kept as a surface of 33 x 33 uint16 per macroblock (index (dy+16)*33 + dx+16; 0xFFFF outside the range), and a
 * first selection with the bits charged against the zero vector.
 *
 * orc_me_select: one more selection over the same surfaces, the bits charged against the 8.4.1.3 median of the vectors the
 * NEIGHBOURS chose in the previous selection (Jacobi iteration: every macroblock at once).  cost = SAD + lambda(qp) *
 * (bits(se(4(dx - px))) + bits(se(4(dy - py))) + ORC_SEL_BONUS), px,py whole samples -- except for the one candidate that equals the
 * 8.4.1.1 P_Skip inference on the same field, which is charged nothing (as P_Skip it would cost no macroblock header at all);
 * winner = lowest cost, then lowest dy, then lowest dx.
 * Where the scene really moves, textured macroblocks find the motion in the first selection and their flat neighbours follow
 * in the next ones (the surface is flat there, the bits decide); where nothing moves everything stays at zero.  After a few
 * iterations the field is what a sequential encoder's predictor-relative search would settle on, without its raster-order
 * dependency -- and the predictor estimates of the fused macroblock stage are taken from that field.
 * Output per macroblock: whole-sample vector in quarter-sample units, its SAD and the vector bits it was charged. */

/* ================================================================== adaptive quantisation (non-normative; device: aq_kernel, qp_chain_kernel)
 * One QP offset per macroblock from the luma variance of the SOURCE macroblock: flat areas, where quantisation noise and blocking show
 * first, get a finer quantiser, busy texture a coarser one (the idea of x264's aq-mode 1, in integers).  With s = sum and s2 = sum of squares
 * of the 256 samples, v = s2 - (s * s >> 8) is 256 x the variance; L2 = floor(2 log2 v) in half-octave steps (2 msb + next bit); the offset
 * is (3 (L2 - 28) + 4) >> 3 clamped to -4 .. +4: 0 at a standard deviation of 8, -3 on flat ground, +3 on noise.  Rate control is not told:
 * it sees the bytes.  Motion search and mode decisions keep the picture's lambda; only quantisation and the macroblock's QP_Y change.
 * A macroblock that sends no mb_qp_delta (P_Skip, no coded block and not Intra_16x16: 7.3.5) has the QP_Y of the macroblock before it in
 * decoding order (7.4.5: QP_Y,PRED) -- which is what the deblocking filter then reads for it: orc_qp_chain rewrites the records' qp
 * accordingly once a picture's records are final. */
static const int8_t *g_aq; /* offsets of the picture being coded; NULL: one QP per picture */
void orc_set_aq_map(const int8_t *off) { g_aq = off; }
static int mb_qp(int qp, int mbn) {
    if (!g_aq) return qp;
    const int q = qp + g_aq[mbn];
    return CLIP3(0, 51, q);
}
int orc_aq_offset_of(uint32_t s, uint32_t s2) {
    const uint32_t v = s2 - ((s * s) >> 8); /* s <= 65280: s * s fits 32 bits */
    int L2 = 0;
    if (v > 1) { int msb = 31; while (!(v >> msb)) msb--; L2 = 2 * msb + (int)((v >> (msb - 1)) & 1u); }
    const int o = (3 * (L2 - 28) + 4) >> 3;
    return CLIP3(-4, 4, o);
}
void orc_aq_offsets(const uint8_t *src_y, int stride, int mbw, int mbh, int8_t *off) {
    for (int my = 0; my < mbh; my++)
        for (int mx = 0; mx < mbw; mx++) {
            uint32_t s = 0, s2 = 0;
            for (int y = 0; y < 16; y++)
                for (int x = 0; x < 16; x++) { const uint32_t p = src_y[(size_t)(my * 16 + y) * stride + mx * 16 + x]; s += p; s2 += p * p; }
            off[my * mbw + mx] = (int8_t)orc_aq_offset_of(s, s2);
        }
}
void orc_qp_chain(orc_mbinfo_t *mbi, int nmb, int slice_qp) { orc_qp_chain_slices(mbi, nmb, slice_qp, 0); }
void orc_qp_chain_slices(orc_mbinfo_t *mbi, int nmb, int slice_qp, int slice_mbs) { /* slice_mbs > 0: a new slice (QP_Y,PRED = the slice's QP) every so many macroblocks */
    int prev = slice_qp;
    for (int i = 0; i < nmb; i++) {
        if (slice_mbs > 0 && i % slice_mbs == 0) prev = slice_qp;
        const int coded = mbi[i].mb_type == 0 || (mbi[i].nzmask & 0x07FFFFFFu) != 0;
        if (coded) prev = mbi[i].qp; else mbi[i].qp = (uint8_t)prev;
    }
}

int g_sel_bonus = ORC_SEL_BONUS; /* dev hook: orc_set_tuning(4, v) */
static void select_mb(const uint16_t *sf, int range, int lambda, int px, int py, int sx, int sy, orc_imv_t *o) {
    uint32_t best_cost = 0xFFFFFFFFu;
    int bdx = 0, bdy = 0, bsad = 0, bbits = 0;
    for (int dy = -range; dy <= range; dy++)
        for (int dx = -range; dx <= range; dx++) {
            const uint32_t sad = sf[(dy + 16) * 33 + dx + 16];
            const int bits = (dx == sx && dy == sy) ? 0 : se_bits(4 * (dx - px)) + se_bits(4 * (dy - py)) + g_sel_bonus;
            const uint32_t cost = sad + (uint32_t)(lambda * bits);
            if (cost < best_cost) { best_cost = cost; bdx = dx; bdy = dy; bsad = (int)sad; bbits = bits; } /* scan order dy then dx resolves ties */
        }
    o->mvx = (int16_t)(4 * bdx); o->mvy = (int16_t)(4 * bdy); o->sad = (uint16_t)bsad; o->bits = (uint16_t)bbits;
}
void orc_me_frame(const uint8_t *cur_y, const uint8_t *ref_y, int stride, int mbw, int mbh,
                  int range, int qp, uint16_t *surf, orc_imv_t *imv, int threads) {
    const int W = mbw * 16, H = mbh * 16;
    const int lambda = orc_me_lambda(qp);
    (void)threads;
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads > 0 ? threads : 1)
    for (int my = 0; my < mbh; my++) {
        for (int mx = 0; mx < mbw; mx++) {
            int x0 = mx * 16, y0 = my * 16;
            uint16_t *sf = surf + (size_t)(my * mbw + mx) * ORC_SURF;
            const uint8_t *c = cur_y + (size_t)y0 * stride + x0;
            for (int i = 0; i < ORC_SURF; i++) sf[i] = 0xFFFF;
            for (int dy = -range; dy <= range; dy++) {
                for (int dx = -range; dx <= range; dx++) {
                    uint32_t sad = 0;
                    if (x0 + dx >= 0 && y0 + dy >= 0 && x0 + dx + 16 <= W && y0 + dy + 16 <= H) {
                        const uint8_t *r = ref_y + (size_t)(y0 + dy) * stride + x0 + dx;
                        for (int y = 0; y < 16; y++)
                            for (int x = 0; x < 16; x++) sad += (uint32_t)iabs(c[y * stride + x] - r[y * stride + x]);
                    } else
                        for (int y = 0; y < 16; y++)
                            for (int x = 0; x < 16; x++) sad += (uint32_t)iabs(c[y * stride + x] - ref_at(ref_y, stride, W, H, x0 + dx + x, y0 + dy + y));
                    sf[(dy + 16) * 33 + dx + 16] = (uint16_t)sad; /* <= 65280 */
                }
            }
            select_mb(sf, range, lambda, 0, 0, 0, 0, &imv[my * mbw + mx]);
        }
    }
}
static int median3(int a, int b, int c) {
    int mn = a < b ? a : b, mx = a < b ? b : a;
    return c < mn ? mn : (c > mx ? mx : c);
}
/* 8.4.1.3 on a whole-sample field (every neighbour taken as inter, refIdx 0) and the 8.4.1.1 skip inference on the same field */
static void field_pred(const orc_imv_t *f, int mbw, int mx, int my, int *px, int *py, int *sx, int *sy) {
    const int top = top_ok(my); /* (the row above belongs to another slice: not available, 6.4.8) */
    const int avA = mx > 0, avB = top, avC = top && mx + 1 < mbw, avD = mx > 0 && top;
    const orc_imv_t *A = avA ? &f[my * mbw + mx - 1] : NULL, *B = avB ? &f[(my - 1) * mbw + mx] : NULL;
    const orc_imv_t *C = avC ? &f[(my - 1) * mbw + mx + 1] : (avD ? &f[(my - 1) * mbw + mx - 1] : NULL);
    const int n = (A != NULL) + (B != NULL) + (C != NULL);
    if (n == 1) { const orc_imv_t *o = A ? A : B ? B : C; *px = o->mvx; *py = o->mvy; }
    else { *px = median3(A ? A->mvx : 0, B ? B->mvx : 0, C ? C->mvx : 0); *py = median3(A ? A->mvy : 0, B ? B->mvy : 0, C ? C->mvy : 0); }
    if (sx) {
        *sx = *px; *sy = *py;
        if (!avA || !avB || (A->mvx == 0 && A->mvy == 0) || (B->mvx == 0 && B->mvy == 0)) { *sx = 0; *sy = 0; }
    }
}
void orc_me_select(const uint16_t *surf, int mbw, int mbh, int range, int qp, const orc_imv_t *in, orc_imv_t *out, int threads) {
    const int lambda = orc_me_lambda(qp);
    (void)threads;
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads > 0 ? threads : 1)
    for (int my = 0; my < mbh; my++)
        for (int mx = 0; mx < mbw; mx++) {
            int px, py, sx, sy;
            field_pred(in, mbw, mx, my, &px, &py, &sx, &sy);
            select_mb(surf + (size_t)(my * mbw + mx) * ORC_SURF, range, lambda, px >> 2, py >> 2, sx >> 2, sy >> 2, &out[my * mbw + mx]);
        }
}

/* ================================================================== chroma helpers */
/* NV12 interleaved chroma: plane c (0 = Cb, 1 = Cr) sample (x,y) lives at uv[y*stride + 2x + c] */
#define UV(p, stride, x, y, c) ((p)[(size_t)(y) * (stride) + 2 * (x) + (c)])

/* shared by intra and inter: transform/quant/reconstruct the two 8x8 chroma blocks of one
 * macroblock, given the prediction already written into rec_uv.  8.5.11 (chroma DC 2x2),
 * 8.5.12; Table 8-15 for QPc. */
static void chroma_tq_recon(const uint8_t *src_uv, uint8_t *rec_uv, int stride, int cx0, int cy0,
                            int qp, int intra, int16_t *lev, uint32_t *nzmask, int ac_drop) {
    int qpc = k_chroma_qp[CLIP3(0, 51, qp)];
    for (int c = 0; c < 2; c++) {
        int16_t dc[4];
        int16_t *ldc = lev + ORC_L_CDC + 4 * c;
        for (int b = 0; b < 4; b++) {
            int bx = (b & 1) * 4, by = (b >> 1) * 4;
            int16_t res[16];
            for (int y = 0; y < 4; y++)
                for (int x = 0; x < 4; x++)
                    res[y * 4 + x] = (int16_t)(UV(src_uv, stride, cx0 + bx + x, cy0 + by + y, c) -
                                               UV(rec_uv, stride, cx0 + bx + x, cy0 + by + y, c));
            int16_t *l = lev + ORC_L_CAC + (4 * c + b) * 16;
            if (tq_block(res, qpc, intra, 1, l, &dc[b])) *nzmask |= 1u << (16 + 4 * c + b);
        }
        /* forward 2x2 Hadamard of the four DCs (raster order), then quantise */
        int f0 = dc[0] + dc[1] + dc[2] + dc[3];
        int f1 = dc[0] - dc[1] + dc[2] - dc[3];
        int f2 = dc[0] + dc[1] - dc[2] - dc[3];
        int f3 = dc[0] - dc[1] - dc[2] + dc[3];
        ldc[0] = (int16_t)quant_dc(f0, qpc, intra);
        ldc[1] = (int16_t)quant_dc(f1, qpc, intra);
        ldc[2] = (int16_t)quant_dc(f2, qpc, intra);
        ldc[3] = (int16_t)quant_dc(f3, qpc, intra);
        if (ldc[0] | ldc[1] | ldc[2] | ldc[3]) *nzmask |= (c ? ORC_NZ_CRDC : ORC_NZ_CBDC);
    }
    if (ac_drop) { /* rate control's ladder for I pictures: chroma levels (both planes, DC and AC) whose magnitudes sum to no more than the threshold are not sent */
        int sum = 0;
        for (int i = 0; i < 128; i++) sum += iabs(lev[ORC_L_CAC + i]);
        for (int i = 0; i < 8; i++) sum += iabs(lev[ORC_L_CDC + i]);
        if (sum <= ac_drop) {
            memset(lev + ORC_L_CAC, 0, 128 * sizeof(int16_t)); memset(lev + ORC_L_CDC, 0, 8 * sizeof(int16_t));
            *nzmask &= ~(0x00FF0000u | ORC_NZ_CBDC | ORC_NZ_CRDC);
        }
    }
    /* 7.3.5: chroma AC levels are only transmitted when coded_block_pattern chroma == 2,
     * which this encoder sets when any AC level of either plane is non-zero. */
    for (int c = 0; c < 2; c++) {
        const int16_t *ldc = lev + ORC_L_CDC + 4 * c;
        /* 8.5.11.1/2: c = [[l0,l1],[l2,l3]], f = H c H, dcC = ((f*LevelScale(0,0)) << (qP/6)) >> 5 */
        int g0 = ldc[0] + ldc[1] + ldc[2] + ldc[3];
        int g1 = ldc[0] - ldc[1] + ldc[2] - ldc[3];
        int g2 = ldc[0] + ldc[1] - ldc[2] - ldc[3];
        int g3 = ldc[0] - ldc[1] - ldc[2] + ldc[3];
        int ls = 16 * k_dequant_v[qpc % 6][0];
        int dcv[4] = {((g0 * ls) << (qpc / 6)) >> 5, ((g1 * ls) << (qpc / 6)) >> 5,
                      ((g2 * ls) << (qpc / 6)) >> 5, ((g3 * ls) << (qpc / 6)) >> 5};
        for (int b = 0; b < 4; b++) {
            int bx = (b & 1) * 4, by = (b >> 1) * 4;
            int32_t d[16];
            dq_block(lev + ORC_L_CAC + (4 * c + b) * 16, qpc, 1, 1, dcv[b], d);
            idct4_add_step(d, &UV(rec_uv, stride, cx0 + bx, cy0 + by, c), stride, 2);
        }
    }
}

/* ================================================================== inter (P) picture */
/* 8.4.2.2.1 luma sample interpolation (6-tap half samples, averaged quarter samples) with the
 * reference picture extended by coordinate clamping, written from Figure 8-4 / Table 8-12. */
static inline int tap6(int a, int b, int c, int d, int e, int f) { return a - 5 * b + 20 * c + 20 * d - 5 * e + f; }
static int half_h(const uint8_t *p, int stride, int W, int H, int x, int y) { /* b1 at the right of (x,y) */
    return tap6(ref_at(p, stride, W, H, x - 2, y), ref_at(p, stride, W, H, x - 1, y), ref_at(p, stride, W, H, x, y),
                ref_at(p, stride, W, H, x + 1, y), ref_at(p, stride, W, H, x + 2, y), ref_at(p, stride, W, H, x + 3, y));
}
static int half_v(const uint8_t *p, int stride, int W, int H, int x, int y) { /* h1 below (x,y) */
    return tap6(ref_at(p, stride, W, H, x, y - 2), ref_at(p, stride, W, H, x, y - 1), ref_at(p, stride, W, H, x, y),
                ref_at(p, stride, W, H, x, y + 1), ref_at(p, stride, W, H, x, y + 2), ref_at(p, stride, W, H, x, y + 3));
}
static int luma_qpel(const uint8_t *p, int stride, int W, int H, int x, int y, int fx, int fy) {
    int G = ref_at(p, stride, W, H, x, y);
    if (!fx && !fy) return G;
    int b = clip1((half_h(p, stride, W, H, x, y) + 16) >> 5), h = clip1((half_v(p, stride, W, H, x, y) + 16) >> 5);
    if (!fy) return fx == 2 ? b : fx == 1 ? (G + b + 1) >> 1 : (ref_at(p, stride, W, H, x + 1, y) + b + 1) >> 1;
    if (!fx) return fy == 2 ? h : fy == 1 ? (G + h + 1) >> 1 : (ref_at(p, stride, W, H, x, y + 1) + h + 1) >> 1;
    int m = clip1((half_v(p, stride, W, H, x + 1, y) + 16) >> 5), s = clip1((half_h(p, stride, W, H, x, y + 1) + 16) >> 5);
    if ((fx & 1) && (fy & 1)) return ((fy == 1 ? b : s) + (fx == 1 ? h : m) + 1) >> 1; /* e, g, p, r */
    int j = clip1((tap6(half_h(p, stride, W, H, x, y - 2), half_h(p, stride, W, H, x, y - 1), half_h(p, stride, W, H, x, y),
                        half_h(p, stride, W, H, x, y + 1), half_h(p, stride, W, H, x, y + 2), half_h(p, stride, W, H, x, y + 3)) + 512) >> 10);
    if (fx == 2 && fy == 2) return j;
    if (fx == 2) return ((fy == 1 ? b : s) + j + 1) >> 1;  /* f, q */
    return ((fx == 1 ? h : m) + j + 1) >> 1;               /* i, k */
}
/* prediction of one macroblock for the quarter-sample vector (mvx, mvy); 8.4.1.4: the chroma
 * vector equals the luma vector, read in units of 1/8 chroma sample (8.4.2.2.2 bilinear). */
static void mc_mb(const uint8_t *ref_y, const uint8_t *ref_uv, uint8_t *rec_y, uint8_t *rec_uv,
                  int stride, int W, int H, int x0, int y0, int mvx, int mvy) {
    for (int y = 0; y < 16; y++)
        for (int x = 0; x < 16; x++)
            rec_y[(size_t)(y0 + y) * stride + x0 + x] =
                (uint8_t)luma_qpel(ref_y, stride, W, H, x0 + x + (mvx >> 2), y0 + y + (mvy >> 2), mvx & 3, mvy & 3);
    int cw = W / 2, ch = H / 2, cx0 = x0 / 2, cy0 = y0 / 2;
    int xi = mvx >> 3, yi = mvy >> 3, xf = mvx & 7, yf = mvy & 7;
    for (int c = 0; c < 2; c++)
        for (int y = 0; y < 8; y++)
            for (int x = 0; x < 8; x++) {
                int ax = CLIP3(0, cw - 1, cx0 + x + xi), bx = CLIP3(0, cw - 1, cx0 + x + xi + 1);
                int ay = CLIP3(0, ch - 1, cy0 + y + yi), cy = CLIP3(0, ch - 1, cy0 + y + yi + 1);
                int A = UV(ref_uv, stride, ax, ay, c), B = UV(ref_uv, stride, bx, ay, c);
                int C = UV(ref_uv, stride, ax, cy, c), D = UV(ref_uv, stride, bx, cy, c);
                UV(rec_uv, stride, cx0 + x, cy0 + y, c) =
                    (uint8_t)(((8 - xf) * (8 - yf) * A + xf * (8 - yf) * B + (8 - xf) * yf * C + xf * yf * D + 32) >> 6);
            }
}

/* Encoder choice: sub-sample refinement of the integer vectors found by orc_me_frame.
 * Two rounds around the current best vector, step 2 (half sample) then step 1 (quarter sample):
 * the 8 neighbours are visited in (dy, dx) raster order and replace the best only when strictly
 * cheaper.  cost = SAD(source, interpolated reference) + lambda * (bits(se(mvx)) + bits(se(mvy))). */
static inline int mvq_bits(int q) { return orc_ue_bits(q > 0 ? (uint32_t)(2 * q - 1) : (uint32_t)(-2 * q), NULL); }
void orc_subpel_frame(const uint8_t *cur_y, const uint8_t *ref_y, int stride, int mbw, int mbh, int qp,
                      orc_mbinfo_t *mbi, int threads) {
    const int W = mbw * 16, H = mbh * 16, lambda = orc_me_lambda(qp);
    (void)threads;
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads > 0 ? threads : 1)
    for (int my = 0; my < mbh; my++)
        for (int mx = 0; mx < mbw; mx++) {
            orc_mbinfo_t *m = &mbi[my * mbw + mx];
            int x0 = mx * 16, y0 = my * 16, bx = m->mvx, by = m->mvy;
            uint32_t best = m->cost;
            for (int step = 2; step >= 1; step--) {
                int cx = bx, cy = by;
                for (int dy = -1; dy <= 1; dy++)
                    for (int dx = -1; dx <= 1; dx++) {
                        if (!dx && !dy) continue;
                        int qx = cx + dx * step, qy = cy + dy * step;
                        uint32_t sad = 0;
                        for (int y = 0; y < 16; y++)
                            for (int x = 0; x < 16; x++)
                                sad += (uint32_t)iabs(cur_y[(size_t)(y0 + y) * stride + x0 + x] -
                                                      luma_qpel(ref_y, stride, W, H, x0 + x + (qx >> 2), y0 + y + (qy >> 2), qx & 3, qy & 3));
                        uint32_t cost = sad + (uint32_t)(lambda * (mvq_bits(qx) + mvq_bits(qy)));
                        if (cost < best) { best = cost; bx = qx; by = qy; }
                    }
            }
            m->mvx = (int16_t)bx; m->mvy = (int16_t)by; m->cost = best;
        }
}

void orc_inter_frame(const uint8_t *src_y, const uint8_t *src_uv, const uint8_t *ref_y,
                     const uint8_t *ref_uv, uint8_t *rec_y, uint8_t *rec_uv, int stride,
                     int mbw, int mbh, int qp, orc_mbinfo_t *mbi, int16_t *levels) {
    const int W = mbw * 16, H = mbh * 16;
    for (int my = 0; my < mbh; my++)
        for (int mx = 0; mx < mbw; mx++) {
            orc_mbinfo_t *m = &mbi[my * mbw + mx];
            int16_t *lev = levels + (size_t)(my * mbw + mx) * ORC_LEVELS_PER_MB;
            memset(lev, 0, ORC_LEVELS_PER_MB * sizeof(int16_t));
            int x0 = mx * 16, y0 = my * 16;
            m->mb_type = 1; m->i16_mode = 0; m->chroma_mode = 0; m->qp = (uint8_t)qp; m->nzmask = 0;
            mc_mb(ref_y, ref_uv, rec_y, rec_uv, stride, W, H, x0, y0, m->mvx, m->mvy);
            if (g_orc_t8) { /* encoder choice: every P_L0_16x16 macroblock uses the 8x8 transform when the stream allows it */
                for (int i8 = 0; i8 < 4; i8++) {
                    size_t o = (size_t)(y0 + (i8 >> 1) * 8) * stride + x0 + (i8 & 1) * 8;
                    int mk = tq8_block(src_y + o, rec_y + o, stride, qp, lev + ORC_L_LUMA + i8 * 64);
                    m->nzmask |= (uint32_t)mk << (4 * i8);
                }
                if (m->nzmask & 0xFFFF) m->nzmask |= ORC_NZ_T8; /* transform_size_8x8_flag is only sent (and only matters) with luma cbp != 0 */
            } else {
            for (int b = 0; b < 16; b++) {
                int bx = x0 + k_blk_x[b], by = y0 + k_blk_y[b];
                int16_t res[16];
                for (int y = 0; y < 4; y++)
                    for (int x = 0; x < 4; x++)
                        res[y * 4 + x] = (int16_t)(src_y[(size_t)(by + y) * stride + bx + x] -
                                                   rec_y[(size_t)(by + y) * stride + bx + x]);
                if (tq_block(res, qp, 0, 0, lev + ORC_L_LUMA + b * 16, NULL)) m->nzmask |= 1u << b;
            }
            for (int b = 0; b < 16; b++) {
                if (!(m->nzmask & (1u << b))) continue;
                int32_t d[16];
                dq_block(lev + ORC_L_LUMA + b * 16, qp, 0, 0, 0, d);
                orc_idct4_add(d, rec_y + (size_t)(y0 + k_blk_y[b]) * stride + x0 + k_blk_x[b], stride);
            }
            }
            chroma_tq_recon(src_uv, rec_uv, stride, x0 / 2, y0 / 2, qp, 0, lev, &m->nzmask, 0);
        }
}

/* ================================================================== intra (I) picture */
/* 8.3.3 Intra_16x16 prediction, written into out[256] */
static void pred16(const uint8_t *rec_y, int stride, int x0, int y0, int mode, int has_top, int has_left,
                   uint8_t out[256]) {
    const uint8_t *p = rec_y + (size_t)y0 * stride + x0;
    if (mode == 0) { /* vertical 8.3.3.1 */
        for (int y = 0; y < 16; y++) for (int x = 0; x < 16; x++) out[y * 16 + x] = p[-stride + x];
    } else if (mode == 1) { /* horizontal 8.3.3.2 */
        for (int y = 0; y < 16; y++) for (int x = 0; x < 16; x++) out[y * 16 + x] = p[y * stride - 1];
    } else if (mode == 2) { /* DC 8.3.3.3 */
        int s = 0, v;
        if (has_top) for (int x = 0; x < 16; x++) s += p[-stride + x];
        if (has_left) for (int y = 0; y < 16; y++) s += p[y * stride - 1];
        if (has_top && has_left) v = (s + 16) >> 5;
        else if (has_top || has_left) v = (s + 8) >> 4;
        else v = 128;
        memset(out, v, 256);
    } else { /* plane 8.3.3.4 */
        int Hh = 0, Vv = 0;
        for (int i = 0; i < 8; i++) {
            Hh += (i + 1) * (p[-stride + 8 + i] - p[-stride + 6 - i]);
            Vv += (i + 1) * (p[(8 + i) * stride - 1] - p[(6 - i) * stride - 1]);
        }
        int a = 16 * (p[15 * stride - 1] + p[-stride + 15]);
        int b = (5 * Hh + 32) >> 6, c = (5 * Vv + 32) >> 6;
        for (int y = 0; y < 16; y++)
            for (int x = 0; x < 16; x++) out[y * 16 + x] = (uint8_t)clip1((a + b * (x - 7) + c * (y - 7) + 16) >> 5);
    }
}
/* 8.3.4 chroma prediction for one plane, out[64] */
static void pred_chroma(const uint8_t *rec_uv, int stride, int cx0, int cy0, int c, int mode, int has_top,
                        int has_left, uint8_t out[64]) {
#define T(x) UV(rec_uv, stride, cx0 + (x), cy0 - 1, c)
#define L(y) UV(rec_uv, stride, cx0 - 1, cy0 + (y), c)
    if (mode == 0) { /* DC 8.3.4.1-3, per 4x4 chroma block */
        for (int b = 0; b < 4; b++) {
            int bx = (b & 1) * 4, by = (b >> 1) * 4, v;
            int st = 0, sl = 0;
            if (has_top) for (int i = 0; i < 4; i++) st += T(bx + i);
            if (has_left) for (int i = 0; i < 4; i++) sl += L(by + i);
            if (b == 0 || b == 3) {
                if (has_top && has_left) v = (st + sl + 4) >> 3;
                else if (has_top) v = (st + 2) >> 2;
                else if (has_left) v = (sl + 2) >> 2;
                else v = 128;
            } else if (b == 1) {
                if (has_top) v = (st + 2) >> 2;
                else if (has_left) v = (sl + 2) >> 2;
                else v = 128;
            } else {
                if (has_left) v = (sl + 2) >> 2;
                else if (has_top) v = (st + 2) >> 2;
                else v = 128;
            }
            for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) out[(by + y) * 8 + bx + x] = (uint8_t)v;
        }
    } else if (mode == 1) { /* horizontal */
        for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) out[y * 8 + x] = L(y);
    } else if (mode == 2) { /* vertical */
        for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) out[y * 8 + x] = T(x);
    } else { /* plane 8.3.4.4, xCF = yCF = 0 */
        int Hh = 0, Vv = 0;
        for (int i = 0; i < 4; i++) {
            Hh += (i + 1) * (T(4 + i) - T(2 - i));
            Vv += (i + 1) * (L(4 + i) - L(2 - i));
        }
        int a = 16 * (L(7) + T(7));
        int b = (34 * Hh + 32) >> 6, cc = (34 * Vv + 32) >> 6;
        for (int y = 0; y < 8; y++)
            for (int x = 0; x < 8; x++) out[y * 8 + x] = (uint8_t)clip1((a + b * (x - 3) + cc * (y - 3) + 16) >> 5);
    }
#undef T
#undef L
}

static int g_orc_i4x4 = 1; /* encoder switch: try Intra_4x4 in I pictures */
void orc_set_i4x4(int on) { g_orc_i4x4 = on; }
/* 8.3.1.2 Intra_4x4 sample prediction.  e[0..12]: e[0] = p[-1,-1], e[1..8] = p[0..7,-1], e[9..12] = p[-1,0..3]. */
static void pred4x4(const int e[13], int mode, int has_up, int has_left, uint8_t out[16]) {
    const int *t = e + 1, *l = e + 9, c = e[0];
    for (int y = 0; y < 4; y++)
        for (int x = 0; x < 4; x++) {
            int v;
            switch (mode) {
            case 0: v = t[x]; break;
            case 1: v = l[y]; break;
            case 2:
                if (has_up && has_left) v = (t[0] + t[1] + t[2] + t[3] + l[0] + l[1] + l[2] + l[3] + 4) >> 3;
                else if (has_left) v = (l[0] + l[1] + l[2] + l[3] + 2) >> 2;
                else if (has_up) v = (t[0] + t[1] + t[2] + t[3] + 2) >> 2;
                else v = 128;
                break;
            case 3: v = (x == 3 && y == 3) ? (t[6] + 3 * t[7] + 2) >> 2 : (t[x + y] + 2 * t[x + y + 1] + t[x + y + 2] + 2) >> 2; break;
#define EDGE(i) ((i) < 0 ? l[-(i) - 1] : (i) == 0 ? c : t[(i) - 1]) /* ... l1 l0 c t0 t1 ... */
            case 4: v = (EDGE(x - y - 1) + 2 * EDGE(x - y) + EDGE(x - y + 1) + 2) >> 2; break;
            case 5: {
                int z = 2 * x - y, k = x - (y >> 1);
                if (z >= 0 && !(z & 1)) v = (EDGE(k) + EDGE(k + 1) + 1) >> 1;
                else if (z >= 0) v = (EDGE(k - 1) + 2 * EDGE(k) + EDGE(k + 1) + 2) >> 2;
                else if (z == -1) v = (l[0] + 2 * c + t[0] + 2) >> 2;
                else v = (EDGE(-y) + 2 * EDGE(-y + 1) + EDGE(-y + 2) + 2) >> 2;
                break; }
            case 6: {
                int z = 2 * y - x, k = y - (x >> 1);
                if (z >= 0 && !(z & 1)) v = (EDGE(-k) + EDGE(-k - 1) + 1) >> 1;
                else if (z >= 0) v = (EDGE(-k + 1) + 2 * EDGE(-k) + EDGE(-k - 1) + 2) >> 2;
                else if (z == -1) v = (l[0] + 2 * c + t[0] + 2) >> 2;
                else v = (EDGE(x) + 2 * EDGE(x - 1) + EDGE(x - 2) + 2) >> 2;
                break; }
#undef EDGE
            case 7: v = !(y & 1) ? (t[x + (y >> 1)] + t[x + (y >> 1) + 1] + 1) >> 1
                                 : (t[x + (y >> 1)] + 2 * t[x + (y >> 1) + 1] + t[x + (y >> 1) + 2] + 2) >> 2; break;
            default: {
                int z = x + 2 * y, k = y + (x >> 1);
                if (z > 5) v = l[3];
                else if (z == 5) v = (l[2] + 3 * l[3] + 2) >> 2;
                else if (!(z & 1)) v = (l[k] + l[k + 1] + 1) >> 1;
                else v = (l[k] + 2 * l[k + 1] + l[k + 2] + 2) >> 2;
                break; }
            }
            out[y * 4 + x] = (uint8_t)v;
        }
}
/* ---- Intra analysis (encoder choice).  Mode decisions are "open loop": every candidate prediction is
 * built from the SOURCE picture's neighbouring samples, so all macroblocks (and all sixteen 4x4 blocks of a
 * macroblock) can be analysed independently -- the device does this in one fully parallel kernel -- while the
 * reconstruction that follows uses the normative reconstructed neighbours with the modes chosen here.
 * 0xFFFF marks a mode whose neighbours do not exist (or, for block 5, that would read the macroblock
 * above-right, which would break the device's x+y wavefront). */
static void e13_from(const uint8_t *p, int stride, int X, int Y, int up, int lf, int ul, int ur, int e[13]) {
    e[0] = ul ? p[(size_t)(Y - 1) * stride + X - 1] : 0;
    for (int i = 0; i < 4; i++) { e[1 + i] = up ? p[(size_t)(Y - 1) * stride + X + i] : 0; e[9 + i] = lf ? p[(size_t)(Y + i) * stride + X - 1] : 0; }
    for (int i = 4; i < 8; i++) e[1 + i] = ur ? p[(size_t)(Y - 1) * stride + X + i] : e[4];
}
static void blk4_avail(int b, int has_top, int has_left, int has_tr, int *up, int *lf, int *ul, int *ur) {
    static const uint8_t raster_blk[16] = {0, 1, 4, 5, 2, 3, 6, 7, 8, 9, 12, 13, 10, 11, 14, 15};
    const int bx = k_blk_x[b] >> 2, by = k_blk_y[b] >> 2;
    *up = by > 0 || has_top; *lf = bx > 0 || has_left;
    *ul = (bx > 0 && by > 0) ? 1 : bx > 0 ? has_top : by > 0 ? has_left : (has_top && has_left);
    *ur = by == 0 ? (bx < 3 ? has_top : has_tr) : (bx < 3 && raster_blk[(by - 1) * 4 + bx + 1] < b);
}
static int mode4_ok(int b, int mode, int up, int lf, int ul) {
    const int need_up = mode == 0 || mode == 3 || mode == 7, need_left = mode == 1 || mode == 8, need_all = mode >= 4 && mode <= 6;
    if ((need_up && !up) || (need_left && !lf) || (need_all && !(up && lf && ul))) return 0;
    return !(b == 5 && (mode == 3 || mode == 7));
}
void orc_intra_analyse(const uint8_t *src_y, const uint8_t *src_uv, int stride, int mbw, int mbh, orc_isad_t *out) {
    for (int my = 0; my < mbh; my++)
        for (int mx = 0; mx < mbw; mx++) {
            orc_isad_t *o = &out[my * mbw + mx];
            const int x0 = mx * 16, y0 = my * 16, has_top = top_ok(my), has_left = mx > 0, has_tr = top_ok(my) && mx + 1 < mbw;
            uint8_t pred[256], cp[64];
            for (int mode = 0; mode < 4; mode++) {
                o->i16[mode] = 0xFFFF;
                if ((mode == 0 && !has_top) || (mode == 1 && !has_left) || (mode == 3 && !(has_top && has_left))) continue;
                pred16(src_y, stride, x0, y0, mode, has_top, has_left, pred);
                uint32_t sad = 0;
                for (int y = 0; y < 16; y++) for (int x = 0; x < 16; x++) sad += (uint32_t)iabs(src_y[(size_t)(y0 + y) * stride + x0 + x] - pred[y * 16 + x]);
                o->i16[mode] = (uint16_t)sad;
            }
            for (int mode = 0; mode < 4; mode++) {
                o->chroma[mode] = 0xFFFF;
                if ((mode == 1 && !has_left) || (mode == 2 && !has_top) || (mode == 3 && !(has_top && has_left))) continue;
                uint32_t sad = 0;
                for (int c = 0; c < 2; c++) {
                    pred_chroma(src_uv, stride, x0 / 2, y0 / 2, c, mode, has_top, has_left, cp);
                    for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) sad += (uint32_t)iabs(UV(src_uv, stride, x0 / 2 + x, y0 / 2 + y, c) - cp[y * 8 + x]);
                }
                o->chroma[mode] = (uint16_t)sad;
            }
            for (int b = 0; b < 16; b++) {
                const int X = x0 + k_blk_x[b], Y = y0 + k_blk_y[b];
                int up, lf, ul, ur, e[13];
                blk4_avail(b, has_top, has_left, has_tr, &up, &lf, &ul, &ur);
                e13_from(src_y, stride, X, Y, up, lf, ul, ur, e);
                for (int mode = 0; mode < 9; mode++) {
                    o->i4[b][mode] = 0xFFFF;
                    if (!mode4_ok(b, mode, up, lf, ul)) continue;
                    uint8_t p4[16];
                    pred4x4(e, mode, up, lf, p4);
                    uint32_t sad = 0;
                    for (int i = 0; i < 16; i++) sad += (uint32_t)iabs(src_y[(size_t)(Y + (i >> 2)) * stride + X + (i & 3)] - p4[i]);
                    o->i4[b][mode] = (uint16_t)sad;
                }
            }
        }
}
static int argmin_u16(const uint16_t *v, int n, uint32_t *best) {
    int bi = 0; uint32_t b = 0xFFFFFFFFu;
    for (int i = 0; i < n; i++) if (v[i] != 0xFFFF && v[i] < b) { b = v[i]; bi = i; }
    *best = b;
    return bi;
}
/* Intra_4x4 modes of one macroblock from the analysed SADs: per block lowest SAD + lambda * (mode == expected ? 1 : 4),
 * ties to the lowest mode.  "expected" follows 8.3.1.1 inside the macroblock; blocks on the left / top border take DC (2)
 * for the neighbouring macroblock's block whatever that macroblock chose, so the decision needs nothing outside the
 * macroblock and all macroblocks can be decided at once (the entropy coder still codes against the true predicted mode).
 * Modes go to lev[ORC_L_LDC + blkIdx]. */
static uint32_t intra4x4_choose(const orc_isad_t *sad, int mx, int my, int lambda, int16_t *lev) {
    static const uint8_t raster_blk[16] = {0, 1, 4, 5, 2, 3, 6, 7, 8, 9, 12, 13, 10, 11, 14, 15};
    uint32_t total = 0;
    for (int b = 0; b < 16; b++) {
        const int bx = k_blk_x[b] >> 2, by = k_blk_y[b] >> 2;
        const int ma = bx > 0 ? lev[ORC_L_LDC + raster_blk[by * 4 + bx - 1]] : (mx > 0 ? 2 : -1);
        const int mb_ = by > 0 ? lev[ORC_L_LDC + raster_blk[(by - 1) * 4 + bx]] : (top_ok(my) ? 2 : -1);
        const int pm = (ma < 0 || mb_ < 0) ? 2 : (ma < mb_ ? ma : mb_);
        uint32_t best = 0xFFFFFFFFu; int best_mode = 2;
        for (int mode = 0; mode < 9; mode++) {
            if (sad->i4[b][mode] == 0xFFFF) continue;
            uint32_t cost = sad->i4[b][mode] + (uint32_t)(lambda * (mode == pm ? 1 : 4));
            if (cost < best) { best = cost; best_mode = mode; }
        }
        lev[ORC_L_LDC + b] = (int16_t)best_mode;
        total += best;
    }
    return total;
}
/* Every mode decision of an I picture from the analysed SADs (no reconstructed sample is involved): Intra_16x16 mode =
 * lowest SAD (ties to the lowest mode number), chroma likewise on Cb+Cr jointly, Intra_4x4 modes by intra4x4_choose;
 * I_NxN is taken when its cost + 32 lambda (its extra header bits) is strictly lower. */
void orc_intra_decide(const orc_isad_t *isad, int mbw, int mbh, int qp, int i4x4, orc_idec_t *out) {
    const int lambda = orc_me_lambda(qp);
    for (int my = 0; my < mbh; my++)
        for (int mx = 0; mx < mbw; mx++) {
            const orc_isad_t *sad = &isad[my * mbw + mx];
            orc_idec_t *d = &out[my * mbw + mx];
            uint32_t luma_sad, chroma_sad;
            int16_t lev[ORC_LEVELS_PER_MB];
            memset(d, 0, sizeof *d);
            d->mode16 = (uint8_t)argmin_u16(sad->i16, 4, &luma_sad);
            d->cmode = (uint8_t)argmin_u16(sad->chroma, 4, &chroma_sad);
            if (i4x4) {
                uint32_t cost4 = intra4x4_choose(sad, mx, my, lambda, lev);
                if (cost4 + (uint32_t)(32 * lambda) < luma_sad) { d->use_i4 = 1; luma_sad = cost4 + (uint32_t)(32 * lambda); }
                for (int b = 0; b < 16; b++) d->modes4[b] = (uint8_t)lev[ORC_L_LDC + b];
            }
            d->cost = luma_sad + chroma_sad;
            d->cost_luma = luma_sad;
        }
}
/* reconstruction of an Intra_4x4 macroblock with the modes in lev[ORC_L_LDC..] (8.3.1.2 + 8.5) */
static void intra4x4_recon(const uint8_t *src_y, uint8_t *rec_y, int stride, int mbw, int mx, int my, int qp, int16_t *lev, uint32_t *nzmask) {
    const int x0 = mx * 16, y0 = my * 16, has_top = top_ok(my), has_left = mx > 0, has_tr = top_ok(my) && mx + 1 < mbw;
    for (int b = 0; b < 16; b++) {
        const int X = x0 + k_blk_x[b], Y = y0 + k_blk_y[b];
        int up, lf, ul, ur, e[13];
        blk4_avail(b, has_top, has_left, has_tr, &up, &lf, &ul, &ur);
        e13_from(rec_y, stride, X, Y, up, lf, ul, ur, e);
        uint8_t p4[16];
        pred4x4(e, lev[ORC_L_LDC + b], up, lf, p4);
        int16_t res[16];
        for (int i = 0; i < 16; i++) res[i] = (int16_t)(src_y[(size_t)(Y + (i >> 2)) * stride + X + (i & 3)] - p4[i]);
        if (tq_block(res, qp, 1, 0, lev + ORC_L_LUMA + b * 16, NULL)) *nzmask |= 1u << b;
        for (int i = 0; i < 16; i++) rec_y[(size_t)(Y + (i >> 2)) * stride + X + (i & 3)] = p4[i];
        int32_t d[16];
        dq_block(lev + ORC_L_LUMA + b * 16, qp, 0, 0, 0, d);
        orc_idct4_add(d, rec_y + (size_t)Y * stride + X, stride);
    }
}

/* ================================================================== Intra_8x8 (High profile; I pictures of a stream with transform_8x8_mode)
 * 8.3.2: four 8x8 luma blocks per macroblock, nine modes each, predicted from reference samples that are low-pass filtered first (8.3.2.2.1).
 * Encoder side (r03): open-loop SADs on source neighbours like Intra_4x4, decision per block = SAD + lambda * (1 when the mode is the expected one,
 * else 4) with the expected mode taken inside the macroblock (neighbours outside count as DC), macroblock taken as Intra_8x8 when that total + 10 lambda
 * is strictly below what Intra_16x16 / Intra_4x4 left.  The record is mb_type 2 with ORC_NZ_T8 set (whatever its levels), the four modes in
 * lev[ORC_L_LDC + 0..3], the levels de-interleaved like the inter macroblocks' 8x8 blocks. */
static int g_orc_i8x8 = 0; /* process-wide (default off, like mi355enc_cfg_t.i8x8): try Intra_8x8 in the I pictures of a stream with the 8x8 transform */
#define ORC_I8_QP_MAX 37    /* ... at picture quantisers up to this one: above it the four CAVLC sub-blocks of an 8x8 block cost more than the prediction saves (measured) */
void orc_set_i8x8(int on) { g_orc_i8x8 = on; }
int orc_get_i8x8(void) { return g_orc_i8x8; }
/* availability of the neighbours of 8x8 block b (raster) of a macroblock with has_top / has_left / has_tr */
static void blk8_avail(int b, int has_top, int has_left, int has_tr, int *up, int *lf, int *ul, int *ur) {
    *up = b >= 2 || has_top; *lf = (b & 1) || has_left;
    *ul = b == 0 ? (has_top && has_left) : b == 1 ? has_top : b == 2 ? has_left : 1;
    *ur = b == 0 ? has_top : b == 1 ? has_tr : b == 2 ? 1 : 0;
}
/* the 25 reference samples of the 8x8 block at (X, Y) of plane p, filtered (8.3.2.2.1): top[0] = p'[-1,-1], top[1 + x] = p'[x,-1] (x = 0..15), left[y] = p'[-1,y] */
static void i8_refs(const uint8_t *p, int stride, int X, int Y, int up, int lf, int ul, int ur, int top[17], int left[8]) {
    int c = ul ? p[(size_t)(Y - 1) * stride + X - 1] : 0, t[16], l[8];
    for (int x = 0; x < 8; x++) t[x] = up ? p[(size_t)(Y - 1) * stride + X + x] : 0;
    for (int x = 8; x < 16; x++) t[x] = (up && ur) ? p[(size_t)(Y - 1) * stride + X + x] : t[7]; /* 8.3.2.2: not available -> p[7,-1] */
    for (int y = 0; y < 8; y++) l[y] = lf ? p[(size_t)(Y + y) * stride + X - 1] : 0;
    for (int i = 0; i < 17; i++) top[i] = 0;
    for (int i = 0; i < 8; i++) left[i] = 0;
    if (up) {
        top[1] = ul ? (c + 2 * t[0] + t[1] + 2) >> 2 : (3 * t[0] + t[1] + 2) >> 2;
        for (int x = 1; x < 15; x++) top[1 + x] = (t[x - 1] + 2 * t[x] + t[x + 1] + 2) >> 2;
        top[16] = (t[14] + 3 * t[15] + 2) >> 2;
    }
    if (lf) {
        left[0] = ul ? (c + 2 * l[0] + l[1] + 2) >> 2 : (3 * l[0] + l[1] + 2) >> 2;
        for (int y = 1; y < 7; y++) left[y] = (l[y - 1] + 2 * l[y] + l[y + 1] + 2) >> 2;
        left[7] = (l[6] + 3 * l[7] + 2) >> 2;
    }
    if (ul) top[0] = (up && lf) ? (t[0] + 2 * c + l[0] + 2) >> 2 : up ? (3 * c + t[0] + 2) >> 2 : lf ? (3 * c + l[0] + 2) >> 2 : c;
}
static int mode8_ok(int mode, int up, int lf, int ul) {
    const int need_up = mode == 0 || mode == 3 || mode == 7, need_left = mode == 1 || mode == 8, need_all = mode >= 4 && mode <= 6;
    return !((need_up && !up) || (need_left && !lf) || (need_all && !(up && lf && ul)));
}
/* 8.3.2.2.2 .. 8.3.2.2.10 on the filtered samples, as one edge array: E(0) = p'[-1,-1], E(k) = p'[k-1,-1] (k = 1..16), E(-k) = p'[-1,k-1] (k = 1..8) */
static void pred8x8(const int top[17], const int left[8], int mode, int up, int lf, uint8_t out[64]) {
    int e[25];
    for (int k = 0; k <= 16; k++) e[8 + k] = top[k];
    for (int k = 1; k <= 8; k++) e[8 - k] = left[k - 1];
#define E(i) e[(i) + 8]
    for (int y = 0; y < 8; y++)
        for (int x = 0; x < 8; x++) {
            int v = 128;
            switch (mode) {
            case 0: v = E(x + 1); break;
            case 1: v = E(-(y + 1)); break;
            case 2: {
                int s = 0;
                if (up) for (int i = 1; i <= 8; i++) s += E(i);
                if (lf) for (int i = 1; i <= 8; i++) s += E(-i);
                v = (up && lf) ? (s + 8) >> 4 : (up || lf) ? (s + 4) >> 3 : 128;
                break; }
            case 3: v = (x == 7 && y == 7) ? (E(15) + 3 * E(16) + 2) >> 2 : (E(x + y + 1) + 2 * E(x + y + 2) + E(x + y + 3) + 2) >> 2; break;
            case 4: { const int i = x - y; v = (E(i - 1) + 2 * E(i) + E(i + 1) + 2) >> 2; break; }
            case 5: {
                const int z = 2 * x - y, j = x - (y >> 1);
                if (z >= 0 && !(z & 1)) v = (E(j) + E(j + 1) + 1) >> 1;
                else if (z >= 0) v = (E(j - 1) + 2 * E(j) + E(j + 1) + 2) >> 2;
                else if (z == -1) v = (E(-1) + 2 * E(0) + E(1) + 2) >> 2;
                else { const int k = y - 2 * x - 1; v = (E(-(k + 1)) + 2 * E(-k) + E(-(k - 1)) + 2) >> 2; }
                break; }
            case 6: {
                const int z = 2 * y - x, j = y - (x >> 1);
                if (z >= 0 && !(z & 1)) v = (E(-j) + E(-(j + 1)) + 1) >> 1;
                else if (z >= 0) v = (E(-(j - 1)) + 2 * E(-j) + E(-(j + 1)) + 2) >> 2;
                else if (z == -1) v = (E(-1) + 2 * E(0) + E(1) + 2) >> 2;
                else { const int k = x - 2 * y - 1; v = (E(k + 1) + 2 * E(k) + E(k - 1) + 2) >> 2; }
                break; }
            case 7: { const int j = x + (y >> 1); v = !(y & 1) ? (E(j + 1) + E(j + 2) + 1) >> 1 : (E(j + 1) + 2 * E(j + 2) + E(j + 3) + 2) >> 2; break; }
            default: {
                const int z = x + 2 * y, j = y + (x >> 1);
                if (z > 13) v = E(-8);
                else if (z == 13) v = (E(-7) + 3 * E(-8) + 2) >> 2;
                else if (!(z & 1)) v = (E(-(j + 1)) + E(-(j + 2)) + 1) >> 1;
                else v = (E(-(j + 1)) + 2 * E(-(j + 2)) + E(-(j + 3)) + 2) >> 2;
                break; }
            }
            out[y * 8 + x] = (uint8_t)v;
        }
#undef E
}
/* open-loop analysis (source neighbours) and decision for one macroblock: returns cost8 = sum of the blocks' best SAD + lambda * mode bits; modes8[4] */
static uint32_t intra8x8_choose(const uint8_t *src_y, int stride, int mbw, int mx, int my, int lambda, uint8_t modes8[4]) {
    const int x0 = mx * 16, y0 = my * 16, has_top = top_ok(my), has_left = mx > 0, has_tr = top_ok(my) && mx + 1 < mbw;
    uint32_t total = 0;
    for (int b = 0; b < 4; b++) {
        const int X = x0 + (b & 1) * 8, Y = y0 + (b >> 1) * 8;
        int up, lf, ul, ur, top[17], left[8];
        blk8_avail(b, has_top, has_left, has_tr, &up, &lf, &ul, &ur);
        i8_refs(src_y, stride, X, Y, up, lf, ul, ur, top, left);
        const int ma = (b & 1) ? modes8[b - 1] : (has_left ? 2 : -1), mb_ = (b >> 1) ? modes8[b - 2] : (has_top ? 2 : -1);
        const int pm = (ma < 0 || mb_ < 0) ? 2 : (ma < mb_ ? ma : mb_);
        uint32_t best = 0xFFFFFFFFu; int best_mode = 2;
        for (int mode = 0; mode < 9; mode++) {
            if (!mode8_ok(mode, up, lf, ul)) continue;
            uint8_t p8[64];
            pred8x8(top, left, mode, up, lf, p8);
            uint32_t sad = 0;
            for (int i = 0; i < 64; i++) sad += (uint32_t)iabs(src_y[(size_t)(Y + (i >> 3)) * stride + X + (i & 7)] - p8[i]);
            const uint32_t cost = sad + (uint32_t)(lambda * (mode == pm ? 1 : 4));
            if (cost < best) { best = cost; best_mode = mode; }
        }
        modes8[b] = (uint8_t)best_mode;
        total += best;
    }
    return total;
}
/* after orc_intra_decide: Intra_8x8 where it is strictly cheaper (use_i4 = 2, the four modes in modes4[0..3]) */
void orc_intra_decide8(const uint8_t *src_y, int stride, int mbw, int mbh, int qp, orc_idec_t *idec) {
    const int lambda = orc_me_lambda(qp);
    for (int my = 0; my < mbh; my++)
        for (int mx = 0; mx < mbw; mx++) {
            orc_idec_t *d = &idec[my * mbw + mx];
            uint8_t m8[4];
            const uint32_t c8 = intra8x8_choose(src_y, stride, mbw, mx, my, lambda, m8) + (uint32_t)(10 * lambda);
            if (c8 < d->cost_luma) {
                d->cost += c8 - d->cost_luma; d->cost_luma = c8; d->use_i4 = 2;
                memset(d->modes4, 0, sizeof d->modes4);
                for (int b = 0; b < 4; b++) d->modes4[b] = m8[b];
            }
        }
}
/* reconstruction of an Intra_8x8 macroblock with the modes in lev[ORC_L_LDC + 0..3] (8.3.2 + 8.5.13) */
static void intra8x8_recon(const uint8_t *src_y, uint8_t *rec_y, int stride, int mbw, int mx, int my, int qp, int16_t *lev, uint32_t *nzmask) {
    const int x0 = mx * 16, y0 = my * 16, has_top = top_ok(my), has_left = mx > 0, has_tr = top_ok(my) && mx + 1 < mbw;
    for (int b = 0; b < 4; b++) {
        const int X = x0 + (b & 1) * 8, Y = y0 + (b >> 1) * 8;
        int up, lf, ul, ur, top[17], left[8];
        blk8_avail(b, has_top, has_left, has_tr, &up, &lf, &ul, &ur);
        i8_refs(rec_y, stride, X, Y, up, lf, ul, ur, top, left);
        uint8_t p8[64];
        pred8x8(top, left, lev[ORC_L_LDC + b], up, lf, p8);
        for (int y = 0; y < 8; y++) memcpy(rec_y + (size_t)(Y + y) * stride + X, p8 + y * 8, 8);
        *nzmask |= (uint32_t)tq8_block_i(src_y + (size_t)Y * stride + X, rec_y + (size_t)Y * stride + X, stride, qp, 1, lev + ORC_L_LUMA + b * 64) << (4 * b);
    }
    *nzmask |= ORC_NZ_T8; /* transform_size_8x8_flag of an I_NxN macroblock is sent whatever its levels */
}

/* Reconstruction of one intra macroblock with the decision `dec` (8.3 + 8.5): prediction from the reconstructed neighbours in
 * rec_y / rec_uv (whatever their type: constrained_intra_pred_flag is 0), residual, levels, record. */
static void intra_mb(const uint8_t *src_y, const uint8_t *src_uv, uint8_t *rec_y, uint8_t *rec_uv, int stride, int mbw, int mx, int my, int qp,
                     const orc_idec_t *dec, orc_mbinfo_t *mbi, int16_t *levels, int iac) {
            orc_mbinfo_t *m = &mbi[my * mbw + mx];
            int16_t *lev = levels + (size_t)(my * mbw + mx) * ORC_LEVELS_PER_MB;
            memset(lev, 0, ORC_LEVELS_PER_MB * sizeof(int16_t));
            qp = mb_qp(qp, my * mbw + mx);
            int x0 = mx * 16, y0 = my * 16, has_top = top_ok(my), has_left = mx > 0;
            m->mb_type = 0; m->mvx = 0; m->mvy = 0; m->qp = (uint8_t)qp; m->nzmask = 0;
            const int best_mode = dec->mode16, best_cmode = dec->cmode, use_i4 = dec->use_i4;
            m->i16_mode = (uint8_t)(use_i4 ? 0 : best_mode);
            m->chroma_mode = (uint8_t)best_cmode;
            if (use_i4) { m->mb_type = 2; for (int b = 0; b < 16; b++) lev[ORC_L_LDC + b] = dec->modes4[b]; } /* (Intra_8x8: four modes, the rest of the slot zero) */
            m->cost = dec->cost;
            if (use_i4 == 2) intra8x8_recon(src_y, rec_y, stride, mbw, mx, my, qp, lev, &m->nzmask);
            else if (use_i4) intra4x4_recon(src_y, rec_y, stride, mbw, mx, my, qp, lev, &m->nzmask);
            else {
            uint8_t best_pred[256];
            pred16(rec_y, stride, x0, y0, best_mode, has_top, has_left, best_pred);
            /* --- luma residual: 16 x (4x4 core), DCs through the 4x4 Hadamard (8.5.10 inverse) */
            int16_t dcs[16]; /* raster over 4x4 blocks: index (by/4)*4 + bx/4 */
            for (int b = 0; b < 16; b++) {
                int bx = k_blk_x[b], by = k_blk_y[b];
                int16_t res[16], dc;
                for (int y = 0; y < 4; y++)
                    for (int x = 0; x < 4; x++)
                        res[y * 4 + x] = (int16_t)(src_y[(size_t)(y0 + by + y) * stride + x0 + bx + x] - best_pred[(by + y) * 16 + bx + x]);
                if (tq_block(res, qp, 1, 1, lev + ORC_L_LUMA + b * 16, &dc)) m->nzmask |= 1u << b;
                dcs[(by / 4) * 4 + bx / 4] = dc;
            }

            /* forward Hadamard, halved with rounding (encoder choice), then DC quantiser */
            int t[16], hd[16];
            for (int i = 0; i < 4; i++) {
                int a = dcs[i * 4], b = dcs[i * 4 + 1], c = dcs[i * 4 + 2], d = dcs[i * 4 + 3];
                t[i * 4 + 0] = a + b + c + d; t[i * 4 + 1] = a + b - c - d;
                t[i * 4 + 2] = a - b - c + d; t[i * 4 + 3] = a - b + c - d;
            }
            for (int j = 0; j < 4; j++) {
                int a = t[j], b = t[4 + j], c = t[8 + j], d = t[12 + j];
                hd[j] = (a + b + c + d + 1) >> 1; hd[4 + j] = (a + b - c - d + 1) >> 1;
                hd[8 + j] = (a - b - c + d + 1) >> 1; hd[12 + j] = (a - b + c - d + 1) >> 1;
            }
            int16_t *ldc = lev + ORC_L_LDC;
            for (int k = 0; k < 16; k++) {
                ldc[k] = (int16_t)quant_dc(hd[k_zigzag4[k]], qp, 1);
                if (ldc[k]) m->nzmask |= ORC_NZ_LDC;
            }
            if (iac) { /* rate control's ladder for I pictures: a macroblock whose luma levels (DC and AC) sum to no more than the threshold sends none */
                int sum = 0;
                for (int i = 0; i < 256; i++) sum += iabs(lev[ORC_L_LUMA + i]);
                for (int k = 0; k < 16; k++) sum += iabs(ldc[k]);
                if (sum <= iac) { memset(lev + ORC_L_LUMA, 0, 256 * sizeof(int16_t)); memset(ldc, 0, 16 * sizeof(int16_t)); m->nzmask &= ~(0xFFFFu | ORC_NZ_LDC); }
            }
            /* 7.3.5.3: Intra16x16 AC levels are sent for all 16 blocks or none (cbp luma 15/0) */
            /* --- luma reconstruction: 8.5.10 (DC) then 8.5.12 per block */
            int cm[16], f[16], dcy[16];
            for (int k = 0; k < 16; k++) cm[k_zigzag4[k]] = ldc[k];
            for (int i = 0; i < 4; i++) {
                int a = cm[i * 4], b = cm[i * 4 + 1], c = cm[i * 4 + 2], d = cm[i * 4 + 3];
                t[i * 4 + 0] = a + b + c + d; t[i * 4 + 1] = a + b - c - d;
                t[i * 4 + 2] = a - b - c + d; t[i * 4 + 3] = a - b + c - d;
            }
            for (int j = 0; j < 4; j++) {
                int a = t[j], b = t[4 + j], c = t[8 + j], d = t[12 + j];
                f[j] = a + b + c + d; f[4 + j] = a + b - c - d;
                f[8 + j] = a - b - c + d; f[12 + j] = a - b + c - d;
            }
            int ls = 16 * k_dequant_v[qp % 6][0];
            for (int k = 0; k < 16; k++)
                dcy[k] = qp >= 36 ? (f[k] * ls) << (qp / 6 - 6) : (f[k] * ls + (1 << (5 - qp / 6))) >> (6 - qp / 6);
            for (int y = 0; y < 16; y++) memcpy(rec_y + (size_t)(y0 + y) * stride + x0, best_pred + y * 16, 16);
            for (int b = 0; b < 16; b++) {
                int bx = k_blk_x[b], by = k_blk_y[b];
                int32_t d[16];
                dq_block(lev + ORC_L_LUMA + b * 16, qp, 1, 1, dcy[(by / 4) * 4 + bx / 4], d);
                orc_idct4_add(d, rec_y + (size_t)(y0 + by) * stride + x0 + bx, stride);
            }
            } /* !use_i4 */
            /* --- chroma: prediction from the reconstructed neighbours with the analysed mode */
            for (int c = 0; c < 2; c++) {
                uint8_t cp[64];
                pred_chroma(rec_uv, stride, x0 / 2, y0 / 2, c, best_cmode, has_top, has_left, cp);
                for (int y = 0; y < 8; y++)
                    for (int x = 0; x < 8; x++) UV(rec_uv, stride, x0 / 2 + x, y0 / 2 + y, c) = cp[y * 8 + x];
            }
            chroma_tq_recon(src_uv, rec_uv, stride, x0 / 2, y0 / 2, qp, 1, lev, &m->nzmask, iac);
}

/* I picture.  Analysis and decisions (above) need only the source picture; then per macroblock in raster order the
 * reconstruction with the chosen modes. */
const int32_t k_idrop_ac[ORC_DROP_MAX + 1] = {0, 1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 64, 0x7FFFFFFF};
void orc_intra_frame(const uint8_t *src_y, const uint8_t *src_uv, uint8_t *rec_y,
                     uint8_t *rec_uv, int stride, int mbw, int mbh, int qp, int drop,
                     orc_mbinfo_t *mbi, int16_t *levels) {
    const int iac = k_idrop_ac[CLIP3(0, ORC_DROP_MAX, drop)];
    orc_isad_t *isad = (orc_isad_t *)malloc((size_t)mbw * mbh * sizeof(orc_isad_t));
    orc_idec_t *idec = (orc_idec_t *)malloc((size_t)mbw * mbh * sizeof(orc_idec_t));
    orc_intra_analyse(src_y, src_uv, stride, mbw, mbh, isad);
    orc_intra_decide(isad, mbw, mbh, qp, drop > 0 ? 0 : g_orc_i4x4, idec); /* on the ladder: Intra_16x16 only (Intra_4x4 costs its mode bits whatever the residual) */
    if (g_orc_t8 && g_orc_i8x8 && drop == 0 && qp <= ORC_I8_QP_MAX) orc_intra_decide8(src_y, stride, mbw, mbh, qp, idec);
    for (int my = 0; my < mbh; my++)
        for (int mx = 0; mx < mbw; mx++) intra_mb(src_y, src_uv, rec_y, rec_uv, stride, mbw, mx, my, qp, &idec[my * mbw + mx], mbi, levels, iac);
    free(isad);
    free(idec);
}
/* The intra macroblocks of a P picture (mb_type 0 / 2 left by orc_pmb_frame), in raster order, after every inter macroblock
 * of the picture has been reconstructed: their neighbours' samples (before deblocking) are what 8.3 predicts from. */
void orc_intra_p_frame(const uint8_t *src_y, const uint8_t *src_uv, uint8_t *rec_y, uint8_t *rec_uv, int stride, int mbw, int mbh, int qp,
                       const orc_idec_t *idec, orc_mbinfo_t *mbi, int16_t *levels) {
    for (int my = 0; my < mbh; my++)
        for (int mx = 0; mx < mbw; mx++)
            if (mbi[my * mbw + mx].mb_type != 1) intra_mb(src_y, src_uv, rec_y, rec_uv, stride, mbw, mx, my, qp, &idec[my * mbw + mx], mbi, levels, 0);
}

/* ================================================================== P pictures: the fused macroblock stage */
/* Everything below is encoder choice unless a clause is cited.  Per P picture and macroblock, independently of every other
 * macroblock's RESULT (neighbours are consulted only through the whole-sample vector field of orc_me_frame, which is complete
 * before this stage starts -- so the device runs the stage as one flat launch):
 *   1. predictor estimates from that field: p_est = 8.4.1.3 median over A, B, C of the field (every neighbour taken as inter,
 *      refIdx 0); ps_est = the 8.4.1.1 P_Skip inference on the field (0 on the left / top border or next to a zero vector).
 *      They equal the true predictors wherever the neighbours keep their whole-sample vectors.
 *   2. skip probe at ps_est (as x264's probe_pskip): if the residual of the prediction at ps_est quantises to nothing (luma
 *      after decimation, chroma DC and AC after decimation) the macroblock takes ps_est and no residual, and nothing else is
 *      evaluated.  Rate control's ladder below QP 51 (drop > 0) widens this: SAD(ps_est) < T[drop] passes as well.
 *   3. sub-sample refinement around the whole-sample winner, bits against p_est: 8 half-sample neighbours by SAD, then the
 *      8 quarter-sample neighbours by SATD (4x4 Hadamard, x264 subme 2) -- strictly cheaper wins, (dy, dx) raster order.
 *   4. intra or inter (when the open-loop intra analysis of the macroblock is available; Intra_16x16 only in P pictures):
 *      SAD-domain comparison.
 *   5. inter residual: 4x4 transform, dead-zone quantiser, coefficient decimation (x264 dct-decimate: an 8x8 whose
 *      run/level score is below 4 and a macroblock whose score is below 6 are emptied; chroma AC of a plane below 7),
 *      normative reconstruction.  drop > 0 and SAD(final) < T[drop]: no residual at all. */
static int g_orc_feat = ORC_F_ALL & ~ORC_F_I4P; /* Intra_4x4 in P pictures is an option (the device's intra_in_p = 2): rate-distortion neutral on the S2 / S4 clips, ten dependent sub-steps per macroblock */
static int g_tune[8] = {ORC_SKIP_MARGIN_BITS, 0, 12, 3, 0, 0, 0, 0}; /* dev: skip margin bits, skip shift (0 = none), intra bias bits, intra shift */
extern int g_sel_bonus;
void orc_set_tuning(int which, int value) { if (which >= 0 && which < 8) g_tune[which] = value; if (which == 4) g_sel_bonus = value; }
void orc_set_features(int mask) { g_orc_feat = mask; }
int orc_get_features(void) { return g_orc_feat; }

const uint32_t k_drop_sad[ORC_DROP_MAX + 1] = {0, 384, 512, 768, 1024, 1536, 2048, 3072, 4096, 6144, 8192, 12288, 0xFFFFFFFFu};
uint32_t orc_drop_threshold(int drop) { return k_drop_sad[CLIP3(0, ORC_DROP_MAX, drop)]; }

/* x264's decimate score of one 4x4 block (levels in scan order from `first`): 9 as soon as a level exceeds 1 in magnitude,
 * otherwise the sum over its +-1 levels of a weight that falls with the run of zeros below the level */
int orc_decimate_score(const int16_t *lev, int first) {
    static const uint8_t w[16] = {3, 2, 2, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    int idx = 15, score = 0;
    while (idx >= first && lev[idx] == 0) idx--;
    while (idx >= first) {
        if (lev[idx] > 1 || lev[idx] < -1) return 9;
        idx--;
        int run = 0;
        while (idx >= first && lev[idx] == 0) { idx--; run++; }
        score += w[run];
    }
    return score;
}
/* luma prediction of the macroblock at quarter-sample vector (qx, qy) into pred[256] */
static void luma_pred16(const uint8_t *ref_y, int stride, int W, int H, int x0, int y0, int qx, int qy, uint8_t *pred) {
    for (int y = 0; y < 16; y++)
        for (int x = 0; x < 16; x++) pred[y * 16 + x] = (uint8_t)luma_qpel(ref_y, stride, W, H, x0 + x + (qx >> 2), y0 + y + (qy >> 2), qx & 3, qy & 3);
}
static uint32_t sad16(const uint8_t *src, int stride, const uint8_t *pred) {
    uint32_t s = 0;
    for (int y = 0; y < 16; y++) for (int x = 0; x < 16; x++) s += (uint32_t)iabs(src[(size_t)y * stride + x] - pred[y * 16 + x]);
    return s;
}
/* sum over the sixteen 4x4 blocks of sum |H d H^T| (4x4 Hadamard of the difference), halved once at the end */
uint32_t orc_satd16(const uint8_t *src, int stride, const uint8_t *pred) {
    uint32_t total = 0;
    for (int by = 0; by < 16; by += 4)
        for (int bx = 0; bx < 16; bx += 4) {
            int d[16], t[16];
            for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) d[y * 4 + x] = src[(size_t)(by + y) * stride + bx + x] - pred[(by + y) * 16 + bx + x];
            for (int i = 0; i < 4; i++) {
                int a = d[i * 4] + d[i * 4 + 3], b = d[i * 4 + 1] + d[i * 4 + 2], c = d[i * 4 + 1] - d[i * 4 + 2], e = d[i * 4] - d[i * 4 + 3];
                t[i * 4] = a + b; t[i * 4 + 1] = e + c; t[i * 4 + 2] = a - b; t[i * 4 + 3] = e - c;
            }
            for (int j = 0; j < 4; j++) {
                int a = t[j] + t[12 + j], b = t[4 + j] + t[8 + j], c = t[4 + j] - t[8 + j], e = t[j] - t[12 + j];
                total += (uint32_t)(iabs(a + b) + iabs(e + c) + iabs(a - b) + iabs(e - c));
            }
        }
    return total >> 1;
}
/* luma of an inter macroblock: residual of src against pred -> levels (scan order), decimation; returns the blkIdx mask of
 * blocks that keep levels.  Nothing is reconstructed here. */
static uint32_t inter_luma_tq(const uint8_t *src, int stride, const uint8_t *pred, int qp, int decimate, int16_t *lev /* 16 x 16 */) {
    uint32_t nz = 0;
    int score[16];
    for (int b = 0; b < 16; b++) {
        int16_t res[16];
        for (int y = 0; y < 4; y++)
            for (int x = 0; x < 4; x++) res[y * 4 + x] = (int16_t)(src[(size_t)(k_blk_y[b] + y) * stride + k_blk_x[b] + x] - pred[(k_blk_y[b] + y) * 16 + k_blk_x[b] + x]);
        if (tq_block(res, qp, 0, 0, lev + b * 16, NULL)) nz |= 1u << b;
        score[b] = orc_decimate_score(lev + b * 16, 0);
    }
    if (decimate) {
        int total = 0;
        for (int g = 0; g < 4; g++) { /* luma4x4BlkIdx 4g .. 4g+3 form one 8x8 block (6.4.3) */
            const int s8 = score[4 * g] + score[4 * g + 1] + score[4 * g + 2] + score[4 * g + 3];
            total += s8;
            if (s8 < 4) { memset(lev + 64 * g, 0, 64 * sizeof(int16_t)); nz &= ~(0xFu << (4 * g)); }
        }
        if (total < 6) { memset(lev, 0, 256 * sizeof(int16_t)); nz = 0; }
    }
    return nz;
}
/* chroma of an inter macroblock whose prediction is already in rec_uv: levels, decimation of each plane's AC, reconstruction */
static void inter_chroma_tq_recon(const uint8_t *src_uv, uint8_t *rec_uv, int stride, int cx0, int cy0, int qp, int decimate,
                                  int16_t *lev, uint32_t *nzmask, int probe_only) {
    const int qpc = k_chroma_qp[CLIP3(0, 51, qp)];
    for (int c = 0; c < 2; c++) {
        int16_t dc[4];
        int16_t *ldc = lev + ORC_L_CDC + 4 * c;
        int sc = 0;
        for (int b = 0; b < 4; b++) {
            int bx = (b & 1) * 4, by = (b >> 1) * 4;
            int16_t res[16];
            for (int y = 0; y < 4; y++)
                for (int x = 0; x < 4; x++)
                    res[y * 4 + x] = (int16_t)(UV(src_uv, stride, cx0 + bx + x, cy0 + by + y, c) - UV(rec_uv, stride, cx0 + bx + x, cy0 + by + y, c));
            int16_t *l = lev + ORC_L_CAC + (4 * c + b) * 16;
            if (tq_block(res, qpc, 0, 1, l, &dc[b])) *nzmask |= 1u << (16 + 4 * c + b);
            sc += orc_decimate_score(l, 1);
        }
        if (decimate && sc < 7) { /* the plane's AC levels are not worth their bits */
            memset(lev + ORC_L_CAC + 4 * c * 16, 0, 64 * sizeof(int16_t));
            *nzmask &= ~(0xFu << (16 + 4 * c));
        }
        int f0 = dc[0] + dc[1] + dc[2] + dc[3], f1 = dc[0] - dc[1] + dc[2] - dc[3];
        int f2 = dc[0] + dc[1] - dc[2] - dc[3], f3 = dc[0] - dc[1] - dc[2] + dc[3];
        ldc[0] = (int16_t)quant_dc(f0, qpc, 0); ldc[1] = (int16_t)quant_dc(f1, qpc, 0);
        ldc[2] = (int16_t)quant_dc(f2, qpc, 0); ldc[3] = (int16_t)quant_dc(f3, qpc, 0);
        if (ldc[0] | ldc[1] | ldc[2] | ldc[3]) *nzmask |= (c ? ORC_NZ_CRDC : ORC_NZ_CBDC);
    }
    if (probe_only) return;
    for (int c = 0; c < 2; c++) { /* 8.5.11.1/2 + 8.5.12 */
        const int16_t *ldc = lev + ORC_L_CDC + 4 * c;
        int g0 = ldc[0] + ldc[1] + ldc[2] + ldc[3], g1 = ldc[0] - ldc[1] + ldc[2] - ldc[3];
        int g2 = ldc[0] + ldc[1] - ldc[2] - ldc[3], g3 = ldc[0] - ldc[1] - ldc[2] + ldc[3];
        int ls = 16 * k_dequant_v[qpc % 6][0];
        int dcv[4] = {((g0 * ls) << (qpc / 6)) >> 5, ((g1 * ls) << (qpc / 6)) >> 5, ((g2 * ls) << (qpc / 6)) >> 5, ((g3 * ls) << (qpc / 6)) >> 5};
        for (int b = 0; b < 4; b++) {
            int bx = (b & 1) * 4, by = (b >> 1) * 4;
            int32_t d[16];
            dq_block(lev + ORC_L_CAC + (4 * c + b) * 16, qpc, 1, 1, dcv[b], d);
            idct4_add_step(d, &UV(rec_uv, stride, cx0 + bx, cy0 + by, c), stride, 2);
        }
    }
}
/* chroma prediction only (8.4.2.2.2) */
static void chroma_pred8(const uint8_t *ref_uv, uint8_t *rec_uv, int stride, int W, int H, int x0, int y0, int mvx, int mvy) {
    int cw = W / 2, ch = H / 2, cx0 = x0 / 2, cy0 = y0 / 2;
    int xi = mvx >> 3, yi = mvy >> 3, xf = mvx & 7, yf = mvy & 7;
    for (int c = 0; c < 2; c++)
        for (int y = 0; y < 8; y++)
            for (int x = 0; x < 8; x++) {
                int ax = CLIP3(0, cw - 1, cx0 + x + xi), bx = CLIP3(0, cw - 1, cx0 + x + xi + 1);
                int ay = CLIP3(0, ch - 1, cy0 + y + yi), cy = CLIP3(0, ch - 1, cy0 + y + yi + 1);
                int A = UV(ref_uv, stride, ax, ay, c), B = UV(ref_uv, stride, bx, ay, c);
                int C = UV(ref_uv, stride, ax, cy, c), D = UV(ref_uv, stride, bx, cy, c);
                UV(rec_uv, stride, cx0 + x, cy0 + y, c) = (uint8_t)(((8 - xf) * (8 - yf) * A + xf * (8 - yf) * B + (8 - xf) * yf * C + xf * yf * D + 32) >> 6);
            }
}

/* ---- inter partitions (ORC_F_PART; encoder choice, oracle-side groundwork: the device does not produce them yet).
 * Record: an inter macroblock's i16_mode holds its shape (0 16x16, 1 16x8, 2 8x16, 3 8x8); mvx / mvy are partition 0's vector, the vectors of partitions
 * 1 .. 3 lie in the (otherwise unused) luma-DC slot of its levels, lev[ORC_L_LDC + 2 (idx - 1) + {0, 1}].
 * Search (the form the device stage can afford: no prediction is built that the macroblock's refinement does not build anyway): every partition chooses among
 * the up to 17 vectors the macroblock's refinement visits -- the whole-sample vector, its eight half-sample neighbours, the eight quarter-sample neighbours of
 * the half-sample winner -- by the SAD of its 8x8 quadrants + lambda * bits against the predictor estimate; a shape is taken when its partitions' costs plus
 * lambda * (its mb_type / sub_mb_type bits) are strictly below the 16x16 cost + lambda.  (part_search below -- whole-sample offsets -2 .. +2 and sub-sample
 * rounds of its own per partition -- is the dearer form it was measured against: 1-3 % fewer bits on the S2 clip against ... for this one.) */
static const int8_t k_part_geo[4][4][4] = {{{0, 0, 16, 16}}, {{0, 0, 16, 8}, {0, 8, 16, 8}}, {{0, 0, 8, 16}, {8, 0, 8, 16}}, {{0, 0, 8, 8}, {8, 0, 8, 8}, {0, 8, 8, 8}, {8, 8, 8, 8}}};
static const int8_t k_part_n[4] = {1, 2, 2, 4}, k_part_hdr_bits[4] = {1, 3, 3, 9}; /* ue(mb_type) (+ four ue(0) sub_mb_type) */
static const int16_t *g_part_lev; /* levels of the picture being filtered / written: where the partitions' vectors lie (NULL: 16x16 only) */
void orc_set_part_levels(const int16_t *levels) { g_part_lev = levels; }
static inline int mb_part(const orc_mbinfo_t *m) { return (g_part_lev && m->mb_type == 1) ? (m->i16_mode & 3) : 0; }
static void mb_qmv(const orc_mbinfo_t *mbi, int mbn, int q, int *vx, int *vy) { /* the vector of 8x8 quadrant q (raster) of an inter macroblock */
    const orc_mbinfo_t *m = &mbi[mbn];
    const int part = mb_part(m), idx = part == 0 ? 0 : part == 1 ? (q >> 1) : part == 2 ? (q & 1) : q;
    if (idx == 0) { *vx = m->mvx; *vy = m->mvy; }
    else { const int16_t *l = g_part_lev + (size_t)mbn * ORC_LEVELS_PER_MB + ORC_L_LDC + 2 * (idx - 1); *vx = l[0]; *vy = l[1]; }
}
static void luma_pred_part(const uint8_t *ref_y, int stride, int W, int H, int X, int Y, int w, int h, int qx, int qy, uint8_t *pred /* stride 16, at (0,0) */) {
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) pred[y * 16 + x] = (uint8_t)luma_qpel(ref_y, stride, W, H, X + x + (qx >> 2), Y + y + (qy >> 2), qx & 3, qy & 3);
}
static uint32_t sad_part(const uint8_t *src, int stride, const uint8_t *pred, int w, int h) {
    uint32_t s = 0;
    for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) s += (uint32_t)iabs(src[(size_t)y * stride + x] - pred[y * 16 + x]);
    return s;
}
static uint32_t satd_part(const uint8_t *src, int stride, const uint8_t *pred, int w, int h) { /* as orc_satd16, over the partition's 4x4 blocks */
    uint32_t total = 0;
    for (int by = 0; by < h; by += 4)
        for (int bx = 0; bx < w; bx += 4) {
            int d[16], t[16];
            for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) d[y * 4 + x] = src[(size_t)(by + y) * stride + bx + x] - pred[(by + y) * 16 + bx + x];
            for (int i = 0; i < 4; i++) {
                int a = d[i * 4] + d[i * 4 + 3], b = d[i * 4 + 1] + d[i * 4 + 2], c = d[i * 4 + 1] - d[i * 4 + 2], e = d[i * 4] - d[i * 4 + 3];
                t[i * 4] = a + b; t[i * 4 + 1] = e + c; t[i * 4 + 2] = a - b; t[i * 4 + 3] = e - c;
            }
            for (int j = 0; j < 4; j++) {
                int a = t[j] + t[12 + j], b = t[4 + j] + t[8 + j], c = t[4 + j] - t[8 + j], e = t[j] - t[12 + j];
                total += (uint32_t)(iabs(a + b) + iabs(e + c) + iabs(a - b) + iabs(e - c));
            }
        }
    return total >> 1;
}
static void chroma_pred_part(const uint8_t *ref_uv, uint8_t *rec_uv, int stride, int W, int H, int x0, int y0, int px0, int py0, int w, int h, int mvx, int mvy) {
    int cw = W / 2, ch = H / 2, cx0 = x0 / 2, cy0 = y0 / 2;
    int xi = mvx >> 3, yi = mvy >> 3, xf = mvx & 7, yf = mvy & 7;
    for (int c = 0; c < 2; c++)
        for (int y = py0 / 2; y < (py0 + h) / 2; y++)
            for (int x = px0 / 2; x < (px0 + w) / 2; x++) {
                int ax = CLIP3(0, cw - 1, cx0 + x + xi), bx = CLIP3(0, cw - 1, cx0 + x + xi + 1);
                int ay = CLIP3(0, ch - 1, cy0 + y + yi), cy = CLIP3(0, ch - 1, cy0 + y + yi + 1);
                int A = UV(ref_uv, stride, ax, ay, c), B = UV(ref_uv, stride, bx, ay, c);
                int C = UV(ref_uv, stride, ax, cy, c), D = UV(ref_uv, stride, bx, cy, c);
                UV(rec_uv, stride, cx0 + x, cy0 + y, c) = (uint8_t)(((8 - xf) * (8 - yf) * A + xf * (8 - yf) * B + (8 - xf) * yf * C + xf * yf * D + 32) >> 6);
            }
}
/* one partition: whole-sample offsets around (cx, cy) (quarter-sample units, the macroblock's vector), then the two sub-sample rounds; returns the final
 * SAD cost (distortion + lambda * bits against (px, py)), the vector in *vx, *vy, the prediction in pred (stride 16) */
static uint32_t part_search(const uint8_t *sy, const uint8_t *ref_y, int stride, int W, int H, int X, int Y, int w, int h, int cx, int cy, int px, int py,
                            int lambda, int range, int refine, int use_satd, int *vx, int *vy, uint8_t *pred) {
    const int ix = cx & ~3, iy = cy & ~3; /* the whole-sample position the local search is centred on */
    int bx = ix, by = iy;
    uint32_t best = 0xFFFFFFFFu;
    for (int dy = -2; dy <= 2; dy++)
        for (int dx = -2; dx <= 2; dx++) {
            const int qx = ix + 4 * dx, qy = iy + 4 * dy;
            if (qx < -4 * range || qx > 4 * range || qy < -4 * range || qy > 4 * range) continue;
            luma_pred_part(ref_y, stride, W, H, X, Y, w, h, qx, qy, pred);
            const uint32_t c = sad_part(sy, stride, pred, w, h) + (uint32_t)(lambda * (se_bits(qx - px) + se_bits(qy - py)));
            if (c < best) { best = c; bx = qx; by = qy; }
        }
    if (refine)
        for (int step = 2; step >= 1; step--) {
            const int satd = step == 1 && use_satd;
            if (satd) { luma_pred_part(ref_y, stride, W, H, X, Y, w, h, bx, by, pred); best = satd_part(sy, stride, pred, w, h) + (uint32_t)(lambda * (se_bits(bx - px) + se_bits(by - py))); }
            const int ox = bx, oy = by;
            for (int dy = -1; dy <= 1; dy++)
                for (int dx = -1; dx <= 1; dx++) {
                    if (!dx && !dy) continue;
                    const int qx = ox + dx * step, qy = oy + dy * step;
                    luma_pred_part(ref_y, stride, W, H, X, Y, w, h, qx, qy, pred);
                    const uint32_t d = satd ? satd_part(sy, stride, pred, w, h) : sad_part(sy, stride, pred, w, h);
                    const uint32_t c = d + (uint32_t)(lambda * (se_bits(qx - px) + se_bits(qy - py)));
                    if (c < best) { best = c; bx = qx; by = qy; }
                }
        }
    luma_pred_part(ref_y, stride, W, H, X, Y, w, h, bx, by, pred);
    *vx = bx; *vy = by;
    return sad_part(sy, stride, pred, w, h) + (uint32_t)(lambda * (se_bits(bx - px) + se_bits(by - py)));
}

void orc_pmb_frame(const uint8_t *src_y, const uint8_t *src_uv, const uint8_t *ref_y, const uint8_t *ref_uv, uint8_t *rec_y, uint8_t *rec_uv,
                   int stride, int mbw, int mbh, int qp, int drop, int refine, const orc_imv_t *imv, const uint16_t *surf, const orc_idec_t *idec,
                   orc_mbinfo_t *mbi, int16_t *levels, int threads) {
    const int W = mbw * 16, H = mbh * 16, lambda = orc_me_lambda(qp), feat = g_orc_feat;
    const uint32_t tdrop = drop > 0 ? orc_drop_threshold(drop) : 0;
    (void)threads;
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads > 0 ? threads : 1)
    for (int my = 0; my < mbh; my++)
        for (int mx = 0; mx < mbw; mx++) {
            const int mbn = my * mbw + mx, x0 = mx * 16, y0 = my * 16;
            orc_mbinfo_t *m = &mbi[mbn];
            int16_t *lev = levels + (size_t)mbn * ORC_LEVELS_PER_MB;
            const uint8_t *sy = src_y + (size_t)y0 * stride + x0;
            uint8_t pred[256];
            int px, py, sx, sy_;
            memset(lev, 0, ORC_LEVELS_PER_MB * sizeof(int16_t));
            const int mqp = mb_qp(qp, mbn); /* quantisation only: search, refinement and decisions keep the picture's lambda */
            m->mb_type = 1; m->i16_mode = 0; m->chroma_mode = 0; m->qp = (uint8_t)mqp; m->nzmask = 0;
            field_pred(imv, mbw, mx, my, &px, &py, &sx, &sy_);
            if (!(feat & ORC_F_MVDCOST)) { px = 0; py = 0; }
            /* ---- 2. skip probe */
            if (feat & ORC_F_SKIPPROBE) {
                /* ps_est is a median of in-range whole-sample vectors: its SAD is on the surface already.  The probe is only worth
                 * running when that prediction is not much worse than the best whole-sample one (at high QP everything quantises to
                 * nothing, and the distortion is all there is to tell two vectors apart). */
                const uint32_t ds = surf[(size_t)mbn * ORC_SURF + ((sy_ >> 2) + 16) * 33 + (sx >> 2) + 16];
                const uint32_t di = imv[mbn].sad;
                int pass = tdrop && ds < tdrop;
                const int worth = ds <= di + (g_tune[1] ? (di >> g_tune[1]) : 0) + (uint32_t)(lambda * g_tune[0]);
                if (pass || worth) luma_pred16(ref_y, stride, W, H, x0, y0, sx, sy_, pred);
                if (!pass && worth) {
                    int16_t pl[ORC_LEVELS_PER_MB];
                    uint32_t pnz = 0;
                    memset(pl, 0, sizeof pl);
                    if (inter_luma_tq(sy, stride, pred, mqp, 1, pl) == 0) {
                        chroma_pred8(ref_uv, rec_uv, stride, W, H, x0, y0, sx, sy_);
                        inter_chroma_tq_recon(src_uv, rec_uv, stride, x0 / 2, y0 / 2, mqp, 1, pl, &pnz, 1);
                        pass = pnz == 0;
                    }
                }
                if (pass) {
                    m->mvx = (int16_t)sx; m->mvy = (int16_t)sy_; m->cost = di;
                    for (int y = 0; y < 16; y++) memcpy(rec_y + (size_t)(y0 + y) * stride + x0, pred + y * 16, 16);
                    chroma_pred8(ref_uv, rec_uv, stride, W, H, x0, y0, sx, sy_);
                    continue;
                }
            }
            /* ---- 3. refinement */
            int bx = imv[mbn].mvx, by = imv[mbn].mvy;
            luma_pred16(ref_y, stride, W, H, x0, y0, bx, by, pred);
            uint32_t best = sad16(sy, stride, pred) + (uint32_t)(lambda * (se_bits(bx - px) + se_bits(by - py)));
            /* (partitions choose among the vectors the macroblock's refinement visits: per candidate, the SAD of each 8x8 quadrant) */
            int ncand = 0, candx[17], candy[17];
            uint32_t candq[17][4];
#define PART_VISIT(qx_, qy_) do { if (feat & ORC_F_PART) { candx[ncand] = (qx_); candy[ncand] = (qy_); \
                for (int q_ = 0; q_ < 4; q_++) { candq[ncand][q_] = sad_part(sy + (size_t)((q_ >> 1) * 8) * stride + (q_ & 1) * 8, stride, pred + (q_ >> 1) * 128 + (q_ & 1) * 8, 8, 8); } \
                ncand++; } } while (0)
            PART_VISIT(bx, by);
            if (refine)
                for (int step = 2; step >= 1; step--) {
                    const int satd = step == 1 && (feat & ORC_F_SATD);
                    if (satd) { /* the quarter-sample round compares in the transform domain: restate the standing best there */
                        luma_pred16(ref_y, stride, W, H, x0, y0, bx, by, pred);
                        best = orc_satd16(sy, stride, pred) + (uint32_t)(lambda * (se_bits(bx - px) + se_bits(by - py)));
                    }
                    const int cx = bx, cy = by;
                    for (int dy = -1; dy <= 1; dy++)
                        for (int dx = -1; dx <= 1; dx++) {
                            if (!dx && !dy) continue;
                            const int qx = cx + dx * step, qy = cy + dy * step;
                            luma_pred16(ref_y, stride, W, H, x0, y0, qx, qy, pred);
                            PART_VISIT(qx, qy);
                            const uint32_t d = satd ? orc_satd16(sy, stride, pred) : sad16(sy, stride, pred);
                            const uint32_t cost = d + (uint32_t)(lambda * (se_bits(qx - px) + se_bits(qy - py)));
                            if (cost < best) { best = cost; bx = qx; by = qy; }
                        }
                }
#undef PART_VISIT
            luma_pred16(ref_y, stride, W, H, x0, y0, bx, by, pred);
            const uint32_t dsad = sad16(sy, stride, pred);
            const uint32_t jinter = dsad + (uint32_t)(lambda * (se_bits(bx - px) + se_bits(by - py)));
            m->mvx = (int16_t)bx; m->mvy = (int16_t)by; m->cost = imv[mbn].sad; /* the record's cost is the whole-sample SAD source against source: what scene-cut detection sums, independent of QP and of what the macroblock became */
            /* ---- 3b. partitions (oracle-side groundwork, off by default) */
            int shape = 0, pvx[4] = {bx, 0, 0, 0}, pvy[4] = {by, 0, 0, 0};
            uint32_t jbest = jinter + (uint32_t)(lambda * k_part_hdr_bits[0]);
            uint8_t ppred[256];
            if ((feat & ORC_F_PART) && !g_orc_t8) {
                for (int sh = 1; sh <= 3; sh++) {
                    uint32_t j = (uint32_t)(lambda * k_part_hdr_bits[sh]);
                    int vx[4], vy[4];
                    for (int i = 0; i < k_part_n[sh]; i++) {
                        const int8_t *g = k_part_geo[sh][i];
                        uint32_t pb = 0xFFFFFFFFu;
                        for (int k = 0; k < ncand; k++) { /* ties: the candidate visited first */
                            uint32_t c = (uint32_t)(lambda * (se_bits(candx[k] - px) + se_bits(candy[k] - py)));
                            for (int q = 0; q < 4; q++) { const int qx = (q & 1) * 8, qy = (q >> 1) * 8; if (qx >= g[0] && qx < g[0] + g[2] && qy >= g[1] && qy < g[1] + g[3]) c += candq[k][q]; }
                            if (c < pb) { pb = c; vx[i] = candx[k]; vy[i] = candy[k]; }
                        }
                        j += pb;
                    }
                    if (j < jbest) { jbest = j; shape = sh; for (int i = 0; i < k_part_n[sh]; i++) { pvx[i] = vx[i]; pvy[i] = vy[i]; } }
                }
                if (shape)
                    for (int i = 0; i < k_part_n[shape]; i++) {
                        const int8_t *g = k_part_geo[shape][i];
                        uint8_t one[256];
                        luma_pred_part(ref_y, stride, W, H, x0 + g[0], y0 + g[1], g[2], g[3], pvx[i], pvy[i], one);
                        for (int y = 0; y < g[3]; y++) memcpy(ppred + (g[1] + y) * 16 + g[0], one + y * 16, (size_t)g[2]);
                    }
                if (shape) { /* the partitions' vectors may coincide: the shape is kept all the same (its cost said so) */
                    memcpy(pred, ppred, 256);
                    m->mvx = (int16_t)pvx[0]; m->mvy = (int16_t)pvy[0]; m->i16_mode = (uint8_t)shape;
                }
            }
            const uint32_t jinter_p = jbest - (uint32_t)(lambda * k_part_hdr_bits[0]); /* (a partitioned macroblock pays for its longer header here too) */
            /* ---- 4. intra instead?  (reconstructed later, by orc_intra_p_frame, once every inter macroblock is in place) */
            if ((feat & ORC_F_INTRAP) && idec && imv[mbn].sad + (uint32_t)(lambda * imv[mbn].bits) >= ORC_INTRA_GATE(lambda)) {
                const orc_idec_t *d = &idec[mbn];
                const uint32_t jintra = d->cost_luma + (g_tune[3] ? (d->cost_luma >> g_tune[3]) : 0) + (uint32_t)(lambda * g_tune[2]);
                if (jintra < jinter_p) {
                    m->mb_type = (uint8_t)(d->use_i4 ? 2 : 0); m->mvx = 0; m->mvy = 0;
                    m->i16_mode = (uint8_t)(d->use_i4 ? 0 : d->mode16); m->chroma_mode = d->cmode; m->cost = d->cost;
                    continue;
                }
            }
            /* ---- 5. residual */
            for (int y = 0; y < 16; y++) memcpy(rec_y + (size_t)(y0 + y) * stride + x0, pred + y * 16, 16);
            if (!shape) chroma_pred8(ref_uv, rec_uv, stride, W, H, x0, y0, bx, by);
            else
                for (int i = 0; i < k_part_n[shape]; i++) {
                    const int8_t *g = k_part_geo[shape][i];
                    chroma_pred_part(ref_uv, rec_uv, stride, W, H, x0, y0, g[0], g[1], g[2], g[3], pvx[i], pvy[i]);
                    if (i) { lev[ORC_L_LDC + 2 * (i - 1)] = (int16_t)pvx[i]; lev[ORC_L_LDC + 2 * (i - 1) + 1] = (int16_t)pvy[i]; }
                }
            if (tdrop && (shape ? sad16(sy, stride, pred) : dsad) < tdrop) continue; /* rate control's ladder below QP 51: prediction only */
            if (g_orc_t8) { /* High profile: every P_L0_16x16 macroblock through the 8x8 transform (encoder choice; no coefficient decimation on this path) */
                for (int i8 = 0; i8 < 4; i8++) {
                    const size_t o = (size_t)(y0 + (i8 >> 1) * 8) * stride + x0 + (i8 & 1) * 8;
                    m->nzmask |= (uint32_t)tq8_block(src_y + o, rec_y + o, stride, mqp, lev + ORC_L_LUMA + i8 * 64) << (4 * i8);
                }
                if (m->nzmask & 0xFFFF) m->nzmask |= ORC_NZ_T8; /* transform_size_8x8_flag is only sent (and only matters) with luma cbp != 0 */
            } else {
            m->nzmask = inter_luma_tq(sy, stride, pred, mqp, (feat & ORC_F_DECIMATE) != 0, lev + ORC_L_LUMA);
            for (int b = 0; b < 16; b++) {
                if (!(m->nzmask & (1u << b))) continue;
                int32_t d[16];
                dq_block(lev + ORC_L_LUMA + b * 16, mqp, 0, 0, 0, d);
                orc_idct4_add(d, rec_y + (size_t)(y0 + k_blk_y[b]) * stride + x0 + k_blk_x[b], stride);
            }
            }
            inter_chroma_tq_recon(src_uv, rec_uv, stride, x0 / 2, y0 / 2, mqp, (feat & ORC_F_DECIMATE) != 0, lev, &m->nzmask, 0);
        }
}

/* ================================================================== deblocking (8.7) */
/* one line of samples across an edge: p[0..3] going away from the edge on one side, q likewise */
static void filter_line(uint8_t *pix, int dstep, int bS, int qp_p, int qp_q, int chroma) {
    if (bS == 0) return;
    int qpav = (qp_p + qp_q + 1) >> 1;
    int indexA = CLIP3(0, 51, qpav), indexB = CLIP3(0, 51, qpav); /* FilterOffsetA = FilterOffsetB = 0 */
    int alpha = k_alpha[indexA], beta = k_beta[indexB];
    int p0 = pix[-1 * dstep], p1 = pix[-2 * dstep], p2 = pix[-3 * dstep];
    int q0 = pix[0], q1 = pix[1 * dstep], q2 = pix[2 * dstep];
    if (!(iabs(p0 - q0) < alpha && iabs(p1 - p0) < beta && iabs(q1 - q0) < beta)) return;
    if (bS < 4) { /* 8.7.2.3 */
        int tc0 = k_tc0[indexA][bS - 1];
        int ap = iabs(p2 - p0), aq = iabs(q2 - q0);
        int tc = chroma ? tc0 + 1 : tc0 + (ap < beta) + (aq < beta);
        int delta = CLIP3(-tc, tc, (((q0 - p0) << 2) + (p1 - q1) + 4) >> 3);
        pix[-1 * dstep] = (uint8_t)clip1(p0 + delta);
        pix[0] = (uint8_t)clip1(q0 - delta);
        if (!chroma) {
            if (ap < beta) pix[-2 * dstep] = (uint8_t)(p1 + CLIP3(-tc0, tc0, (p2 + ((p0 + q0 + 1) >> 1) - (p1 << 1)) >> 1));
            if (aq < beta) pix[1 * dstep] = (uint8_t)(q1 + CLIP3(-tc0, tc0, (q2 + ((p0 + q0 + 1) >> 1) - (q1 << 1)) >> 1));
        }
    } else { /* 8.7.2.4 */
        if (chroma) {
            pix[-1 * dstep] = (uint8_t)((2 * p1 + p0 + q1 + 2) >> 2);
            pix[0] = (uint8_t)((2 * q1 + q0 + p1 + 2) >> 2);
        } else {
            int p3 = pix[-4 * dstep], q3 = pix[3 * dstep];
            int ap = iabs(p2 - p0), aq = iabs(q2 - q0);
            int small = iabs(p0 - q0) < ((alpha >> 2) + 2);
            if (ap < beta && small) {
                pix[-1 * dstep] = (uint8_t)((p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3);
                pix[-2 * dstep] = (uint8_t)((p2 + p1 + p0 + q0 + 2) >> 2);
                pix[-3 * dstep] = (uint8_t)((2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3);
            } else pix[-1 * dstep] = (uint8_t)((2 * p1 + p0 + q1 + 2) >> 2);
            if (aq < beta && small) {
                pix[0] = (uint8_t)((p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3);
                pix[1 * dstep] = (uint8_t)((p0 + q0 + q1 + q2 + 2) >> 2);
                pix[2 * dstep] = (uint8_t)((2 * q3 + 3 * q2 + q1 + q0 + p0 + 4) >> 3);
            } else pix[0] = (uint8_t)((2 * q1 + q0 + p1 + 2) >> 2);
        }
    }
}
/* 8.7.2.1 boundary strength between the 4x4 luma blocks at raster positions (bxp,byp) of
 * macroblock mp and (bxq,byq) of macroblock mq (frame pictures, one reference picture). */
static int blk_has_coef(const orc_mbinfo_t *m, int bx4, int by4) {
    static const uint8_t raster_to_blk[16] = {0, 1, 4, 5, 2, 3, 6, 7, 8, 9, 12, 13, 10, 11, 14, 15};
    int b = raster_to_blk[by4 * 4 + bx4];
    if (m->mb_type != 1) return 1; /* intra handled before this is consulted */
    if (m->nzmask & ORC_NZ_T8) return ((m->nzmask >> (b & ~3)) & 0xF) != 0; /* 8.7.2.1: the 8x8 block containing the sample */
    return (m->nzmask >> b) & 1;
}
static const orc_mbinfo_t *g_db_base; /* the picture's records (orc_deblock_frame): a record's index is what finds its partitions' vectors */
static int bs_of(const orc_mbinfo_t *mp, int bxp, int byp, const orc_mbinfo_t *mq, int bxq, int byq, int mb_edge) {
    if (mp->mb_type != 1 || mq->mb_type != 1) return mb_edge ? 4 : 3; /* mb_type 0 (I16x16) and 2 (I4x4) are intra */
    if (blk_has_coef(mp, bxp, byp) || blk_has_coef(mq, bxq, byq)) return 2;
    int pvx = mp->mvx, pvy = mp->mvy, qvx = mq->mvx, qvy = mq->mvy;
    if (g_part_lev && g_db_base) { /* the vectors of the 8x8 quadrants the two 4x4 blocks lie in */
        mb_qmv(g_db_base, (int)(mp - g_db_base), (byp >> 1) * 2 + (bxp >> 1), &pvx, &pvy);
        mb_qmv(g_db_base, (int)(mq - g_db_base), (byq >> 1) * 2 + (bxq >> 1), &qvx, &qvy);
    }
    if (iabs(pvx - qvx) >= 4 || iabs(pvy - qvy) >= 4) return 1; /* quarter-sample units */
    return 0;
}
void orc_deblock_frame(uint8_t *rec_y, uint8_t *rec_uv, int stride, int mbw, int mbh,
                       const orc_mbinfo_t *mbi) {
    g_db_base = mbi;
    for (int my = 0; my < mbh; my++)
        for (int mx = 0; mx < mbw; mx++) {
            const orc_mbinfo_t *m = &mbi[my * mbw + mx];
            int x0 = mx * 16, y0 = my * 16;
            int qpc_q = k_chroma_qp[m->qp];
            /* vertical edges, left to right */
            const int t8 = (m->nzmask & ORC_NZ_T8) != 0; /* transform_size_8x8_flag: luma edges 1 and 3 are not transform edges */
            for (int e = 0; e < 4; e++) {
                if (e == 0 && mx == 0) continue;
                const orc_mbinfo_t *mp = e == 0 ? m - 1 : m;
                int qpc_p = k_chroma_qp[mp->qp];
                if (!(t8 && (e & 1)))
                for (int k = 0; k < 16; k++) {
                    int bS = bs_of(mp, e == 0 ? 3 : e - 1, k >> 2, m, e, k >> 2, e == 0);
                    filter_line(rec_y + (size_t)(y0 + k) * stride + x0 + 4 * e, 1, bS, mp->qp, m->qp, 0);
                }
                if ((e & 1) == 0) /* chroma edges at chroma x = 0 and 4 */
                    for (int c = 0; c < 2; c++)
                        for (int k = 0; k < 8; k++) {
                            int bS = bs_of(mp, e == 0 ? 3 : e - 1, k >> 1, m, e, k >> 1, e == 0);
                            filter_line(&UV(rec_uv, stride, x0 / 2 + 2 * e, y0 / 2 + k, c), 2, bS, qpc_p, qpc_q, 1);
                        }
            }
            /* horizontal edges, top to bottom */
            for (int e = 0; e < 4; e++) {
                if (e == 0 && (my == 0 || (g_slice_dbf == 2 && !top_ok(my)))) continue; /* 8.7: filterTopMbEdgeFlag = 0 at the picture's top, and with disable_deblocking_filter_idc 2 where the macroblock above is in another slice */
                const orc_mbinfo_t *mp = e == 0 ? m - mbw : m;
                int qpc_p = k_chroma_qp[mp->qp];
                if (!(t8 && (e & 1)))
                for (int k = 0; k < 16; k++) {
                    int bS = bs_of(mp, k >> 2, e == 0 ? 3 : e - 1, m, k >> 2, e, e == 0);
                    filter_line(rec_y + (size_t)(y0 + 4 * e) * stride + x0 + k, stride, bS, mp->qp, m->qp, 0);
                }
                if ((e & 1) == 0)
                    for (int c = 0; c < 2; c++)
                        for (int k = 0; k < 8; k++) {
                            int bS = bs_of(mp, k >> 1, e == 0 ? 3 : e - 1, m, k >> 1, e, e == 0);
                            filter_line(&UV(rec_uv, stride, x0 / 2 + k, y0 / 2 + 2 * e, c), stride, bS, qpc_p, qpc_q, 1);
                        }
            }
        }
}

/* ================================================================== CAVLC (9.2) */
/* One residual block: coef[0..maxnum-1] in scan order.  Returns TotalCoeff. */
static int cavlc_block(bw_t *bw, const int16_t *coef, int maxnum, int nC) {
    int idx[16], n = 0;
    for (int i = 0; i < maxnum; i++) if (coef[i]) idx[n++] = i;
    int total = n, t1 = 0;
    for (int i = n - 1; i >= 0 && t1 < 3; i--) {
        if (coef[idx[i]] == 1 || coef[idx[i]] == -1) t1++;
        else break;
    }
    vlc_t tok;
    if (nC < 0) tok = t_coeff_token_cdc[total][t1];
    else tok = t_coeff_token[nC < 2 ? 0 : nC < 4 ? 1 : nC < 8 ? 2 : 3][total][t1];
    bw_put(bw, tok.len, tok.bits);
    if (!total) return 0;
    for (int i = 0; i < t1; i++) bw_put(bw, 1, coef[idx[n - 1 - i]] < 0); /* trailing_ones_sign_flag */
    /* 9.2.2.1 levels, highest frequency first */
    int suffix_len = (total > 10 && t1 < 3) ? 1 : 0;
    for (int i = n - 1 - t1; i >= 0; i--) {
        int lv = coef[idx[i]];
        int code = lv > 0 ? 2 * lv - 2 : -2 * lv - 1; /* levelCode */
        if (i == n - 1 - t1 && t1 < 3) code -= 2;
        if (suffix_len == 0) {
            if (code < 14) bw_put(bw, code + 1, 1);
            else if (code < 30) { bw_put(bw, 15, 1); bw_put(bw, 4, (uint32_t)(code - 14)); }
            else { bw_put(bw, 16, 1); bw_put(bw, 12, (uint32_t)(code - 30)); }
        } else {
            if (code < (15 << suffix_len)) {
                bw_put(bw, (code >> suffix_len) + 1, 1);
                bw_put(bw, suffix_len, (uint32_t)(code & ((1 << suffix_len) - 1)));
            } else { bw_put(bw, 16, 1); bw_put(bw, 12, (uint32_t)(code - (15 << suffix_len))); }
        }
        if (suffix_len == 0) suffix_len = 1;
        if (iabs(lv) > (3 << (suffix_len - 1)) && suffix_len < 6) suffix_len++;
    }
    /* total_zeros */
    int zeros_left = idx[n - 1] + 1 - total;
    if (total < maxnum) {
        vlc_t tz = maxnum == 4 ? t_total_zeros_cdc[total - 1][zeros_left] : t_total_zeros[total - 1][zeros_left];
        bw_put(bw, tz.len, tz.bits);
    }
    /* run_before, highest frequency first, not for the last (lowest) coefficient */
    for (int i = n - 1; i > 0 && zeros_left > 0; i--) {
        int run = idx[i] - idx[i - 1] - 1;
        vlc_t rb = t_run_before[(zeros_left > 7 ? 7 : zeros_left) - 1][run];
        bw_put(bw, rb.len, rb.bits);
        zeros_left -= run;
    }
    return total;
}

int orc_cavlc_block_bits(const int16_t *coef, int maxnum, int nC, uint8_t *out, size_t cap) {
    init_tables();
    bw_t b;
    memset(out, 0, cap);
    bw_init(&b, out, cap);
    cavlc_block(&b, coef, maxnum, nC);
    int n = (int)(8 * b.pos) + b.nbits;
    if (b.nbits) bw_put(&b, 8 - b.nbits, 0);
    return b.overflow ? -1 : n;
}

/* 8.4.1.3 motion vector prediction for a 16x16 partition with refIdx 0 (quarter-pel units
 * are not needed: vectors are compared/added as integer-pel*4 by the caller).
 * type[] : -1 unavailable, 0 intra (refIdx -1), 1 inter (refIdx 0). */
/* 6.4.11.7 / 8.4.1.3.2: motion data of the 8x8 block covering luma sample (X, Y) as a neighbour of a partition of macroblock (mx, my), whose own
 * quadrants in `done` carry the vectors cur[q]; macroblocks later in raster order and macroblocks of another slice are not available. */
static void nb_blk(const orc_mbinfo_t *mbi, int mbw, int mbh, int mx, int my, unsigned done, int cur[4][2], int X, int Y, int *avail, int *ref, int *vx, int *vy) {
    *avail = 0; *ref = -1; *vx = *vy = 0;
    if (X < 0 || Y < 0 || X >= mbw * 16 || Y >= mbh * 16) return;
    const int nx = X >> 4, ny = Y >> 4, q = ((Y & 15) >> 3) * 2 + ((X & 15) >> 3);
    if (nx == mx && ny == my) { if ((done >> q) & 1) { *avail = 1; *ref = 0; *vx = cur[q][0]; *vy = cur[q][1]; } return; }
    if (!(ny < my || (ny == my && nx < mx))) return;
    if (ny < my && !top_ok(my)) return; /* another slice (slices are whole rows: only the row above can be one) */
    *avail = 1;
    if (mbi[ny * mbw + nx].mb_type == 1) { *ref = 0; mb_qmv(mbi, ny * mbw + nx, q, vx, vy); }
}
/* 8.4.1.3 for partition idx of shape part at (x0, y0), width w; skip: 8.4.1.1's inference */
static void mv_pred_part(const orc_mbinfo_t *mbi, int mbw, int mbh, int mx, int my, unsigned done, int cur[4][2], int part, int idx, int x0, int y0, int w, int skip, int *px, int *py) {
    const int X = mx * 16 + x0, Y = my * 16 + y0;
    int aA, rA, ax, ay, aB, rB, bx, by, aC, rC, cx, cy;
    nb_blk(mbi, mbw, mbh, mx, my, done, cur, X - 1, Y, &aA, &rA, &ax, &ay);
    nb_blk(mbi, mbw, mbh, mx, my, done, cur, X, Y - 1, &aB, &rB, &bx, &by);
    nb_blk(mbi, mbw, mbh, mx, my, done, cur, X + w, Y - 1, &aC, &rC, &cx, &cy);
    if (!aC) nb_blk(mbi, mbw, mbh, mx, my, done, cur, X - 1, Y - 1, &aC, &rC, &cx, &cy);
    *px = 0; *py = 0;
    if (skip && (!aA || !aB || (rA == 0 && !ax && !ay) || (rB == 0 && !bx && !by))) return;
    if (part == 1 && idx == 0 && rB == 0) { *px = bx; *py = by; return; }
    if (part == 1 && idx == 1 && rA == 0) { *px = ax; *py = ay; return; }
    if (part == 2 && idx == 0 && rA == 0) { *px = ax; *py = ay; return; }
    if (part == 2 && idx == 1 && rC == 0) { *px = cx; *py = cy; return; }
    if (!aB && !aC && aA) { rB = rC = rA; bx = cx = ax; by = cy = ay; }
    const int hits = (rA == 0) + (rB == 0) + (rC == 0);
    if (hits == 1) {
        if (rA == 0) { *px = ax; *py = ay; } else if (rB == 0) { *px = bx; *py = by; } else { *px = cx; *py = cy; }
        return;
    }
    *px = median3(ax, bx, cx);
    *py = median3(ay, by, cy);
}
static int g_pred_mbh = 1 << 20; /* (the 16x16 forms below never look below the current row) */
/* 8.4.1.1 P_Skip vector */
static void mv_pred_skip(const orc_mbinfo_t *mbi, int mbw, int mx, int my, int *px, int *py) { int cur[4][2] = {{0}}; mv_pred_part(mbi, mbw, g_pred_mbh, mx, my, 0, cur, 0, 0, 0, 0, 16, 1, px, py); }

static int level_idc_for(int mbw, int mbh, int fps_num, int fps_den) {
    /* Table A-1: {level_idc, MaxMBPS, MaxFS} */
    static const int lv[][3] = {{10, 1485, 99},      {11, 3000, 396},     {12, 6000, 396},    {13, 11880, 396},
                                {20, 11880, 396},    {21, 19800, 792},    {22, 20250, 1620},  {30, 40500, 1620},
                                {31, 108000, 3600},  {32, 216000, 5120},  {40, 245760, 8192}, {42, 522240, 8704},
                                {50, 589824, 22080}, {51, 983040, 36864}, {52, 2073600, 36864}};
    int64_t fs = (int64_t)mbw * mbh;
    int64_t mbps = (fs * fps_num + fps_den - 1) / fps_den;
    for (size_t i = 0; i < sizeof lv / sizeof lv[0]; i++)
        if (fs <= lv[i][2] && mbps <= lv[i][1] && mbw * mbw <= 8 * lv[i][2] && mbh * mbh <= 8 * lv[i][2]) return lv[i][0];
    return 52;
}

/* 7.3.2.1 SPS + 7.3.2.2 PPS, Annex B.  Constrained Baseline (profile_idc 66, set0+set1). */
size_t orc_write_headers(uint8_t *out, size_t cap, int width, int height, int fps_num, int fps_den) {
    init_tables();
    uint8_t rb[128];
    bw_t b;
    int mbw = (width + 15) / 16, mbh = (height + 15) / 16;
    bw_init(&b, rb, sizeof rb);
    bw_put(&b, 8, g_orc_t8 ? 100 : 66); /* profile_idc: High when the 8x8 transform is enabled, else (Constrained) Baseline */
    bw_put(&b, 8, g_orc_t8 ? 0x00 : 0xC0); /* constraint_set0,1 = 1 for Constrained Baseline */
    bw_put(&b, 8, (uint32_t)level_idc_for(mbw, mbh, fps_num, fps_den));
    bw_ue(&b, 0);                      /* seq_parameter_set_id */
    if (g_orc_t8) { bw_ue(&b, 1); bw_ue(&b, 0); bw_ue(&b, 0); bw_put(&b, 2, 0); } /* chroma_format_idc 1, 8-bit, no bypass, no scaling matrix */
    bw_ue(&b, 4);                      /* log2_max_frame_num_minus4 -> 8 bits */
    bw_ue(&b, 2);                      /* pic_order_cnt_type */
    bw_ue(&b, 1);                      /* max_num_ref_frames */
    bw_put(&b, 1, 0);                  /* gaps_in_frame_num_value_allowed_flag */
    bw_ue(&b, (uint32_t)(mbw - 1));
    bw_ue(&b, (uint32_t)(mbh - 1));
    bw_put(&b, 1, 1);                  /* frame_mbs_only_flag */
    bw_put(&b, 1, 1);                  /* direct_8x8_inference_flag */
    int crop_r = (mbw * 16 - width) / 2, crop_b = (mbh * 16 - height) / 2;
    if (crop_r || crop_b) {
        bw_put(&b, 1, 1);
        bw_ue(&b, 0); bw_ue(&b, (uint32_t)crop_r); bw_ue(&b, 0); bw_ue(&b, (uint32_t)crop_b);
    } else bw_put(&b, 1, 0);
    bw_put(&b, 1, 1);                  /* vui_parameters_present_flag (E.1.1) */
    bw_put(&b, 1, 0);                  /* aspect_ratio_info_present_flag */
    bw_put(&b, 1, 0);                  /* overscan_info_present_flag */
    bw_put(&b, 1, 0);                  /* video_signal_type_present_flag */
    bw_put(&b, 1, 0);                  /* chroma_loc_info_present_flag */
    bw_put(&b, 1, 1);                  /* timing_info_present_flag */
    bw_put(&b, 32, (uint32_t)fps_den); /* num_units_in_tick */
    bw_put(&b, 32, (uint32_t)(2 * fps_num)); /* time_scale */
    bw_put(&b, 1, 1);                  /* fixed_frame_rate_flag */
    bw_put(&b, 1, 0);                  /* nal_hrd_parameters_present_flag */
    bw_put(&b, 1, 0);                  /* vcl_hrd_parameters_present_flag */
    bw_put(&b, 1, 0);                  /* pic_struct_present_flag */
    bw_put(&b, 1, 1);                  /* bitstream_restriction_flag */
    bw_put(&b, 1, 1);                  /* motion_vectors_over_pic_boundaries_flag */
    bw_ue(&b, 0);                      /* max_bytes_per_pic_denom */
    bw_ue(&b, 0);                      /* max_bits_per_mb_denom */
    bw_ue(&b, 10);                     /* log2_max_mv_length_horizontal */
    bw_ue(&b, 10);                     /* log2_max_mv_length_vertical */
    bw_ue(&b, 0);                      /* max_num_reorder_frames */
    bw_ue(&b, 1);                      /* max_dec_frame_buffering */
    bw_trailing(&b);
    size_t n = write_nal(out, cap, 3, 7, rb, b.pos);
    if (!n) return 0;
    bw_init(&b, rb, sizeof rb);
    bw_ue(&b, 0);                      /* pic_parameter_set_id */
    bw_ue(&b, 0);                      /* seq_parameter_set_id */
    bw_put(&b, 1, 0);                  /* entropy_coding_mode_flag: CAVLC */
    bw_put(&b, 1, 0);                  /* bottom_field_pic_order_in_frame_present_flag */
    bw_ue(&b, 0);                      /* num_slice_groups_minus1 */
    bw_ue(&b, 0);                      /* num_ref_idx_l0_default_active_minus1 */
    bw_ue(&b, 0);                      /* num_ref_idx_l1_default_active_minus1 */
    bw_put(&b, 1, 0);                  /* weighted_pred_flag */
    bw_put(&b, 2, 0);                  /* weighted_bipred_idc */
    bw_se(&b, 0);                      /* pic_init_qp_minus26 */
    bw_se(&b, 0);                      /* pic_init_qs_minus26 */
    bw_se(&b, 0);                      /* chroma_qp_index_offset */
    bw_put(&b, 1, 1);                  /* deblocking_filter_control_present_flag */
    bw_put(&b, 1, 0);                  /* constrained_intra_pred_flag */
    bw_put(&b, 1, 0);                  /* redundant_pic_cnt_present_flag */
    if (g_orc_t8) { bw_put(&b, 1, 1); bw_put(&b, 1, 0); bw_se(&b, 0); } /* transform_8x8_mode_flag, no pic scaling matrix, second_chroma_qp_index_offset */
    bw_trailing(&b);
    size_t m = write_nal(out + n, cap - n, 3, 8, rb, b.pos);
    if (!m) return 0;
    return n + m;
}

/* 7.3.3 slice header + 7.3.4 slice data + 7.3.5 macroblock layer, one slice per picture. */
size_t orc_write_slice(uint8_t *out, size_t cap, int mbw, int mbh, int is_idr, int frame_num,
                       int idr_pic_id, int qp, const orc_mbinfo_t *mbi, const int16_t *levels) {
    init_tables();
    int nmb = mbw * mbh;
    size_t rcap = (size_t)nmb * 1024 + 256; /* > 3200 bits/MB worst case (A.3.1) */
    uint8_t *rb = (uint8_t *)malloc(rcap);
    uint8_t *tc_l = (uint8_t *)calloc((size_t)nmb, 16); /* TotalCoeff per luma blkIdx   */
    uint8_t *tc_c = (uint8_t *)calloc((size_t)nmb, 8);  /* per chroma AC block (Cb 0-3, Cr 4-7) */
    if (!rb || !tc_l || !tc_c) { free(rb); free(tc_l); free(tc_c); return 0; }
    const int srows = g_slice_rows > 0 ? g_slice_rows : 0; /* a new slice every so many macroblock rows */
    size_t total = 0;
    bw_t b;
#define SLICE_HEADER(first_mb)                                                                        \
    do {                                                                                              \
        bw_init(&b, rb, rcap);                                                                        \
        bw_ue(&b, (uint32_t)(first_mb));              /* first_mb_in_slice */                         \
        bw_ue(&b, is_idr ? 7 : 5);                    /* slice_type: all slices of the picture I / P */ \
        bw_ue(&b, 0);                                 /* pic_parameter_set_id */                      \
        bw_put(&b, 8, (uint32_t)(frame_num & 0xFF));  /* frame_num, log2_max_frame_num = 8 */         \
        if (is_idr) bw_ue(&b, (uint32_t)idr_pic_id);                                                  \
        if (!is_idr) bw_put(&b, 1, 0);                /* num_ref_idx_active_override_flag */          \
        if (!is_idr) bw_put(&b, 1, 0);                /* ref_pic_list_modification_flag_l0 */         \
        if (is_idr) { bw_put(&b, 1, 0); bw_put(&b, 1, 0); } /* no_output_of_prior_pics, long_term_reference */ \
        else bw_put(&b, 1, 0);                        /* adaptive_ref_pic_marking_mode_flag */        \
        bw_se(&b, qp - 26);                           /* slice_qp_delta */                            \
        bw_ue(&b, (uint32_t)g_slice_dbf);             /* disable_deblocking_filter_idc */             \
        bw_se(&b, 0);                                 /* slice_alpha_c0_offset_div2 */                \
        bw_se(&b, 0);                                 /* slice_beta_offset_div2 */                    \
    } while (0)
    SLICE_HEADER(0);

    static const uint8_t blk_raster[16] = {0, 1, 4, 5, 2, 3, 6, 7, 8, 9, 12, 13, 10, 11, 14, 15}; /* blkIdx -> by*4+bx */
    static const uint8_t raster_blk[16] = {0, 1, 4, 5, 2, 3, 6, 7, 8, 9, 12, 13, 10, 11, 14, 15}; /* by*4+bx -> blkIdx */
    int skip_run = 0, prev_qp = qp;
    for (int my = 0; my < mbh; my++) {
        if (srows && my > 0 && my % srows == 0) { /* the slice ends (P slices: with the skip run still pending, 7.3.4), the next one starts */
            if (!is_idr && skip_run) { bw_ue(&b, (uint32_t)skip_run); skip_run = 0; }
            bw_trailing(&b);
            const size_t n = b.overflow ? 0 : write_nal(out + total, cap - total, is_idr ? 3 : 2, is_idr ? 5 : 1, rb, b.pos);
            if (!n) { free(rb); free(tc_l); free(tc_c); return 0; }
            total += n;
            SLICE_HEADER(my * mbw);
            prev_qp = qp;
        }
        const int top = srows ? my % srows != 0 : my > 0; /* the row above belongs to this slice */
        for (int mx = 0; mx < mbw; mx++) {
            int mbn = my * mbw + mx;
            const orc_mbinfo_t *m = &mbi[mbn];
            const int16_t *lev = levels + (size_t)mbn * ORC_LEVELS_PER_MB;
            int cbp_luma = 0, cbp_chroma = 0;
            const int intra = m->mb_type != 1, i16 = m->mb_type == 0;
            if (i16) cbp_luma = (m->nzmask & 0xFFFF) ? 15 : 0;
            else for (int g = 0; g < 4; g++) if ((m->nzmask >> (4 * g)) & 0xF) cbp_luma |= 1 << g;
            if (m->nzmask & 0x00FF0000u) cbp_chroma = 2;
            else if (m->nzmask & (ORC_NZ_CBDC | ORC_NZ_CRDC)) cbp_chroma = 1;

            if (!is_idr && m->mb_type == 1) {
                int sx, sy;
                mv_pred_skip(mbi, mbw, mx, my, &sx, &sy);
                if (cbp_luma == 0 && cbp_chroma == 0 && mb_part(m) == 0 && m->mvx == sx && m->mvy == sy) { skip_run++; continue; }
            }
            if (!is_idr) { bw_ue(&b, (uint32_t)skip_run); skip_run = 0; }
            if (i16) {
                int t = 1 + m->i16_mode + 4 * cbp_chroma + (cbp_luma ? 12 : 0); /* Table 7-11 */
                bw_ue(&b, (uint32_t)(is_idr ? t : t + 5));
                bw_ue(&b, m->chroma_mode);                                        /* intra_chroma_pred_mode */
            } else if (intra) { /* I_NxN (Intra_4x4 / Intra_8x8): 7.3.5.1 mb_pred */
                static const uint8_t rb[16] = {0, 1, 4, 5, 2, 3, 6, 7, 8, 9, 12, 13, 10, 11, 14, 15}; /* raster -> blkIdx (self-inverse) */
                const int i8 = (m->nzmask & ORC_NZ_T8) != 0;
                bw_ue(&b, is_idr ? 0u : 5u);
                if (g_orc_t8) bw_put(&b, 1, (uint32_t)i8); /* transform_size_8x8_flag: Intra_8x8 or Intra_4x4 */
                /* the mode a neighbouring macroblock contributes for its 4x4 block blk (8.3.1.1 / 8.3.2.1): Intra4x4PredMode[blk], Intra8x8PredMode[blk >> 2], or DC for any other type */
#define NXN_MODE(mm, ll, blk) ((mm)->mb_type != 2 ? 2 : ((mm)->nzmask & ORC_NZ_T8) ? (ll)[ORC_L_LDC + ((blk) >> 2)] : (ll)[ORC_L_LDC + (blk)])
                if (i8) {
                    for (int b8 = 0; b8 < 4; b8++) {
                        int ma = -1, mb_ = -1;
                        if (b8 & 1) ma = lev[ORC_L_LDC + b8 - 1];
                        else if (mx > 0) ma = NXN_MODE(&m[-1], lev - ORC_LEVELS_PER_MB, (b8 + 1) * 4 + 1);
                        if (b8 >> 1) mb_ = lev[ORC_L_LDC + b8 - 2];
                        else if (top) mb_ = NXN_MODE(&m[-mbw], lev - (ptrdiff_t)mbw * ORC_LEVELS_PER_MB, (b8 + 2) * 4 + 2);
                        const int pm = (ma < 0 || mb_ < 0) ? 2 : (ma < mb_ ? ma : mb_), mode = lev[ORC_L_LDC + b8];
                        if (mode == pm) bw_put(&b, 1, 1);
                        else { bw_put(&b, 1, 0); bw_put(&b, 3, (uint32_t)(mode < pm ? mode : mode - 1)); }
                    }
                } else
                for (int blk = 0; blk < 16; blk++) {
                    int bx = rb[blk] & 3, by = rb[blk] >> 2, ma = -1, mb_ = -1;
                    if (bx > 0) ma = lev[ORC_L_LDC + rb[by * 4 + bx - 1]];
                    else if (mx > 0) ma = NXN_MODE(&m[-1], lev - ORC_LEVELS_PER_MB, rb[by * 4 + 3]);
                    if (by > 0) mb_ = lev[ORC_L_LDC + rb[(by - 1) * 4 + bx]];
                    else if (top) mb_ = NXN_MODE(&m[-mbw], lev - (ptrdiff_t)mbw * ORC_LEVELS_PER_MB, rb[12 + bx]);
                    int pm = (ma < 0 || mb_ < 0) ? 2 : (ma < mb_ ? ma : mb_), mode = lev[ORC_L_LDC + blk];
                    if (mode == pm) bw_put(&b, 1, 1);
                    else { bw_put(&b, 1, 0); bw_put(&b, 3, (uint32_t)(mode < pm ? mode : mode - 1)); }
                }
#undef NXN_MODE
                bw_ue(&b, m->chroma_mode);
                bw_ue(&b, k_cbp_to_codenum_intra[cbp_chroma * 16 + cbp_luma]);
            } else {
                const int part = mb_part(m); /* 0 P_L0_16x16, 1 P_L0_L0_16x8, 2 P_L0_L0_8x16, 3 P_8x8 (four P_L0_8x8) */
                bw_ue(&b, (uint32_t)part);
                if (part == 3) for (int i = 0; i < 4; i++) bw_ue(&b, 0); /* sub_mb_type */
                int cur[4][2];
                unsigned done = 0;
                for (int i = 0; i < k_part_n[part]; i++) { /* mvd_l0 of the partitions in order, quarter-sample units (ref_idx is not sent: one reference) */
                    const int8_t *g = k_part_geo[part][i];
                    int px, py, vx, vy;
                    mv_pred_part(mbi, mbw, mbh, mx, my, done, cur, part, i, g[0], g[1], g[2], 0, &px, &py);
                    mb_qmv(mbi, mbn, (g[1] >> 3) * 2 + (g[0] >> 3), &vx, &vy);
                    bw_se(&b, vx - px);
                    bw_se(&b, vy - py);
                    for (int q = 0; q < 4; q++) { const int qx = (q & 1) * 8, qy = (q >> 1) * 8; if (qx >= g[0] && qx < g[0] + g[2] && qy >= g[1] && qy < g[1] + g[3]) { cur[q][0] = vx; cur[q][1] = vy; done |= 1u << q; } }
                }
                bw_ue(&b, k_cbp_to_codenum_inter[cbp_chroma * 16 + cbp_luma]);
                if (g_orc_t8 && cbp_luma) bw_put(&b, 1, (m->nzmask & ORC_NZ_T8) ? 1 : 0); /* transform_size_8x8_flag */
            }
            if (i16 || cbp_luma || cbp_chroma) {
                bw_se(&b, m->qp - prev_qp); /* mb_qp_delta */
                prev_qp = m->qp;
            }
            /* neighbour TotalCoeff lookup, 9.2.1 */
#define NC_LUMA(bx, by, out)                                                                        \
    do {                                                                                            \
        int na = -1, nb = -1;                                                                       \
        if ((bx) > 0) na = tc_l[mbn * 16 + raster_blk[(by) * 4 + (bx) - 1]];                         \
        else if (mx > 0) na = tc_l[(mbn - 1) * 16 + raster_blk[(by) * 4 + 3]];                      \
        if ((by) > 0) nb = tc_l[mbn * 16 + raster_blk[((by) - 1) * 4 + (bx)]];                       \
        else if (top) nb = tc_l[(mbn - mbw) * 16 + raster_blk[12 + (bx)]];                          \
        (out) = (na >= 0 && nb >= 0) ? (na + nb + 1) >> 1 : (na >= 0 ? na : (nb >= 0 ? nb : 0));    \
    } while (0)
            if (i16) {
                int nC;
                NC_LUMA(0, 0, nC);
                cavlc_block(&b, lev + ORC_L_LDC, 16, nC);
            }
            for (int blk = 0; blk < 16; blk++) {
                if (!(cbp_luma & (1 << (blk >> 2)))) continue;
                int bx = blk_raster[blk] & 3, by = blk_raster[blk] >> 2, nC;
                NC_LUMA(bx, by, nC);
                int tc = i16 ? cavlc_block(&b, lev + ORC_L_LUMA + blk * 16 + 1, 15, nC)
                                         : cavlc_block(&b, lev + ORC_L_LUMA + blk * 16, 16, nC);
                tc_l[mbn * 16 + blk] = (uint8_t)tc;
            }
            if (cbp_chroma) {
                cavlc_block(&b, lev + ORC_L_CDC, 4, -1);
                cavlc_block(&b, lev + ORC_L_CDC + 4, 4, -1);
            }
            if (cbp_chroma == 2)
                for (int c = 0; c < 2; c++)
                    for (int blk = 0; blk < 4; blk++) {
                        int bx = blk & 1, by = blk >> 1, na = -1, nb = -1, nC;
                        if (bx > 0) na = tc_c[mbn * 8 + 4 * c + blk - 1];
                        else if (mx > 0) na = tc_c[(mbn - 1) * 8 + 4 * c + by * 2 + 1];
                        if (by > 0) nb = tc_c[mbn * 8 + 4 * c + blk - 2];
                        else if (top) nb = tc_c[(mbn - mbw) * 8 + 4 * c + 2 + bx];
                        nC = (na >= 0 && nb >= 0) ? (na + nb + 1) >> 1 : (na >= 0 ? na : (nb >= 0 ? nb : 0));
                        tc_c[mbn * 8 + 4 * c + blk] = (uint8_t)cavlc_block(&b, lev + ORC_L_CAC + (4 * c + blk) * 16 + 1, 15, nC);
                    }
#undef NC_LUMA
        }
    }
#undef SLICE_HEADER
    if (!is_idr && skip_run) bw_ue(&b, (uint32_t)skip_run);
    bw_trailing(&b);
    size_t n = b.overflow ? 0 : write_nal(out + total, cap - total, is_idr ? 3 : 2, is_idr ? 5 : 1, rb, b.pos);
    free(rb); free(tc_l); free(tc_c);
    return n ? total + n : 0;
}

/* ================================================================== encoder wrapper */
struct orc_enc {
    int width, height, mbw, mbh, stride, fps_num, fps_den, gop, me_range, threads, subpel;
    int frames_since_idr, idr_count, have_ref;
    int intra_slices;               /* slices per I picture (0: orc_auto_intra_slices) */
    int p_slices, slice_dbf;        /* slices per P picture (0 / 1: one); disable_deblocking_filter_idc of every slice (0 or 2) */
    int aq; int8_t *aq_off;                                                    /* adaptive quantisation: per-macroblock QP offsets of the picture being coded */
    int scenecut, sc_cnt, prev_idr, sc_lag, prev_all_skip;                      /* scene-cut recovery: mirrors enc_schedule.cpp (collect / enqueue_picture) */
    unsigned long long sc_sum, sc_force_at, pic_index;
    uint8_t *src_y, *src_uv, *rec_y[2], *rec_uv[2], *pre_y, *pre_uv, *prev_src_y;
    int prev_src_valid;
    int cur; /* index of the surface holding the last reconstructed picture */
    orc_mbinfo_t *mbi, *prev_mbi;   /* records of this picture / of the previous one (temporal vector predictor) */
    int prev_is_p;                  /* prev_mbi holds a P picture's records */
    orc_imv_t *imv, *imv2;
    uint16_t *surf;
    int me_iters;
    orc_isad_t *isad;
    orc_idec_t *idec;
    int16_t *levels;
    int last_qp;
};

orc_enc_t *orc_enc_open(int width, int height, int fps_num, int fps_den, int gop, int me_range, int threads) {
    if (width < 16 || height < 16 || (width & 1) || (height & 1) || gop < 1) return NULL;
    init_tables();
    orc_enc_t *e = (orc_enc_t *)calloc(1, sizeof *e);
    if (!e) return NULL;
    e->width = width; e->height = height;
    e->mbw = (width + 15) / 16; e->mbh = (height + 15) / 16; e->stride = e->mbw * 16;
    e->fps_num = fps_num; e->fps_den = fps_den; e->gop = gop; e->me_range = me_range; e->threads = threads; e->subpel = 1;
    e->scenecut = 1; e->sc_lag = 2; e->sc_force_at = ~0ull; e->last_qp = 26;
    size_t ysz = (size_t)e->stride * e->mbh * 16, csz = ysz / 2, nmb = (size_t)e->mbw * e->mbh;
    e->src_y = (uint8_t *)malloc(ysz); e->src_uv = (uint8_t *)malloc(csz);
    e->pre_y = (uint8_t *)malloc(ysz); e->pre_uv = (uint8_t *)malloc(csz); e->prev_src_y = (uint8_t *)malloc(ysz);
    for (int i = 0; i < 2; i++) { e->rec_y[i] = (uint8_t *)malloc(ysz); e->rec_uv[i] = (uint8_t *)malloc(csz); }
    e->mbi = (orc_mbinfo_t *)calloc(nmb, sizeof(orc_mbinfo_t));
    e->prev_mbi = (orc_mbinfo_t *)calloc(nmb, sizeof(orc_mbinfo_t));
    e->imv = (orc_imv_t *)calloc(nmb, sizeof(orc_imv_t));
    e->imv2 = (orc_imv_t *)calloc(nmb, sizeof(orc_imv_t));
    e->surf = (uint16_t *)malloc(nmb * ORC_SURF * sizeof(uint16_t));
    e->me_iters = ORC_ME_ITERS;
    e->isad = (orc_isad_t *)calloc(nmb, sizeof(orc_isad_t));
    e->idec = (orc_idec_t *)calloc(nmb, sizeof(orc_idec_t));
    e->levels = (int16_t *)calloc(nmb * ORC_LEVELS_PER_MB, sizeof(int16_t));
    e->aq_off = (int8_t *)calloc(nmb, 1);
    return e;
}
void orc_enc_close(orc_enc_t *e) {
    if (!e) return;
    free(e->src_y); free(e->src_uv); free(e->pre_y); free(e->pre_uv);
    for (int i = 0; i < 2; i++) { free(e->rec_y[i]); free(e->rec_uv[i]); }
    free(e->mbi); free(e->prev_mbi); free(e->imv); free(e->imv2); free(e->surf); free(e->isad); free(e->idec); free(e->levels); free(e->aq_off); free(e);
}
/* copy the visible picture into the coded-size surface, replicating the last column/row */
static void load_padded(orc_enc_t *e, const uint8_t *y, int ys, const uint8_t *uv, int uvs) {
    int W = e->stride, H = e->mbh * 16;
    for (int r = 0; r < H; r++) {
        const uint8_t *s = y + (size_t)(r < e->height ? r : e->height - 1) * ys;
        uint8_t *d = e->src_y + (size_t)r * W;
        memcpy(d, s, (size_t)e->width);
        for (int x = e->width; x < W; x++) d[x] = s[e->width - 1];
    }
    for (int r = 0; r < H / 2; r++) {
        const uint8_t *s = uv + (size_t)(r < e->height / 2 ? r : e->height / 2 - 1) * uvs;
        uint8_t *d = e->src_uv + (size_t)r * W;
        memcpy(d, s, (size_t)e->width);
        for (int x = e->width; x < W; x += 2) { d[x] = s[e->width - 2]; d[x + 1] = s[e->width - 1]; }
    }
}
/* One picture.  qp 0..51; drop 0..ORC_DROP_MAX (P pictures: rate control's ladder below QP 51); drop == ORC_DROP_SKIP: the
 * picture is coded as one run of P_Skip macroblocks (no source sample is looked at; the reconstruction is the reference) --
 * what rate control emits when even the ladder's last step would overshoot.  I pictures have a ladder of their own (drop
 * 1 .. ORC_DROP_MAX: Intra_16x16 only, and a macroblock's luma / chroma levels are not sent when their magnitudes sum to
 * no more than the level's threshold -- at the last level an I picture is prediction only); ORC_DROP_SKIP means nothing to
 * an I picture. */
int orc_enc_frame2(orc_enc_t *e, const uint8_t *y, int y_stride, const uint8_t *uv, int uv_stride,
                   int qp, int drop, int force_idr, uint8_t *out, size_t out_cap, size_t *out_len, int *is_idr) {
    if (!e || qp < 0 || qp > 51 || drop < 0 || (drop > ORC_DROP_MAX && drop != ORC_DROP_SKIP)) return -1;
    int idr = force_idr || !e->have_ref || e->frames_since_idr >= e->gop || (e->pic_index == e->sc_force_at && e->frames_since_idr >= e->sc_lag);
    if (idr) { e->frames_since_idr = 0; }
    const int nmb = e->mbw * e->mbh;
    const int all_skip = !idr && drop == ORC_DROP_SKIP;
    int nxt = e->cur ^ 1;
    size_t ysz = (size_t)e->stride * e->mbh * 16;
    {   /* the picture's slices (an all-skip picture stays one slice: there is nothing in it to cut) */
        const int ns = idr ? (e->intra_slices > 0 ? e->intra_slices : orc_auto_intra_slices(e->mbh)) : e->p_slices;
        g_slice_rows = all_skip ? 0 : orc_slice_rows_for(e->mbh, ns, e->slice_dbf == 2);
        g_slice_dbf = e->slice_dbf;
    }
    if (all_skip) {
        nxt = e->cur; /* the reconstruction IS the reference: every macroblock P_Skip with the zero vector, nothing to filter */
        for (int i = 0; i < nmb; i++) { memset(&e->mbi[i], 0, sizeof e->mbi[i]); e->mbi[i].mb_type = 1; e->mbi[i].qp = (uint8_t)qp; }
        memset(e->levels, 0, (size_t)nmb * ORC_LEVELS_PER_MB * sizeof(int16_t));
        memcpy(e->pre_y, e->rec_y[nxt], ysz);
        memcpy(e->pre_uv, e->rec_uv[nxt], ysz / 2);
    } else {
        load_padded(e, y, y_stride, uv, uv_stride);
        if (e->aq) { orc_aq_offsets(e->src_y, e->stride, e->mbw, e->mbh, e->aq_off); g_aq = e->aq_off; }
        if (idr)
            orc_intra_frame(e->src_y, e->src_uv, e->rec_y[nxt], e->rec_uv[nxt], e->stride, e->mbw, e->mbh, qp, drop == ORC_DROP_SKIP ? 0 : drop, e->mbi, e->levels);
        else {
            /* The whole-sample search runs SOURCE against SOURCE: the padded source of the last coded picture stands in for the
             * reference.  True motion is what it finds -- the fields are smoother than against a quantised reference (fewer vector
             * bits at the same distortion from QP 32 up) -- and the search of a picture no longer waits for anything of the picture
             * before it (on the device it runs beside that picture's deblocking).  Refinement and prediction use the reference. */
            orc_me_frame(e->src_y, e->prev_src_y, e->stride, e->mbw, e->mbh, e->me_range, qp, e->surf, e->imv, e->threads);
            if (g_orc_feat & ORC_F_MVDCOST)
                for (int it = 0; it < e->me_iters; it++) {
                    orc_me_select(e->surf, e->mbw, e->mbh, e->me_range, qp, e->imv, e->imv2, e->threads);
                    orc_imv_t *t = e->imv; e->imv = e->imv2; e->imv2 = t;
                }
            { /* (r03: the High-profile stream goes through the fused stage as well -- skip probe, refinement against the predictor estimates, intra macroblocks --
               * with the 8x8 transform for the luma residual of its inter macroblocks; the two-stage form, orc_subpel_frame + orc_inter_frame, remains as stage functions) */
                const int intra_p = (g_orc_feat & ORC_F_INTRAP) != 0;
                if (intra_p) {
                    orc_intra_analyse(e->src_y, e->src_uv, e->stride, e->mbw, e->mbh, e->isad);
                    orc_intra_decide(e->isad, e->mbw, e->mbh, qp, (g_orc_feat & ORC_F_I4P) && g_orc_i4x4, e->idec); /* (r03) Intra_4x4 as well: what x264's superfast keeps of its partition search (i8x8, i4x4) */
                }
                orc_pmb_frame(e->src_y, e->src_uv, e->rec_y[e->cur], e->rec_uv[e->cur], e->rec_y[nxt], e->rec_uv[nxt], e->stride, e->mbw, e->mbh,
                              qp, drop, e->subpel, e->imv, e->surf, intra_p ? e->idec : NULL, e->mbi, e->levels, e->threads);
                if (intra_p) orc_intra_p_frame(e->src_y, e->src_uv, e->rec_y[nxt], e->rec_uv[nxt], e->stride, e->mbw, e->mbh, qp, e->idec, e->mbi, e->levels);
            }
        }
        memcpy(e->pre_y, e->rec_y[nxt], ysz);
        memcpy(e->pre_uv, e->rec_uv[nxt], ysz / 2);
        if (g_aq) { orc_qp_chain_slices(e->mbi, nmb, qp, g_slice_rows * e->mbw); g_aq = NULL; }
        g_part_lev = (!idr && (g_orc_feat & ORC_F_PART)) ? e->levels : NULL;
        orc_deblock_frame(e->rec_y[nxt], e->rec_uv[nxt], e->stride, e->mbw, e->mbh, e->mbi);
    }
    size_t n = 0;
    if (idr) {
        n = orc_write_headers(out, out_cap, e->width, e->height, e->fps_num, e->fps_den);
        if (!n) return -2;
    }
    size_t s = orc_write_slice(out + n, out_cap - n, e->mbw, e->mbh, idr, e->frames_since_idr, e->idr_count & 0xFFFF,
                               qp, e->mbi, e->levels);
    g_slice_rows = 0; g_slice_dbf = 0; g_part_lev = NULL;
    if (!s) return -2;
    *out_len = n + s;
    if (is_idr) *is_idr = idr;
    if (idr) { e->idr_count++; e->sc_sum = 0; e->sc_cnt = 0; }
    else if (!all_skip && !e->prev_all_skip) { /* summed macroblock cost of this P picture against the mean of the P pictures since the last IDR
                                                  (not of a picture that follows P_Skip-run pictures: it was searched against an older source) */
        unsigned long long cost = 0;
        for (int i = 0; i < nmb; i++) cost += e->mbi[i].cost;
        const int pending = e->sc_force_at != ~0ull && e->sc_force_at > e->pic_index;
        if (e->scenecut && !pending && e->sc_cnt >= 2 && cost > 3 * (e->sc_sum / (unsigned long long)e->sc_cnt)) e->sc_force_at = e->pic_index + (unsigned long long)e->sc_lag;
        e->sc_sum += cost; e->sc_cnt++;
    }
    if (!all_skip) { memcpy(e->prev_src_y, e->src_y, ysz); e->prev_src_valid = 1; }
    memcpy(e->prev_mbi, e->mbi, (size_t)nmb * sizeof(orc_mbinfo_t));
    e->prev_is_p = !idr;
    e->prev_idr = idr; e->prev_all_skip = all_skip; e->pic_index++;
    e->frames_since_idr++;
    e->cur = nxt; e->have_ref = 1; e->last_qp = qp;
    return 0;
}
int orc_enc_frame(orc_enc_t *e, const uint8_t *y, int y_stride, const uint8_t *uv, int uv_stride,
                  int qp, int force_idr, uint8_t *out, size_t out_cap, size_t *out_len, int *is_idr) {
    return orc_enc_frame2(e, y, y_stride, uv, uv_stride, qp, 0, force_idr, out, out_cap, out_len, is_idr);
}
void orc_enc_set_subpel(orc_enc_t *e, int on) { e->subpel = on; }
void orc_enc_set_aq(orc_enc_t *e, int on) { e->aq = on; }
void orc_enc_set_intra_slices(orc_enc_t *e, int n) { e->intra_slices = n < 0 ? 0 : n; }
void orc_enc_set_p_slices(orc_enc_t *e, int n) { e->p_slices = n < 0 ? 0 : n; }
void orc_enc_set_slice_deblock(orc_enc_t *e, int local) { e->slice_dbf = local ? 2 : 0; }
void orc_enc_set_scenecut(orc_enc_t *e, int on) { e->scenecut = on; }
void orc_enc_set_sc_lag(orc_enc_t *e, int lag) { e->sc_lag = lag < 2 ? 2 : lag; } /* enc_schedule.cpp sc_lag(): pipeline_depth + 1 from depth 2 on */
void orc_enc_set_me_iters(orc_enc_t *e, int n) { e->me_iters = n < 0 ? 0 : n; }
const uint8_t *orc_enc_recon_y(const orc_enc_t *e) { return e->rec_y[e->cur]; }
const uint8_t *orc_enc_recon_uv(const orc_enc_t *e) { return e->rec_uv[e->cur]; }
const uint8_t *orc_enc_prefilter_y(const orc_enc_t *e) { return e->pre_y; }
const uint8_t *orc_enc_prefilter_uv(const orc_enc_t *e) { return e->pre_uv; }
const orc_mbinfo_t *orc_enc_mbinfo(const orc_enc_t *e) { return e->mbi; }
const orc_imv_t *orc_enc_imv(const orc_enc_t *e) { return e->imv; }
const orc_idec_t *orc_enc_idec(const orc_enc_t *e) { return e->idec; }
const int16_t *orc_enc_levels(const orc_enc_t *e) { return e->levels; }
int orc_enc_mbw(const orc_enc_t *e) { return e->mbw; }
int orc_enc_mbh(const orc_enc_t *e) { return e->mbh; }
