/*
 * oracle/h264_dec_oracle.c -- TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see h264_oracle.h).
 *
 * An independently written H.264 decoder (ITU-T H.264 clauses 7.3 parsing, 8.3 intra,
 * 8.4 inter, 8.5 transform, 8.7 deblocking, 9.1/9.2 entropy) used to pin the encoder
 * oracle and the HIP product: decode(bitstream) must equal the encoder's reconstruction
 * bit for bit.  It shares no code and no table text with h264_enc_oracle.c; its VLC
 * tables are transcribed in numeric (length, value) form, the encoder's as bit strings.
 * Stands in for the decoder the reference's graph leaves to the receiving end (the
 * byte stream that /root/reference/src/ceracoder.c:297-339 forwards over SRT).
 *
 * Supported: Baseline-style streams -- CAVLC, frame macroblocks, I/P slices (several per
 * picture allowed), I4x4, I8x8 (transform_size_8x8_flag of an I_NxN macroblock), I16x16, P_L0_16x16, P_L0_L0_16x8, P_L0_L0_8x16, P_8x8 with sub_mb_type P_L0_8x8,
 * P_Skip, quarter-sample luma motion, one
 * reference picture, in-loop filter with all three disable_deblocking_filter_idc values.
 * Anything else sets an error string and returns < 0.
 */
#include "h264_oracle.h"

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------------- numeric tables */
static const uint8_t D_ctok_len[4][68] = {
    {1, 0, 0, 0, 6, 2, 0, 0, 8, 6, 3, 0, 9, 8, 7, 5, 10, 9, 8, 6, 11, 10, 9, 7, 13, 11, 10, 8, 13, 13, 11, 9, 13, 13,
     13, 10, 14, 14, 13, 11, 14, 14, 14, 13, 15, 15, 14, 14, 15, 15, 15, 14, 16, 15, 15, 15, 16, 16, 16, 15, 16, 16,
     16, 16, 16, 16, 16, 16},
    {2, 0, 0, 0, 6, 2, 0, 0, 6, 5, 3, 0, 7, 6, 6, 4, 8, 6, 6, 4, 8, 7, 7, 5, 9, 8, 8, 6, 11, 9, 9, 6, 11, 11,
     11, 7, 12, 11, 11, 9, 12, 12, 12, 11, 12, 12, 12, 11, 13, 13, 13, 12, 13, 13, 13, 13, 13, 14, 13, 13, 14, 14,
     14, 13, 14, 14, 14, 14},
    {4, 0, 0, 0, 6, 4, 0, 0, 6, 5, 4, 0, 6, 5, 5, 4, 7, 5, 5, 4, 7, 5, 5, 4, 7, 6, 6, 4, 7, 6, 6, 4, 8, 7,
     7, 5, 8, 8, 7, 6, 9, 8, 8, 7, 9, 9, 8, 8, 9, 9, 9, 8, 10, 9, 9, 9, 10, 10, 10, 10, 10, 10,
     10, 10, 10, 10, 10, 10},
    {6, 0, 0, 0, 6, 6, 0, 0, 6, 6, 6, 0, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6,
     6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6,
     6, 6, 6, 6, 6, 6}};
static const uint8_t D_ctok_bits[4][68] = {
    {1, 0, 0, 0, 5, 1, 0, 0, 7, 4, 1, 0, 7, 6, 5, 3, 7, 6, 5, 3, 7, 6, 5, 4, 15, 6, 5, 4, 11, 14, 5, 4, 8, 10,
     13, 4, 15, 14, 9, 4, 11, 10, 13, 12, 15, 14, 9, 12, 11, 10, 13, 8, 15, 1, 9, 12, 11, 14, 13, 8, 7, 10,
     9, 12, 4, 6, 5, 8},
    {3, 0, 0, 0, 11, 2, 0, 0, 7, 7, 3, 0, 7, 10, 9, 5, 7, 6, 5, 4, 4, 6, 5, 6, 7, 6, 5, 8, 15, 6, 5, 4, 11, 14,
     13, 4, 15, 10, 9, 4, 11, 14, 13, 12, 8, 10, 9, 8, 15, 14, 13, 12, 11, 10, 9, 12, 7, 11, 6, 8, 9, 8,
     10, 1, 7, 6, 5, 4},
    {15, 0, 0, 0, 15, 14, 0, 0, 11, 15, 13, 0, 8, 12, 14, 12, 15, 10, 11, 11, 11, 8, 9, 10, 9, 14, 13, 9, 8, 10, 9, 8, 15, 14,
     13, 13, 11, 14, 10, 12, 15, 10, 13, 12, 11, 14, 9, 12, 8, 10, 13, 8, 13, 7, 9, 12, 9, 12, 11, 10, 5, 8,
     7, 6, 1, 4, 3, 2},
    {3, 0, 0, 0, 0, 1, 0, 0, 4, 5, 6, 0, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29,
     30, 31, 32, 33, 34, 35, 36, 37, 38, 39, 40, 41, 42, 43, 44, 45, 46, 47, 48, 49, 50, 51, 52, 53, 54, 55, 56, 57,
     58, 59, 60, 61, 62, 63}};
static const uint8_t D_cdc_len[20] = {2, 0, 0, 0, 6, 1, 0, 0, 6, 6, 3, 0, 6, 7, 7, 6, 6, 8, 8, 7};
static const uint8_t D_cdc_bits[20] = {1, 0, 0, 0, 7, 1, 0, 0, 4, 6, 1, 0, 3, 3, 2, 5, 2, 3, 2, 0};
static const uint8_t D_tz_len[15][16] = {
    {1, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 9}, {3, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 6, 6, 6, 6},
    {4, 3, 3, 3, 4, 4, 3, 3, 4, 5, 5, 6, 5, 6},       {5, 3, 4, 4, 3, 3, 3, 4, 3, 4, 5, 5, 5},
    {4, 4, 4, 3, 3, 3, 3, 3, 4, 5, 4, 5},             {6, 5, 3, 3, 3, 3, 3, 3, 4, 3, 6},
    {6, 5, 3, 3, 3, 2, 3, 4, 3, 6},                   {6, 4, 5, 3, 2, 2, 3, 3, 6},
    {6, 6, 4, 2, 2, 3, 2, 5},                         {5, 5, 3, 2, 2, 2, 4},
    {4, 4, 3, 3, 1, 3},                               {4, 4, 2, 1, 3},
    {3, 3, 1, 2},                                     {2, 2, 1},
    {1, 1}};
static const uint8_t D_tz_bits[15][16] = {
    {1, 3, 2, 3, 2, 3, 2, 3, 2, 3, 2, 3, 2, 3, 2, 1}, {7, 6, 5, 4, 3, 5, 4, 3, 2, 3, 2, 3, 2, 1, 0},
    {5, 7, 6, 5, 4, 3, 4, 3, 2, 3, 2, 1, 1, 0},       {3, 7, 5, 4, 6, 5, 4, 3, 3, 2, 2, 1, 0},
    {5, 4, 3, 7, 6, 5, 4, 3, 2, 1, 1, 0},             {1, 1, 7, 6, 5, 4, 3, 2, 1, 1, 0},
    {1, 1, 5, 4, 3, 3, 2, 1, 1, 0},                   {1, 1, 1, 3, 3, 2, 2, 1, 0},
    {1, 0, 1, 3, 2, 1, 1, 1},                         {1, 0, 1, 3, 2, 1, 1},
    {0, 1, 1, 2, 1, 3},                               {0, 1, 1, 1, 1},
    {0, 1, 1, 1},                                     {0, 1, 1},
    {0, 1}};
static const uint8_t D_tzc_len[3][4] = {{1, 2, 3, 3}, {1, 2, 2, 0}, {1, 1, 0, 0}};
static const uint8_t D_tzc_bits[3][4] = {{1, 1, 1, 0}, {1, 1, 0, 0}, {1, 0, 0, 0}};
static const uint8_t D_run_len[7][16] = {{1, 1},
                                         {1, 2, 2},
                                         {2, 2, 2, 2},
                                         {2, 2, 2, 3, 3},
                                         {2, 2, 3, 3, 3, 3},
                                         {2, 3, 3, 3, 3, 3, 3},
                                         {3, 3, 3, 3, 3, 3, 3, 4, 5, 6, 7, 8, 9, 10, 11}};
static const uint8_t D_run_bits[7][16] = {{1, 0},
                                          {1, 1, 0},
                                          {3, 2, 1, 0},
                                          {3, 2, 1, 1, 0},
                                          {3, 2, 3, 2, 1, 0},
                                          {3, 0, 1, 3, 2, 5, 4},
                                          {7, 6, 5, 4, 3, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1}};
static const uint8_t D_cbp_intra[48] = {47, 31, 15, 0,  23, 27, 29, 30, 7,  11, 13, 14, 39, 43, 45, 46,
                                        16, 3,  5,  10, 12, 19, 21, 26, 28, 35, 37, 42, 44, 1,  2,  4,
                                        8,  17, 18, 20, 24, 6,  9,  22, 25, 32, 33, 34, 36, 40, 38, 41};
static const uint8_t D_cbp_inter[48] = {0,  16, 1,  2,  4,  8,  32, 3,  5,  10, 12, 15, 47, 7,  11, 13,
                                        14, 6,  9,  31, 35, 37, 42, 44, 33, 34, 36, 40, 39, 43, 45, 46,
                                        17, 18, 20, 24, 19, 21, 26, 28, 23, 27, 29, 30, 22, 25, 38, 41};
/* inverse zig-zag: scan index -> (x,y) */
static const uint8_t D_zz_x[16] = {0, 1, 0, 0, 1, 2, 3, 2, 1, 0, 1, 2, 3, 3, 2, 3};
static const uint8_t D_zz_y[16] = {0, 0, 1, 2, 1, 0, 0, 1, 2, 3, 3, 2, 1, 2, 3, 3};
/* LevelScale core: rows qP%6; a = positions with both coordinates even, b = both odd, c = mixed */
static const uint8_t D_ls_a[6] = {10, 11, 13, 14, 16, 18};
static const uint8_t D_ls_b[6] = {16, 18, 20, 23, 25, 29};
static const uint8_t D_ls_c[6] = {13, 14, 16, 18, 20, 23};
static const uint8_t D_qpc[22] = {29, 30, 31, 32, 32, 33, 34, 34, 35, 35, 36, 36, 37, 37, 37, 38, 38, 38, 39, 39, 39, 39}; /* qPI 30..51 */
static const uint8_t D_alpha[36] = {4,  4,  5,  6,  7,  8,  9,  10, 12, 13, 15, 17, 20,  22,  25,  28,  32,  36,
                                    40, 45, 50, 56, 63, 71, 80, 90, 101, 113, 127, 144, 162, 182, 203, 226, 255, 255}; /* indexA 16..51 */
static const uint8_t D_beta[36] = {2, 2,  2,  3,  3,  3,  3,  4,  4,  4,  6,  6,  7,  7,  8,  8,  9,  9,
                                   10, 10, 11, 11, 12, 12, 13, 13, 14, 14, 15, 15, 16, 16, 17, 17, 18, 18};
/* tC0 columns for bS 1,2,3; indexA 17..51 (all zero below) */
static const uint8_t D_tc0_1[35] = {0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 4, 4, 4, 5, 6, 6, 7, 8, 9, 10, 11, 13};
static const uint8_t D_tc0_2[35] = {0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 4, 4, 5, 5, 6, 7, 8, 8, 10, 11, 12, 13, 15, 17};
static const uint8_t D_tc0_3[35] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 4, 4, 4, 5, 6, 6, 7, 8, 9, 10, 11, 13, 14, 16, 18, 20, 23, 25};

/* exported for the table cross-check test */
int orc_dec_vlc(int table, int a, int b, int c, int *len, int *bits) {
    *len = 0; *bits = 0;
    switch (table) {
    case 0: if (a < 4 && b < 17 && c < 4) { *len = D_ctok_len[a][4 * b + c]; *bits = D_ctok_bits[a][4 * b + c]; } break;
    case 1: if (b < 5 && c < 4) { *len = D_cdc_len[4 * b + c]; *bits = D_cdc_bits[4 * b + c]; } break;
    case 2: if (b < 15 && c < 16) { *len = D_tz_len[b][c]; *bits = D_tz_bits[b][c]; } break;
    case 3: if (b < 3 && c < 4) { *len = D_tzc_len[b][c]; *bits = D_tzc_bits[b][c]; } break;
    case 4: if (b < 7 && c < 15) { *len = D_run_len[b][c]; *bits = D_run_bits[b][c]; } break;
    default: break;
    }
    return *len != 0;
}
int orc_dec_cbp(int intra, int codenum) { return intra ? D_cbp_intra[codenum] : D_cbp_inter[codenum]; }
int orc_dec_const(int which, int idx) {
    switch (which) {
    case 0: return idx < 16 ? 0 : D_alpha[idx - 16];
    case 1: return idx < 16 ? 0 : D_beta[idx - 16];
    case 2: return idx < 17 ? 0 : D_tc0_1[idx - 17];
    case 3: return idx < 17 ? 0 : D_tc0_2[idx - 17];
    case 4: return idx < 17 ? 0 : D_tc0_3[idx - 17];
    case 5: return idx < 30 ? idx : D_qpc[idx - 30];
    case 6: return D_zz_y[idx] * 4 + D_zz_x[idx];
    case 7: return D_ls_a[idx];
    case 8: return D_ls_b[idx];
    case 9: return D_ls_c[idx];
    default: return -1;
    }
}

/* ---------------------------------------------------------------- state */
typedef struct {
    int8_t kind;      /* -1 not decoded, 0 intra, 1 inter */
    int8_t is_i4;     /* intra 4x4 */
    int8_t qp, qpc;
    int16_t mvx, mvy; /* quarter-sample units (partition 0) */
    int16_t qmv[4][2]; /* the vector of each 8x8 quadrant (raster), whatever the partitioning */
    int8_t part;      /* 0 16x16, 1 16x8, 2 8x16, 3 8x8 */
    int16_t slice;
    uint8_t tc_l[16]; /* TotalCoeff per luma 4x4, raster by*4+bx */
    uint8_t tc_c[2][4];
    uint8_t i4mode[16]; /* raster */
    uint16_t coded;   /* luma 4x4 blocks with non-zero levels, raster bit */
    int8_t dbf_idc, dbf_a, dbf_b;
    int8_t t8;        /* transform_size_8x8_flag */
} dmb_t;

struct orc_dec {
    /* SPS */
    int have_sps, mbw, mbh, crop_l, crop_r, crop_t, crop_b, log2_fn, poc_type, log2_poc_lsb;
    int delta_pic_order_always_zero;
    /* PPS */
    int have_pps, pic_init_qp, cqp_off, dbf_present, constrained_intra, redundant_cnt, num_ref_default, bottom_field_poc, t8_mode;
    uint8_t *cur_y, *cur_uv, *ref_y, *ref_uv;
    int have_ref, pic_open, n_slices;
    dmb_t *mb;
    char err[256];
    /* bit reader over the current RBSP */
    const uint8_t *rb; size_t rb_bits, pos; int rb_fail;
    uint8_t *rbsp; size_t rbsp_cap;
    /* optional capture of the parsed syntax in the encoder's record layout (tests: syntax round trip) */
    orc_mbinfo_t *cap_mbi; int16_t *cap_lev;
};

static int fail(orc_dec_t *d, const char *fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(d->err, sizeof d->err, fmt, ap); va_end(ap);
    return -1;
}
static unsigned rd_bit(orc_dec_t *d) {
    if (d->pos >= d->rb_bits) { d->rb_fail = 1; return 0; }
    unsigned v = (d->rb[d->pos >> 3] >> (7 - (d->pos & 7))) & 1; d->pos++; return v;
}
static unsigned rd_bits(orc_dec_t *d, int n) { unsigned v = 0; while (n--) v = (v << 1) | rd_bit(d); return v; }
static unsigned rd_ue(orc_dec_t *d) {
    int z = 0;
    while (!rd_bit(d)) { if (++z > 32 || d->rb_fail) { d->rb_fail = 1; return 0; } }
    return (unsigned)(((uint64_t)1 << z) - 1 + (z ? rd_bits(d, z) : 0));
}
static int rd_se(orc_dec_t *d) { unsigned k = rd_ue(d); return (k & 1) ? (int)((k + 1) >> 1) : -(int)(k >> 1); }
/* 7.2 more_rbsp_data: true while the read position is before the rbsp_stop_one_bit */
static int more_data(orc_dec_t *d) { return d->pos < d->rb_bits; }

static int set_rbsp(orc_dec_t *d, const uint8_t *nal, size_t n) { /* strip emulation prevention, find stop bit */
    if (d->rbsp_cap < n + 8) { free(d->rbsp); d->rbsp_cap = n + 1024; d->rbsp = (uint8_t *)malloc(d->rbsp_cap); if (!d->rbsp) return -1; }
    size_t o = 0; int z = 0;
    for (size_t i = 0; i < n; i++) {
        if (z >= 2 && nal[i] == 3) { z = 0; continue; }
        d->rbsp[o++] = nal[i];
        z = nal[i] == 0 ? z + 1 : 0;
    }
    while (o && d->rbsp[o - 1] == 0) o--; /* cabac_zero_words / trailing zero bytes */
    if (!o) return -1;
    int last = d->rbsp[o - 1], tz = 0;
    while (!((last >> tz) & 1)) tz++;
    d->rb = d->rbsp; d->rb_bits = o * 8 - (size_t)tz - 1; d->pos = 0; d->rb_fail = 0;
    return 0;
}

/* ---------------------------------------------------------------- parameter sets */
static int parse_sps(orc_dec_t *d) {
    int profile = (int)rd_bits(d, 8); rd_bits(d, 8); rd_bits(d, 8); rd_ue(d);
    if (profile == 100 || profile == 110 || profile == 122 || profile == 244 || profile == 44 || profile == 83 ||
        profile == 86 || profile == 118 || profile == 128) {
        if (rd_ue(d) != 1) return fail(d, "chroma_format_idc != 1");
        if (rd_ue(d) || rd_ue(d)) return fail(d, "bit depth != 8");
        rd_bit(d);
        if (rd_bit(d)) return fail(d, "scaling matrices unsupported");
    }
    d->log2_fn = (int)rd_ue(d) + 4;
    d->poc_type = (int)rd_ue(d);
    if (d->poc_type == 0) d->log2_poc_lsb = (int)rd_ue(d) + 4;
    else if (d->poc_type == 1) {
        d->delta_pic_order_always_zero = (int)rd_bit(d); rd_se(d); rd_se(d);
        unsigned n = rd_ue(d); for (unsigned i = 0; i < n && i < 256; i++) rd_se(d);
    }
    rd_ue(d); rd_bit(d);
    int mbw = (int)rd_ue(d) + 1, mbh = (int)rd_ue(d) + 1;
    if (!rd_bit(d)) return fail(d, "interlace unsupported");
    rd_bit(d);
    d->crop_l = d->crop_r = d->crop_t = d->crop_b = 0;
    if (rd_bit(d)) { d->crop_l = 2 * (int)rd_ue(d); d->crop_r = 2 * (int)rd_ue(d); d->crop_t = 2 * (int)rd_ue(d); d->crop_b = 2 * (int)rd_ue(d); }
    if (d->rb_fail) return fail(d, "SPS truncated");
    if (mbw != d->mbw || mbh != d->mbh || !d->mb) {
        free(d->cur_y); free(d->cur_uv); free(d->ref_y); free(d->ref_uv); free(d->mb);
        d->mbw = mbw; d->mbh = mbh;
        size_t ysz = (size_t)mbw * mbh * 256;
        d->cur_y = (uint8_t *)calloc(ysz, 1); d->cur_uv = (uint8_t *)calloc(ysz / 2, 1);
        d->ref_y = (uint8_t *)calloc(ysz, 1); d->ref_uv = (uint8_t *)calloc(ysz / 2, 1);
        d->mb = (dmb_t *)calloc((size_t)mbw * mbh, sizeof(dmb_t));
        d->have_ref = 0;
        if (!d->cur_y || !d->cur_uv || !d->ref_y || !d->ref_uv || !d->mb) return fail(d, "out of memory");
    }
    d->have_sps = 1;
    return 0;
}
static int parse_pps(orc_dec_t *d) {
    rd_ue(d); rd_ue(d);
    if (rd_bit(d)) return fail(d, "CABAC unsupported");
    d->bottom_field_poc = (int)rd_bit(d);
    if (rd_ue(d)) return fail(d, "slice groups unsupported");
    d->num_ref_default = (int)rd_ue(d) + 1; rd_ue(d);
    if (rd_bit(d)) return fail(d, "weighted prediction unsupported");
    rd_bits(d, 2);
    d->pic_init_qp = 26 + rd_se(d); rd_se(d);
    d->cqp_off = rd_se(d);
    d->dbf_present = (int)rd_bit(d);
    d->constrained_intra = (int)rd_bit(d);
    d->redundant_cnt = (int)rd_bit(d);
    if (d->rb_fail) return fail(d, "PPS truncated");
    d->t8_mode = 0;
    if (more_data(d)) {
        d->t8_mode = (int)rd_bit(d);
        if (rd_bit(d)) return fail(d, "scaling matrices unsupported");
        if (rd_se(d) != d->cqp_off) return fail(d, "second_chroma_qp_index_offset != chroma_qp_index_offset unsupported");
    }
    d->have_pps = 1;
    return 0;
}

/* ---------------------------------------------------------------- CAVLC residual */
static int vlc_read(orc_dec_t *d, const uint8_t *len, const uint8_t *bits, int n) {
    unsigned code = 0;
    for (int l = 1; l <= 16; l++) {
        code = (code << 1) | rd_bit(d);
        for (int i = 0; i < n; i++) if (len[i] == l && bits[i] == code) return i;
    }
    d->rb_fail = 1;
    return -1;
}
/* 9.2: parse one block into coef[0..maxnum-1] (scan order).  Returns TotalCoeff or <0. */
static int residual_block(orc_dec_t *d, int16_t *coef, int maxnum, int nC) {
    memset(coef, 0, sizeof(int16_t) * (size_t)maxnum);
    int tok = nC < 0 ? vlc_read(d, D_cdc_len, D_cdc_bits, 20)
                     : vlc_read(d, D_ctok_len[nC < 2 ? 0 : nC < 4 ? 1 : nC < 8 ? 2 : 3], D_ctok_bits[nC < 2 ? 0 : nC < 4 ? 1 : nC < 8 ? 2 : 3], 68);
    if (tok < 0) return -1;
    int total = tok >> 2, t1 = tok & 3;
    if (total == 0) return 0;
    if (total > maxnum) return -1;
    int level[16];
    int suffix_len = (total > 10 && t1 < 3) ? 1 : 0;
    for (int i = 0; i < total; i++) {
        if (i < t1) { level[i] = rd_bit(d) ? -1 : 1; continue; }
        int prefix = 0;
        while (!rd_bit(d)) { if (++prefix > 15 || d->rb_fail) return -1; }
        int code = (prefix < 15 ? prefix : 15) << suffix_len;
        int ssize = (prefix == 14 && suffix_len == 0) ? 4 : (prefix >= 15 ? prefix - 3 : suffix_len);
        if (ssize) code += (int)rd_bits(d, ssize);
        if (prefix >= 15 && suffix_len == 0) code += 15;
        if (i == t1 && t1 < 3) code += 2;
        level[i] = (code & 1) ? (-code - 1) >> 1 : (code + 2) >> 1;
        if (suffix_len == 0) suffix_len = 1;
        if (abs(level[i]) > (3 << (suffix_len - 1)) && suffix_len < 6) suffix_len++;
    }
    int zeros = 0;
    if (total < maxnum) {
        zeros = maxnum == 4 ? vlc_read(d, D_tzc_len[total - 1], D_tzc_bits[total - 1], 4)
                            : vlc_read(d, D_tz_len[total - 1], D_tz_bits[total - 1], 16);
        if (zeros < 0) return -1;
    }
    int p = zeros + total - 1; /* scan index of the highest-frequency coefficient */
    if (p >= maxnum) return -1;
    for (int i = 0; i < total; i++) {
        coef[p] = (int16_t)level[i];
        if (i + 1 < total) {
            int run = 0;
            if (zeros > 0) {
                int t = (zeros > 7 ? 7 : zeros) - 1;
                run = vlc_read(d, D_run_len[t], D_run_bits[t], 15);
                if (run < 0 || run > zeros) return -1;
                zeros -= run;
            }
            p -= 1 + run;
            if (p < 0) return -1;
        }
    }
    return total;
}

/* ---------------------------------------------------------------- sample helpers */
#define DY(d, x, y) ((d)->cur_y[(size_t)(y) * (d)->mbw * 16 + (x)])
#define DC_(d, x, y, c) ((d)->cur_uv[(size_t)(y) * (d)->mbw * 16 + 2 * (x) + (c)])
static int u8clip(int v) { return v < 0 ? 0 : v > 255 ? 255 : v; }
static int ls_at(int qp, int x, int y) {
    int m = qp % 6;
    if (!(x & 1) && !(y & 1)) return D_ls_a[m];
    if ((x & 1) && (y & 1)) return D_ls_b[m];
    return D_ls_c[m];
}
/* 8.5.12: scaled coefficients c[y][x] -> residual r[y][x] */
static void inv4x4(int c[4][4], int r[4][4]) {
    int t[4][4];
    for (int y = 0; y < 4; y++) {
        int a = c[y][0] + c[y][2], b = c[y][0] - c[y][2];
        int e = (c[y][1] >> 1) - c[y][3], f = c[y][1] + (c[y][3] >> 1);
        t[y][0] = a + f; t[y][1] = b + e; t[y][2] = b - e; t[y][3] = a - f;
    }
    for (int x = 0; x < 4; x++) {
        int a = t[0][x] + t[2][x], b = t[0][x] - t[2][x];
        int e = (t[1][x] >> 1) - t[3][x], f = t[1][x] + (t[3][x] >> 1);
        r[0][x] = (a + f + 32) >> 6; r[1][x] = (b + e + 32) >> 6;
        r[2][x] = (b - e + 32) >> 6; r[3][x] = (a - f + 32) >> 6;
    }
}
/* scan-order levels -> scaled coefficient matrix (8.5.6 + 8.5.12.1); from scan index `first` */
static void scale4x4(const int16_t *lev, int first, int qp, int c[4][4]) {
    memset(c, 0, sizeof(int) * 16);
    for (int k = first; k < 16; k++) {
        int x = D_zz_x[k], y = D_zz_y[k];
        int w = lev[k - first];
        c[y][x] = (w * ls_at(qp, x, y) * 16 << (qp / 6)) >> 4;
    }
}

/* ---------------------------------------------------------------- 8x8 residual (High profile) */
static const uint8_t D_ls8[6][6] = {{20, 18, 32, 19, 25, 24}, {22, 19, 35, 21, 28, 26}, {26, 23, 42, 24, 33, 31},
                                    {28, 25, 45, 26, 35, 33}, {32, 28, 51, 30, 40, 38}, {36, 32, 58, 34, 46, 43}};
static int ls8_class(int x, int y) { /* 8.5.9 */
    if (x % 4 == 0 && y % 4 == 0) return 0;
    if (x % 2 == 1 && y % 2 == 1) return 1;
    if (x % 4 == 2 && y % 4 == 2) return 2;
    if ((x % 4 == 0 && y % 2 == 1) || (x % 2 == 1 && y % 4 == 0)) return 3;
    if ((x % 4 == 0 && y % 4 == 2) || (x % 4 == 2 && y % 4 == 0)) return 4;
    return 5;
}
/* Figure 6-?? 8x8 zig-zag generated by walking the anti-diagonals (not transcribed): scan index -> (x, y) */
static void zz8_xy(int k, int *px, int *py) {
    int x = 0, y = 0;
    for (int i = 0; i < k; i++) {
        if (((x + y) & 1) == 0) { /* moving up-right */
            if (x == 7) y++; else if (y == 0) x++; else { x++; y--; }
        } else {                  /* moving down-left */
            if (y == 7) x++; else if (x == 0) y++; else { x--; y++; }
        }
    }
    *px = x; *py = y;
}
static void inv8_1d(const int *s, int st, int *o, int ot) { /* 8.5.13 */
    int e0 = s[0] + s[4 * st], e1 = -s[3 * st] + s[5 * st] - s[7 * st] - (s[7 * st] >> 1);
    int e2 = s[0] - s[4 * st], e3 = s[st] + s[7 * st] - s[3 * st] - (s[3 * st] >> 1);
    int e4 = (s[2 * st] >> 1) - s[6 * st], e5 = -s[st] + s[7 * st] + s[5 * st] + (s[5 * st] >> 1);
    int e6 = s[2 * st] + (s[6 * st] >> 1), e7 = s[3 * st] + s[5 * st] + s[st] + (s[st] >> 1);
    int f0 = e0 + e6, f1 = e1 + (e7 >> 2), f2 = e2 + e4, f3 = e3 + (e5 >> 2);
    int f4 = e2 - e4, f5 = (e3 >> 2) - e5, f6 = e0 - e6, f7 = e7 - (e1 >> 2);
    o[0] = f0 + f7; o[ot] = f2 + f5; o[2 * ot] = f4 + f3; o[3 * ot] = f6 + f1;
    o[4 * ot] = f6 - f1; o[5 * ot] = f4 - f3; o[6 * ot] = f2 - f5; o[7 * ot] = f0 - f7;
}
int orc_dec_zz8(int k) { int x, y; zz8_xy(k, &x, &y); return y * 8 + x; }

/* ---------------------------------------------------------------- neighbour availability */
static int mb_avail(const orc_dec_t *d, int mx, int my, int slice) {
    if (mx < 0 || my < 0 || mx >= d->mbw || my >= d->mbh) return 0;
    const dmb_t *m = &d->mb[my * d->mbw + mx];
    return m->kind >= 0 && m->slice == slice;
}
/* TotalCoeff context, 9.2.1.  plane: 0 luma, 1 Cb, 2 Cr; (bx,by) block coords inside the MB */
static int ctx_nC(const orc_dec_t *d, int mx, int my, int slice, int plane, int bx, int by) {
    int na = -1, nb = -1, dim = plane ? 2 : 4;
    const dmb_t *cur = &d->mb[my * d->mbw + mx];
    if (bx > 0) na = plane ? cur->tc_c[plane - 1][by * 2 + bx - 1] : cur->tc_l[by * 4 + bx - 1];
    else if (mb_avail(d, mx - 1, my, slice)) {
        const dmb_t *m = cur - 1;
        na = plane ? m->tc_c[plane - 1][by * 2 + dim - 1] : m->tc_l[by * 4 + dim - 1];
    }
    if (by > 0) nb = plane ? cur->tc_c[plane - 1][(by - 1) * 2 + bx] : cur->tc_l[(by - 1) * 4 + bx];
    else if (mb_avail(d, mx, my - 1, slice)) {
        const dmb_t *m = cur - d->mbw;
        nb = plane ? m->tc_c[plane - 1][(dim - 1) * 2 + bx] : m->tc_l[(dim - 1) * 4 + bx];
    }
    if (na >= 0 && nb >= 0) return (na + nb + 1) >> 1;
    if (na >= 0) return na;
    if (nb >= 0) return nb;
    return 0;
}

/* ---------------------------------------------------------------- intra prediction */
static int intra_nb_ok(const orc_dec_t *d, int mx, int my, int slice) { /* usable for intra prediction */
    if (!mb_avail(d, mx, my, slice)) return 0;
    if (d->constrained_intra && d->mb[my * d->mbw + mx].kind != 0) return 0;
    return 1;
}
static void pred_i16(orc_dec_t *d, int mx, int my, int slice, int mode, int *err) {
    int X = mx * 16, Y = my * 16;
    int up = intra_nb_ok(d, mx, my - 1, slice), lf = intra_nb_ok(d, mx - 1, my, slice), ul = intra_nb_ok(d, mx - 1, my - 1, slice);
    int top[16], left[16], corner = 0;
    for (int i = 0; i < 16; i++) { top[i] = up ? DY(d, X + i, Y - 1) : 0; left[i] = lf ? DY(d, X - 1, Y + i) : 0; }
    if (ul) corner = DY(d, X - 1, Y - 1);
    if ((mode == 0 && !up) || (mode == 1 && !lf) || (mode == 3 && !(up && lf && ul))) { *err = 1; return; }
    for (int y = 0; y < 16; y++)
        for (int x = 0; x < 16; x++) {
            int v;
            if (mode == 0) v = top[x];
            else if (mode == 1) v = left[y];
            else if (mode == 2) {
                int s = 0;
                if (up) for (int i = 0; i < 16; i++) s += top[i];
                if (lf) for (int i = 0; i < 16; i++) s += left[i];
                v = (up && lf) ? (s + 16) >> 5 : (up || lf) ? (s + 8) >> 4 : 128;
            } else {
                int H = 0, V = 0;
                for (int k = 1; k <= 8; k++) {
                    H += k * (top[7 + k] - (k == 8 ? corner : top[7 - k]));
                    V += k * (left[7 + k] - (k == 8 ? corner : left[7 - k]));
                }
                int a = 16 * (left[15] + top[15]), b = (5 * H + 32) >> 6, c = (5 * V + 32) >> 6;
                v = u8clip((a + b * (x - 7) + c * (y - 7) + 16) >> 5);
            }
            DY(d, X + x, Y + y) = (uint8_t)v;
        }
}
static void pred_chroma8(orc_dec_t *d, int mx, int my, int slice, int mode, int *err) {
    int X = mx * 8, Y = my * 8;
    int up = intra_nb_ok(d, mx, my - 1, slice), lf = intra_nb_ok(d, mx - 1, my, slice), ul = intra_nb_ok(d, mx - 1, my - 1, slice);
    if ((mode == 1 && !lf) || (mode == 2 && !up) || (mode == 3 && !(up && lf && ul))) { *err = 1; return; }
    for (int c = 0; c < 2; c++) {
        int top[8], left[8], corner = ul ? DC_(d, X - 1, Y - 1, c) : 0;
        for (int i = 0; i < 8; i++) { top[i] = up ? DC_(d, X + i, Y - 1, c) : 0; left[i] = lf ? DC_(d, X - 1, Y + i, c) : 0; }
        for (int y = 0; y < 8; y++)
            for (int x = 0; x < 8; x++) {
                int v;
                if (mode == 0) {
                    int qx = x >> 2, qy = y >> 2, st = 0, sl = 0;
                    for (int i = 0; i < 4; i++) { st += top[qx * 4 + i]; sl += left[qy * 4 + i]; }
                    int use_t = up, use_l = lf;
                    if (qx == 1 && qy == 0 && up) use_l = 0;      /* top-right quadrant prefers the row above */
                    if (qx == 0 && qy == 1 && lf) use_t = 0;      /* bottom-left quadrant prefers the left column */
                    if (use_t && use_l) v = (st + sl + 4) >> 3;
                    else if (use_t) v = (st + 2) >> 2;
                    else if (use_l) v = (sl + 2) >> 2;
                    else v = 128;
                } else if (mode == 1) v = left[y];
                else if (mode == 2) v = top[x];
                else {
                    int H = 0, V = 0;
                    for (int k = 1; k <= 4; k++) {
                        H += k * (top[3 + k] - (k == 4 ? corner : top[3 - k]));
                        V += k * (left[3 + k] - (k == 4 ? corner : left[3 - k]));
                    }
                    int a = 16 * (left[7] + top[7]), b = (34 * H + 32) >> 6, cc = (34 * V + 32) >> 6;
                    v = u8clip((a + b * (x - 3) + cc * (y - 3) + 16) >> 5);
                }
                DC_(d, X + x, Y + y, c) = (uint8_t)v;
            }
    }
}
/* 8.3.1.2 Intra_4x4 sample prediction for the block at pixel (X,Y); edge[] = p[-1..7,-1] then left */
static void pred_i4(orc_dec_t *d, int X, int Y, int mode, int has_up, int has_left, int has_ul, int has_ur, int *err) {
    int t[8], l[4], c = has_ul ? DY(d, X - 1, Y - 1) : 0;
    for (int i = 0; i < 4; i++) { t[i] = has_up ? DY(d, X + i, Y - 1) : 0; l[i] = has_left ? DY(d, X - 1, Y + i) : 0; }
    for (int i = 4; i < 8; i++) t[i] = has_ur ? DY(d, X + i, Y - 1) : t[3];
    int need_up = mode == 0 || mode == 3 || mode == 7, need_left = mode == 1 || mode == 8;
    int need_all = mode == 4 || mode == 5 || mode == 6;
    if ((need_up && !has_up) || (need_left && !has_left) || (need_all && !(has_up && has_left && has_ul))) { *err = 1; return; }
    for (int y = 0; y < 4; y++)
        for (int x = 0; x < 4; x++) {
            int v = 0;
            switch (mode) {
            case 0: v = t[x]; break;
            case 1: v = l[y]; break;
            case 2:
                if (has_up && has_left) v = (t[0] + t[1] + t[2] + t[3] + l[0] + l[1] + l[2] + l[3] + 4) >> 3;
                else if (has_left) v = (l[0] + l[1] + l[2] + l[3] + 2) >> 2;
                else if (has_up) v = (t[0] + t[1] + t[2] + t[3] + 2) >> 2;
                else v = 128;
                break;
            case 3: /* diagonal down-left */
                v = (x == 3 && y == 3) ? (t[6] + 3 * t[7] + 2) >> 2 : (t[x + y] + 2 * t[x + y + 1] + t[x + y + 2] + 2) >> 2;
                break;
            case 4: { /* diagonal down-right */
                #define E4(i) ((i) < 0 ? l[-(i) - 1] : (i) == 0 ? c : t[(i) - 1]) /* ... l1 l0 c t0 t1 ... */
                int k = x - y;
                v = (E4(k - 1) + 2 * E4(k) + E4(k + 1) + 2) >> 2;
                break; }
            case 5: { /* vertical-right */
                int z = 2 * x - y;
                if (z >= 0 && !(z & 1)) v = (E4(x - (y >> 1)) + E4(x - (y >> 1) + 1) + 1) >> 1;
                else if (z >= 0) v = (E4(x - (y >> 1) - 1) + 2 * E4(x - (y >> 1)) + E4(x - (y >> 1) + 1) + 2) >> 2;
                else if (z == -1) v = (l[0] + 2 * c + t[0] + 2) >> 2;
                else v = (E4(-y) + 2 * E4(-y + 1) + E4(-y + 2) + 2) >> 2; /* p[-1,y-1], p[-1,y-2], p[-1,y-3] */
                break; }
            case 6: { /* horizontal-down */
                int z = 2 * y - x;
                if (z >= 0 && !(z & 1)) v = (E4(-(y - (x >> 1))) + E4(-(y - (x >> 1)) - 1) + 1) >> 1;
                else if (z >= 0) v = (E4(-(y - (x >> 1)) + 1) + 2 * E4(-(y - (x >> 1))) + E4(-(y - (x >> 1)) - 1) + 2) >> 2;
                else if (z == -1) v = (l[0] + 2 * c + t[0] + 2) >> 2;
                else v = (E4(x - 1 + 1) + 2 * E4(x - 2 + 1) + E4(x - 3 + 1) + 2) >> 2; /* p[x-1,-1],p[x-2,-1],p[x-3,-1] */
                break; }
            case 7: /* vertical-left */
                v = !(y & 1) ? (t[x + (y >> 1)] + t[x + (y >> 1) + 1] + 1) >> 1
                             : (t[x + (y >> 1)] + 2 * t[x + (y >> 1) + 1] + t[x + (y >> 1) + 2] + 2) >> 2;
                break;
            case 8: { /* horizontal-up */
                int z = x + 2 * y;
                if (z > 5) v = l[3];
                else if (z == 5) v = (l[2] + 3 * l[3] + 2) >> 2;
                else if (!(z & 1)) v = (l[y + (x >> 1)] + l[y + (x >> 1) + 1] + 1) >> 1;
                else v = (l[y + (x >> 1)] + 2 * l[y + (x >> 1) + 1] + l[y + (x >> 1) + 2] + 2) >> 2;
                break; }
            default: *err = 1; return;
            }
            DY(d, X + x, Y + y) = (uint8_t)v;
        }
}

/* 8.3.2.2: Intra_8x8 prediction of the block at (X, Y); the neighbouring samples are filtered first (8.3.2.2.1) */
static void pred_i8(orc_dec_t *d, int X, int Y, int mode, int has_up, int has_left, int has_ul, int has_ur, int *err) {
    int raw_t[16], raw_l[8], corner = 0, pt[16], pl[8], pc = 0; /* p[x,-1], p[-1,y], p[-1,-1] and their filtered forms */
    if (has_ul) corner = DY(d, X - 1, Y - 1);
    for (int x = 0; x < 16; x++) raw_t[x] = has_up ? (x < 8 || has_ur ? DY(d, X + x, Y - 1) : DY(d, X + 7, Y - 1)) : 0;
    for (int y = 0; y < 8; y++) raw_l[y] = has_left ? DY(d, X - 1, Y + y) : 0;
    memset(pt, 0, sizeof pt); memset(pl, 0, sizeof pl);
    if (has_up) {
        pt[0] = has_ul ? (corner + 2 * raw_t[0] + raw_t[1] + 2) >> 2 : (3 * raw_t[0] + raw_t[1] + 2) >> 2;
        for (int x = 1; x < 15; x++) pt[x] = (raw_t[x - 1] + 2 * raw_t[x] + raw_t[x + 1] + 2) >> 2;
        pt[15] = (raw_t[14] + 3 * raw_t[15] + 2) >> 2;
    }
    if (has_left) {
        pl[0] = has_ul ? (corner + 2 * raw_l[0] + raw_l[1] + 2) >> 2 : (3 * raw_l[0] + raw_l[1] + 2) >> 2;
        for (int y = 1; y < 7; y++) pl[y] = (raw_l[y - 1] + 2 * raw_l[y] + raw_l[y + 1] + 2) >> 2;
        pl[7] = (raw_l[6] + 3 * raw_l[7] + 2) >> 2;
    }
    if (has_ul) {
        if (has_up && has_left) pc = (raw_t[0] + 2 * corner + raw_l[0] + 2) >> 2;
        else if (has_up) pc = (3 * corner + raw_t[0] + 2) >> 2;
        else if (has_left) pc = (3 * corner + raw_l[0] + 2) >> 2;
        else pc = corner;
    }
    const int need_up = mode == 0 || mode == 3 || mode == 7, need_left = mode == 1 || mode == 8, need_all = mode >= 4 && mode <= 6;
    if ((need_up && !has_up) || (need_left && !has_left) || (need_all && !(has_up && has_left && has_ul)) || mode > 8) { *err = 1; return; }
#define PT(x) ((x) < 0 ? pc : pt[x])  /* p'[x,-1], x = -1 .. 15 */
#define PL(y) ((y) < 0 ? pc : pl[y])  /* p'[-1,y], y = -1 .. 7 */
    for (int y = 0; y < 8; y++)
        for (int x = 0; x < 8; x++) {
            int v;
            switch (mode) {
            case 0: v = PT(x); break;
            case 1: v = PL(y); break;
            case 2: {
                int s = 0;
                if (has_up) for (int i = 0; i < 8; i++) s += pt[i];
                if (has_left) for (int i = 0; i < 8; i++) s += pl[i];
                v = has_up && has_left ? (s + 8) >> 4 : has_up || has_left ? (s + 4) >> 3 : 128;
                break; }
            case 3: v = x == 7 && y == 7 ? (PT(14) + 3 * PT(15) + 2) >> 2 : (PT(x + y) + 2 * PT(x + y + 1) + PT(x + y + 2) + 2) >> 2; break;
            case 4:
                if (x > y) v = (PT(x - y - 2) + 2 * PT(x - y - 1) + PT(x - y) + 2) >> 2;
                else if (x < y) v = (PL(y - x - 2) + 2 * PL(y - x - 1) + PL(y - x) + 2) >> 2;
                else v = (PT(0) + 2 * pc + PL(0) + 2) >> 2;
                break;
            case 5: {
                int z = 2 * x - y;
                if (z >= 0 && (z & 1) == 0) v = (PT(x - (y >> 1) - 1) + PT(x - (y >> 1)) + 1) >> 1;
                else if (z >= 0) v = (PT(x - (y >> 1) - 2) + 2 * PT(x - (y >> 1) - 1) + PT(x - (y >> 1)) + 2) >> 2;
                else if (z == -1) v = (PL(0) + 2 * pc + PT(0) + 2) >> 2;
                else v = (PL(y - 2 * x - 1) + 2 * PL(y - 2 * x - 2) + PL(y - 2 * x - 3) + 2) >> 2;
                break; }
            case 6: {
                int z = 2 * y - x;
                if (z >= 0 && (z & 1) == 0) v = (PL(y - (x >> 1) - 1) + PL(y - (x >> 1)) + 1) >> 1;
                else if (z >= 0) v = (PL(y - (x >> 1) - 2) + 2 * PL(y - (x >> 1) - 1) + PL(y - (x >> 1)) + 2) >> 2;
                else if (z == -1) v = (PL(0) + 2 * pc + PT(0) + 2) >> 2;
                else v = (PT(x - 2 * y - 1) + 2 * PT(x - 2 * y - 2) + PT(x - 2 * y - 3) + 2) >> 2;
                break; }
            case 7:
                v = (y & 1) == 0 ? (PT(x + (y >> 1)) + PT(x + (y >> 1) + 1) + 1) >> 1 : (PT(x + (y >> 1)) + 2 * PT(x + (y >> 1) + 1) + PT(x + (y >> 1) + 2) + 2) >> 2;
                break;
            default: {
                int z = x + 2 * y;
                if (z > 13) v = pl[7];
                else if (z == 13) v = (pl[6] + 3 * pl[7] + 2) >> 2;
                else if ((z & 1) == 0) v = (PL(y + (x >> 1)) + PL(y + (x >> 1) + 1) + 1) >> 1;
                else v = (PL(y + (x >> 1)) + 2 * PL(y + (x >> 1) + 1) + PL(y + (x >> 1) + 2) + 2) >> 2;
                break; }
            }
            DY(d, X + x, Y + y) = (uint8_t)v;
        }
#undef PT
#undef PL
}

/* ---------------------------------------------------------------- inter prediction (8.4.2.2) */
static int rpx(const orc_dec_t *d, int x, int y) {
    int W = d->mbw * 16, H = d->mbh * 16;
    x = x < 0 ? 0 : x >= W ? W - 1 : x; y = y < 0 ? 0 : y >= H ? H - 1 : y;
    return d->ref_y[(size_t)y * W + x];
}
static int tap6(int a, int b, int c, int e, int f, int g) { return a - 5 * b + 20 * c + 20 * e - 5 * f + g; }
static int hb1(const orc_dec_t *d, int x, int y) { return tap6(rpx(d, x - 2, y), rpx(d, x - 1, y), rpx(d, x, y), rpx(d, x + 1, y), rpx(d, x + 2, y), rpx(d, x + 3, y)); }
static int vh1(const orc_dec_t *d, int x, int y) { return tap6(rpx(d, x, y - 2), rpx(d, x, y - 1), rpx(d, x, y), rpx(d, x, y + 1), rpx(d, x, y + 2), rpx(d, x, y + 3)); }
static int luma_sample(const orc_dec_t *d, int x, int y, int fx, int fy) {
    if (!fx && !fy) return rpx(d, x, y);
    int G = rpx(d, x, y), Hh = rpx(d, x + 1, y), M = rpx(d, x, y + 1);
    int b = u8clip((hb1(d, x, y) + 16) >> 5), h = u8clip((vh1(d, x, y) + 16) >> 5);
    if (fy == 0) return fx == 1 ? (G + b + 1) >> 1 : fx == 2 ? b : (Hh + b + 1) >> 1;
    if (fx == 0) return fy == 1 ? (G + h + 1) >> 1 : fy == 2 ? h : (M + h + 1) >> 1;
    int s = u8clip((hb1(d, x, y + 1) + 16) >> 5), m = u8clip((vh1(d, x + 1, y) + 16) >> 5);
    if ((fx & 1) && (fy & 1)) { int p = fy == 1 ? b : s, q = fx == 1 ? h : m; return (p + q + 1) >> 1; }
    int j = u8clip((tap6(hb1(d, x, y - 2), hb1(d, x, y - 1), hb1(d, x, y), hb1(d, x, y + 1), hb1(d, x, y + 2), hb1(d, x, y + 3)) + 512) >> 10);
    if (fx == 2 && fy == 2) return j;
    if (fx == 2) return fy == 1 ? (b + j + 1) >> 1 : (j + s + 1) >> 1;
    return fx == 1 ? (h + j + 1) >> 1 : (j + m + 1) >> 1; /* fy == 2 */
}
static void inter_pred_part(orc_dec_t *d, int mx, int my, int x0, int y0, int w, int h, int mvx, int mvy) { /* one partition: luma w x h at (x0, y0) of the macroblock */
    int X = mx * 16, Y = my * 16;
    for (int y = y0; y < y0 + h; y++)
        for (int x = x0; x < x0 + w; x++)
            DY(d, X + x, Y + y) = (uint8_t)luma_sample(d, X + x + (mvx >> 2), Y + y + (mvy >> 2), mvx & 3, mvy & 3);
    int cw = d->mbw * 8, ch = d->mbh * 8, xf = mvx & 7, yf = mvy & 7;
    for (int c = 0; c < 2; c++)
        for (int y = y0 / 2; y < (y0 + h) / 2; y++)
            for (int x = x0 / 2; x < (x0 + w) / 2; x++) {
                int xa = mx * 8 + x + (mvx >> 3), ya = my * 8 + y + (mvy >> 3), xb = xa + 1, yb = ya + 1;
                xa = xa < 0 ? 0 : xa >= cw ? cw - 1 : xa; xb = xb < 0 ? 0 : xb >= cw ? cw - 1 : xb;
                ya = ya < 0 ? 0 : ya >= ch ? ch - 1 : ya; yb = yb < 0 ? 0 : yb >= ch ? ch - 1 : yb;
                const uint8_t *r = d->ref_uv; size_t st = (size_t)d->mbw * 16;
                int A = r[ya * st + 2 * xa + c], B = r[ya * st + 2 * xb + c], C = r[yb * st + 2 * xa + c], D = r[yb * st + 2 * xb + c];
                DC_(d, mx * 8 + x, my * 8 + y, c) = (uint8_t)(((8 - xf) * (8 - yf) * A + xf * (8 - yf) * B + (8 - xf) * yf * C + xf * yf * D + 32) >> 6);
            }
}
static void inter_pred_mb(orc_dec_t *d, int mx, int my, int mvx, int mvy) { inter_pred_part(d, mx, my, 0, 0, 16, 16, mvx, mvy); }
/* 6.4.11.7 / 8.4.1.3.2: motion data of the 8x8 block that covers luma sample (X, Y) of the picture, as a neighbour of a partition of macroblock
 * (mx, my) whose quadrants in `done` (bit q) are decoded already.  Macroblocks after the current one in decoding order are not available. */
static void nb_blk(const orc_dec_t *d, int mx, int my, int slice, unsigned done, int X, int Y, int *avail, int *ref, int *vx, int *vy) {
    *avail = 0; *ref = -1; *vx = *vy = 0;
    if (X < 0 || Y < 0 || X >= d->mbw * 16 || Y >= d->mbh * 16) return;
    const int nx = X >> 4, ny = Y >> 4, q = ((Y & 15) >> 3) * 2 + ((X & 15) >> 3);
    const dmb_t *n = &d->mb[ny * d->mbw + nx];
    if (nx == mx && ny == my) { if (!((done >> q) & 1)) return; *avail = 1; *ref = 0; *vx = n->qmv[q][0]; *vy = n->qmv[q][1]; return; }
    if (!mb_avail(d, nx, ny, slice)) return;
    *avail = 1;
    if (n->kind == 1) { *ref = 0; *vx = n->qmv[q][0]; *vy = n->qmv[q][1]; }
}
static int med(int a, int b, int c) { return a > b ? (b > c ? b : (a > c ? c : a)) : (a > c ? a : (b > c ? c : b)); }
/* 8.4.1.3 for the partition (x0, y0, w, h) of shape `part` (0 16x16, 1 16x8, 2 8x16, 3 8x8), index idx; skip: 8.4.1.1's inference for P_Skip */
static void predict_part(const orc_dec_t *d, int mx, int my, int slice, unsigned done, int part, int idx, int x0, int y0, int w, int skip, int *px, int *py) {
    const int X = mx * 16 + x0, Y = my * 16 + y0;
    int aA, rA, ax, ay, aB, rB, bx, by, aC, rC, cx, cy;
    nb_blk(d, mx, my, slice, done, X - 1, Y, &aA, &rA, &ax, &ay);
    nb_blk(d, mx, my, slice, done, X, Y - 1, &aB, &rB, &bx, &by);
    nb_blk(d, mx, my, slice, done, X + w, Y - 1, &aC, &rC, &cx, &cy);
    if (!aC) nb_blk(d, mx, my, slice, done, X - 1, Y - 1, &aC, &rC, &cx, &cy);
    if (skip && (!aA || !aB || (rA == 0 && !ax && !ay) || (rB == 0 && !bx && !by))) { *px = *py = 0; return; }
    if (part == 1 && idx == 0 && rB == 0) { *px = bx; *py = by; return; } /* directional predictors (refIdx is 0 everywhere) */
    if (part == 1 && idx == 1 && rA == 0) { *px = ax; *py = ay; return; }
    if (part == 2 && idx == 0 && rA == 0) { *px = ax; *py = ay; return; }
    if (part == 2 && idx == 1 && rC == 0) { *px = cx; *py = cy; return; }
    if (!aB && !aC && aA) { rB = rC = rA; bx = cx = ax; by = cy = ay; }
    int hits = (rA == 0) + (rB == 0) + (rC == 0);
    if (hits == 1) {
        if (rA == 0) { *px = ax; *py = ay; } else if (rB == 0) { *px = bx; *py = by; } else { *px = cx; *py = cy; }
    } else { *px = med(ax, bx, cx); *py = med(ay, by, cy); }
}
static void predict_mv(const orc_dec_t *d, int mx, int my, int slice, int skip, int *px, int *py) { predict_part(d, mx, my, slice, 0, 0, 0, 0, 0, 16, skip, px, py); }
static void set_qmv(dmb_t *m, int x0, int y0, int w, int h, int vx, int vy) {
    for (int q = 0; q < 4; q++) { const int qx = (q & 1) * 8, qy = (q >> 1) * 8; if (qx >= x0 && qx < x0 + w && qy >= y0 && qy < y0 + h) { m->qmv[q][0] = (int16_t)vx; m->qmv[q][1] = (int16_t)vy; } }
}

/* ---------------------------------------------------------------- macroblock decode */
static int chroma_qp(const orc_dec_t *d, int qp) {
    int i = qp + d->cqp_off; i = i < 0 ? 0 : i > 51 ? 51 : i;
    return i < 30 ? i : D_qpc[i - 30];
}
static void add_res4(orc_dec_t *d, int X, int Y, int c[4][4]) {
    int r[4][4]; inv4x4(c, r);
    for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) DY(d, X + x, Y + y) = (uint8_t)u8clip(DY(d, X + x, Y + y) + r[y][x]);
}
static const uint8_t D_blkx[16] = {0, 1, 0, 1, 2, 3, 2, 3, 0, 1, 0, 1, 2, 3, 2, 3}; /* blkIdx -> 4x4 column */
static const uint8_t D_blky[16] = {0, 0, 1, 1, 0, 0, 1, 1, 2, 2, 3, 3, 2, 2, 3, 3};

/* capture (tests only): where the parsed syntax of macroblock (mx, my) goes, in the layout of orc_mbinfo_t / ORC_L_* */
static orc_mbinfo_t *cap_rec(orc_dec_t *d, int mx, int my) { return d->cap_mbi ? &d->cap_mbi[my * d->mbw + mx] : NULL; }
static int16_t *cap_levels(orc_dec_t *d, int mx, int my) { return d->cap_lev ? d->cap_lev + (size_t)(my * d->mbw + mx) * ORC_LEVELS_PER_MB : NULL; }
static int any_nz(const int16_t *l, int n) { for (int i = 0; i < n; i++) if (l[i]) return 1; return 0; }

static int decode_chroma_residual(orc_dec_t *d, int mx, int my, int slice, int cbp_chroma, int qpc) {
    dmb_t *m = &d->mb[my * d->mbw + mx];
    int16_t dc[2][4] = {{0}}, ac[2][4][15];
    memset(ac, 0, sizeof ac);
    if (cbp_chroma) for (int c = 0; c < 2; c++) if (residual_block(d, dc[c], 4, -1) < 0) return fail(d, "chroma DC residual");
    if (cbp_chroma == 2)
        for (int c = 0; c < 2; c++)
            for (int b = 0; b < 4; b++) {
                int n = residual_block(d, ac[c][b], 15, ctx_nC(d, mx, my, slice, 1 + c, b & 1, b >> 1));
                if (n < 0) return fail(d, "chroma AC residual");
                m->tc_c[c][b] = (uint8_t)n;
            }
    if (d->cap_mbi && d->cap_lev) {
        orc_mbinfo_t *r = cap_rec(d, mx, my); int16_t *cl = cap_levels(d, mx, my);
        for (int c = 0; c < 2; c++) {
            memcpy(cl + ORC_L_CDC + 4 * c, dc[c], 8);
            if (any_nz(dc[c], 4)) r->nzmask |= c ? ORC_NZ_CRDC : ORC_NZ_CBDC;
            for (int b = 0; b < 4; b++) {
                memcpy(cl + ORC_L_CAC + (4 * c + b) * 16 + 1, ac[c][b], 30);
                if (any_nz(ac[c][b], 15)) r->nzmask |= 1u << (16 + 4 * c + b);
            }
        }
    }
    for (int c = 0; c < 2; c++) {
        /* 8.5.11: c = [[dc0 dc1][dc2 dc3]], f = A c A */
        int f[4] = {dc[c][0] + dc[c][1] + dc[c][2] + dc[c][3], dc[c][0] - dc[c][1] + dc[c][2] - dc[c][3],
                    dc[c][0] + dc[c][1] - dc[c][2] - dc[c][3], dc[c][0] - dc[c][1] - dc[c][2] + dc[c][3]};
        for (int b = 0; b < 4; b++) {
            int co[4][4];
            scale4x4(ac[c][b], 1, qpc, co);
            co[0][0] = ((f[b] * D_ls_a[qpc % 6] * 16) << (qpc / 6)) >> 5;
            int r[4][4]; inv4x4(co, r);
            int X = mx * 8 + (b & 1) * 4, Y = my * 8 + (b >> 1) * 4;
            for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) DC_(d, X + x, Y + y, c) = (uint8_t)u8clip(DC_(d, X + x, Y + y, c) + r[y][x]);
        }
    }
    return 0;
}

static int decode_mb(orc_dec_t *d, int mx, int my, int slice, int is_p, int skipped, int *qp) {
    dmb_t *m = &d->mb[my * d->mbw + mx];
    memset(m->tc_l, 0, 16); memset(m->tc_c, 0, 8); memset(m->i4mode, 2, 16);
    m->slice = (int16_t)slice; m->coded = 0; m->is_i4 = 0; m->mvx = m->mvy = 0; m->t8 = 0;
    orc_mbinfo_t *cr = cap_rec(d, mx, my); int16_t *cl = cap_levels(d, mx, my);
    if (cr) memset(cr, 0, sizeof *cr);
    if (cl) memset(cl, 0, ORC_LEVELS_PER_MB * sizeof(int16_t));
    if (skipped) {
        int px, py; predict_mv(d, mx, my, slice, 1, &px, &py);
        m->mvx = (int16_t)px; m->mvy = (int16_t)py; m->part = 0; set_qmv(m, 0, 0, 16, 16, px, py);
        inter_pred_mb(d, mx, my, px, py);
        m->kind = 1; m->qp = (int8_t)*qp; m->qpc = (int8_t)chroma_qp(d, *qp);
        if (cr) { cr->mb_type = 1; cr->mvx = m->mvx; cr->mvy = m->mvy; cr->qp = (uint8_t)*qp; }
        return 0;
    }
    int t = (int)rd_ue(d);
    int intra = 1, i16 = 0, i16mode = 0, cbp = 0, cap_cmode = 0;
    if (is_p) { if (t < 5) intra = 0; else t -= 5; }
    if (!intra) {
        if (t > 3) return fail(d, "P macroblock type %d unsupported", t);
        if (d->num_ref_default > 1) return fail(d, "multiple references unsupported");
        static const int8_t geo[4][4][4] = { /* shape -> partitions (x0, y0, w, h) */
            {{0, 0, 16, 16}}, {{0, 0, 16, 8}, {0, 8, 16, 8}}, {{0, 0, 8, 16}, {8, 0, 8, 16}}, {{0, 0, 8, 8}, {8, 0, 8, 8}, {0, 8, 8, 8}, {8, 8, 8, 8}}};
        const int np = t == 0 ? 1 : t == 3 ? 4 : 2;
        if (t == 3) for (int i = 0; i < 4; i++) if (rd_ue(d) != 0) return fail(d, "sub_mb_type other than P_L0_8x8 unsupported");
        m->part = (int8_t)t; m->kind = 1; /* (neighbour derivation inside the macroblock goes by `done`) */
        unsigned done = 0;
        int pv[4][2];
        for (int i = 0; i < np; i++) { /* 7.3.5.1 / 7.3.5.2: the vector differences of the partitions in order (ref_idx is not sent: one reference) */
            const int8_t *g = geo[t][i];
            int px, py; predict_part(d, mx, my, slice, done, t, i, g[0], g[1], g[2], 0, &px, &py);
            int dx = rd_se(d), dy = rd_se(d);
            pv[i][0] = px + dx; pv[i][1] = py + dy;
            set_qmv(m, g[0], g[1], g[2], g[3], pv[i][0], pv[i][1]);
            for (int q = 0; q < 4; q++) { const int qx = (q & 1) * 8, qy = (q >> 1) * 8; if (qx >= g[0] && qx < g[0] + g[2] && qy >= g[1] && qy < g[1] + g[3]) done |= 1u << q; }
        }
        m->mvx = (int16_t)pv[0][0]; m->mvy = (int16_t)pv[0][1];
        for (int i = 0; i < np; i++) { const int8_t *g = geo[t][i]; inter_pred_part(d, mx, my, g[0], g[1], g[2], g[3], pv[i][0], pv[i][1]); }
        unsigned k = rd_ue(d); if (k > 47) return fail(d, "cbp codeNum %u", k);
        cbp = D_cbp_inter[k];
        if (d->t8_mode && (cbp & 15)) m->t8 = (int8_t)rd_bit(d);
    } else {
        m->kind = 0;
        if (t == 0) { m->is_i4 = 1; if (d->t8_mode) m->t8 = (int8_t)rd_bit(d); } /* transform_size_8x8_flag: Intra_8x8 (is_i4 with t8) or Intra_4x4 */
        else if (t <= 24) { i16 = 1; i16mode = (t - 1) & 3; cbp = (((t - 1) >> 2) % 3) << 4 | (t > 12 ? 15 : 0); }
        else return fail(d, "I_PCM / mb_type %d unsupported", t);
        if (m->is_i4 && m->t8) {
            /* 8.3.2.1: the four Intra8x8PredMode, each predicted from the blocks left and above -- an Intra_4x4 neighbour contributes the mode of its 4x4
             * block 4 n + 1 (left) / 4 n + 2 (above), an Intra_8x8 one its own (i4mode holds an 8x8 block's mode at all four of its 4x4 positions) */
            for (int b8 = 0; b8 < 4; b8++) {
                int ma = -1, mb_ = -1;
                const int la = (b8 + 1) * 4 + 1, lb = (b8 + 2) * 4 + 2; /* luma4x4BlkIdx in the neighbouring macroblock */
                if (b8 & 1) ma = m->i4mode[((b8 >> 1) * 2) * 4 + 0];
                else if (intra_nb_ok(d, mx - 1, my, slice)) { const dmb_t *n = m - 1; ma = n->is_i4 ? n->i4mode[D_blky[la] * 4 + D_blkx[la]] : 2; }
                if (b8 >> 1) mb_ = m->i4mode[0 * 4 + (b8 & 1) * 2];
                else if (intra_nb_ok(d, mx, my - 1, slice)) { const dmb_t *n = m - d->mbw; mb_ = n->is_i4 ? n->i4mode[D_blky[lb] * 4 + D_blkx[lb]] : 2; }
                int pred = (ma < 0 || mb_ < 0) ? 2 : (ma < mb_ ? ma : mb_), mode;
                if (rd_bit(d)) mode = pred;
                else { int rem = (int)rd_bits(d, 3); mode = rem < pred ? rem : rem + 1; }
                for (int j = 0; j < 4; j++) m->i4mode[((b8 >> 1) * 2 + (j >> 1)) * 4 + (b8 & 1) * 2 + (j & 1)] = (uint8_t)mode;
            }
        } else if (m->is_i4) {
            /* 8.3.1.1 mode prediction */
            for (int b = 0; b < 16; b++) {
                int bx = D_blkx[b], by = D_blky[b];
                int ma = -1, mb_ = -1; /* -1: dcPredModePredictedFlag */
                if (bx > 0) ma = m->i4mode[by * 4 + bx - 1];
                else if (intra_nb_ok(d, mx - 1, my, slice)) { const dmb_t *n = m - 1; ma = n->is_i4 ? n->i4mode[by * 4 + 3] : 2; }
                if (by > 0) mb_ = m->i4mode[(by - 1) * 4 + bx];
                else if (intra_nb_ok(d, mx, my - 1, slice)) { const dmb_t *n = m - d->mbw; mb_ = n->is_i4 ? n->i4mode[12 + bx] : 2; }
                int pred = (ma < 0 || mb_ < 0) ? 2 : (ma < mb_ ? ma : mb_);
                if (rd_bit(d)) m->i4mode[by * 4 + bx] = (uint8_t)pred;
                else { int rem = (int)rd_bits(d, 3); m->i4mode[by * 4 + bx] = (uint8_t)(rem < pred ? rem : rem + 1); }
            }
        }
        int cmode = (int)rd_ue(d);
        if (cmode > 3) return fail(d, "intra_chroma_pred_mode %d", cmode);
        cap_cmode = cmode;
        if (m->is_i4) { unsigned k = rd_ue(d); if (k > 47) return fail(d, "cbp codeNum %u", k); cbp = D_cbp_intra[k]; }
        int err = 0;
        pred_chroma8(d, mx, my, slice, cmode, &err);
        if (i16) pred_i16(d, mx, my, slice, i16mode, &err);
        if (err) return fail(d, "intra mode needs an unavailable neighbour at MB %d,%d", mx, my);
    }
    if (i16 || cbp) {
        int dq = rd_se(d);
        *qp = (*qp + dq + 52) % 52;
    }
    m->qp = (int8_t)*qp; m->qpc = (int8_t)chroma_qp(d, *qp);
    int q = *qp;
    if (cr) {
        cr->mb_type = (uint8_t)(!intra ? 1 : m->is_i4 ? 2 : 0); cr->mvx = m->mvx; cr->mvy = m->mvy; cr->qp = (uint8_t)q;
        cr->i16_mode = (uint8_t)i16mode; cr->chroma_mode = (uint8_t)cap_cmode;
        if (m->t8) cr->nzmask |= ORC_NZ_T8;
        if (cl && m->is_i4) for (int b = 0; b < 16; b++) cl[ORC_L_LDC + b] = m->i4mode[D_blky[b] * 4 + D_blkx[b]];
    }
    /* --- luma residual */
    int dcy[16]; memset(dcy, 0, sizeof dcy);
    if (i16) {
        int16_t l[16];
        if (residual_block(d, l, 16, ctx_nC(d, mx, my, slice, 0, 0, 0)) < 0) return fail(d, "I16 DC residual");
        if (cl) { memcpy(cl + ORC_L_LDC, l, 32); if (any_nz(l, 16)) cr->nzmask |= ORC_NZ_LDC; }
        int c[4][4], t2[4][4], f[4][4];
        for (int k = 0; k < 16; k++) c[D_zz_y[k]][D_zz_x[k]] = l[k];
        for (int y = 0; y < 4; y++) { /* 8.5.10: f = A c A with the +1 +1 +1 +1 / +1 +1 -1 -1 / +1 -1 -1 +1 / +1 -1 +1 -1 matrix */
            t2[y][0] = c[y][0] + c[y][1] + c[y][2] + c[y][3]; t2[y][1] = c[y][0] + c[y][1] - c[y][2] - c[y][3];
            t2[y][2] = c[y][0] - c[y][1] - c[y][2] + c[y][3]; t2[y][3] = c[y][0] - c[y][1] + c[y][2] - c[y][3];
        }
        for (int x = 0; x < 4; x++) {
            f[0][x] = t2[0][x] + t2[1][x] + t2[2][x] + t2[3][x]; f[1][x] = t2[0][x] + t2[1][x] - t2[2][x] - t2[3][x];
            f[2][x] = t2[0][x] - t2[1][x] - t2[2][x] + t2[3][x]; f[3][x] = t2[0][x] - t2[1][x] + t2[2][x] - t2[3][x];
        }
        int ls = D_ls_a[q % 6] * 16;
        for (int y = 0; y < 4; y++)
            for (int x = 0; x < 4; x++)
                dcy[y * 4 + x] = q >= 36 ? (f[y][x] * ls) << (q / 6 - 6) : (f[y][x] * ls + (1 << (5 - q / 6))) >> (6 - q / 6);
    }
    if (m->t8) { /* 7.3.5.3.2: each 8x8 block arrives as four interleaved 4x4 CAVLC blocks; 8.5.13 reconstruction */
        for (int i8 = 0; i8 < 4; i8++) {
            if (m->is_i4) { /* Intra_8x8: the block is predicted from what has been reconstructed so far, residual or not */
                int err = 0;
                const int up = i8 >= 2 || intra_nb_ok(d, mx, my - 1, slice), lf = (i8 & 1) || intra_nb_ok(d, mx - 1, my, slice);
                const int ul = i8 == 0 ? intra_nb_ok(d, mx - 1, my - 1, slice) : i8 == 1 ? intra_nb_ok(d, mx, my - 1, slice) : i8 == 2 ? intra_nb_ok(d, mx - 1, my, slice) : 1;
                const int ur = i8 == 0 ? intra_nb_ok(d, mx, my - 1, slice) : i8 == 1 ? intra_nb_ok(d, mx + 1, my - 1, slice) : i8 == 2;
                pred_i8(d, mx * 16 + (i8 & 1) * 8, my * 16 + (i8 >> 1) * 8, m->i4mode[((i8 >> 1) * 2) * 4 + (i8 & 1) * 2], up, lf, ul, ur, &err);
                if (err) return fail(d, "I8x8 mode needs an unavailable neighbour");
            }
            if (!(cbp & (1 << i8))) continue;
            int lev8[64], any = 0;
            memset(lev8, 0, sizeof lev8);
            for (int j = 0; j < 4; j++) {
                int b = 4 * i8 + j, bx = D_blkx[b], by = D_blky[b];
                int16_t l[16];
                int n = residual_block(d, l, 16, ctx_nC(d, mx, my, slice, 0, bx, by));
                if (n < 0) return fail(d, "luma 8x8 residual at MB %d,%d", mx, my);
                m->tc_l[by * 4 + bx] = (uint8_t)n;
                any |= n;
                if (cl) { memcpy(cl + ORC_L_LUMA + b * 16, l, 32); if (n) cr->nzmask |= 1u << b; }
                for (int k = 0; k < 16; k++) lev8[4 * k + j] = l[k];
            }
            if (!any) continue;
            for (int j = 0; j < 4; j++) m->coded |= (uint16_t)(1u << (D_blky[4 * i8 + j] * 4 + D_blkx[4 * i8 + j])); /* 8x8 granularity */
            int c8[64], t8[64], r8[64];
            memset(c8, 0, sizeof c8);
            for (int k = 0; k < 64; k++) {
                int x, y; zz8_xy(k, &x, &y);
                int ls = 16 * D_ls8[q % 6][ls8_class(x, y)];
                c8[y * 8 + x] = q >= 36 ? (lev8[k] * ls) << (q / 6 - 6) : (lev8[k] * ls + (1 << (5 - q / 6))) >> (6 - q / 6);
            }
            for (int y = 0; y < 8; y++) inv8_1d(c8 + 8 * y, 1, t8 + 8 * y, 1);
            for (int x = 0; x < 8; x++) inv8_1d(t8 + x, 8, r8 + x, 8);
            int X = mx * 16 + (i8 & 1) * 8, Y = my * 16 + (i8 >> 1) * 8;
            for (int y = 0; y < 8; y++)
                for (int x = 0; x < 8; x++) DY(d, X + x, Y + y) = (uint8_t)u8clip(DY(d, X + x, Y + y) + ((r8[y * 8 + x] + 32) >> 6));
        }
    } else
    for (int b = 0; b < 16; b++) {
        int bx = D_blkx[b], by = D_blky[b], X = mx * 16 + bx * 4, Y = my * 16 + by * 4;
        if (m->is_i4) {
            int err = 0;
            int has_up = by > 0 || intra_nb_ok(d, mx, my - 1, slice);
            int has_left = bx > 0 || intra_nb_ok(d, mx - 1, my, slice);
            int has_ul = (bx > 0 && by > 0) ? 1 : (bx > 0) ? intra_nb_ok(d, mx, my - 1, slice) : (by > 0) ? intra_nb_ok(d, mx - 1, my, slice) : intra_nb_ok(d, mx - 1, my - 1, slice);
            int has_ur;
            if (by == 0) has_ur = bx < 3 ? intra_nb_ok(d, mx, my - 1, slice) : intra_nb_ok(d, mx + 1, my - 1, slice);
            else { /* inside the macroblock: the block up-right must precede this one in decoding order */
                static const uint8_t idx_of[16] = {0, 1, 4, 5, 2, 3, 6, 7, 8, 9, 12, 13, 10, 11, 14, 15}; /* raster -> blkIdx */
                has_ur = bx < 3 && idx_of[(by - 1) * 4 + bx + 1] < b;
            }
            pred_i4(d, X, Y, m->i4mode[by * 4 + bx], has_up, has_left, has_ul, has_ur, &err);
            if (err) return fail(d, "I4x4 mode needs an unavailable neighbour");
        }
        int16_t l[16]; memset(l, 0, sizeof l);
        int n = 0;
        if (cbp & (1 << (b >> 2))) {
            n = i16 ? residual_block(d, l, 15, ctx_nC(d, mx, my, slice, 0, bx, by)) : residual_block(d, l, 16, ctx_nC(d, mx, my, slice, 0, bx, by));
            if (n < 0) return fail(d, "luma residual at MB %d,%d blk %d", mx, my, b);
            m->tc_l[by * 4 + bx] = (uint8_t)n;
            if (n) m->coded |= (uint16_t)(1u << (by * 4 + bx));
            if (cl) { memcpy(cl + ORC_L_LUMA + b * 16 + (i16 ? 1 : 0), l, i16 ? 30 : 32); if (n) cr->nzmask |= 1u << b; }
        }
        if (n || i16) {
            int co[4][4];
            scale4x4(l, i16 ? 1 : 0, q, co);
            if (i16) co[0][0] = dcy[by * 4 + bx];
            add_res4(d, X, Y, co);
        }
    }
    return decode_chroma_residual(d, mx, my, slice, cbp >> 4, m->qpc);
}

/* ---------------------------------------------------------------- deblocking */
static void edge_filter(uint8_t *s, ptrdiff_t across, int bS, int qpav, int off_a, int off_b, int is_chroma) {
    int ia = qpav + off_a, ib = qpav + off_b;
    ia = ia < 0 ? 0 : ia > 51 ? 51 : ia; ib = ib < 0 ? 0 : ib > 51 ? 51 : ib;
    int alpha = ia < 16 ? 0 : D_alpha[ia - 16], beta = ib < 16 ? 0 : D_beta[ib - 16];
    int p0 = s[-across], p1 = s[-2 * across], q0 = s[0], q1 = s[across];
    if (abs(p0 - q0) >= alpha || abs(p1 - p0) >= beta || abs(q1 - q0) >= beta) return;
    if (is_chroma) {
        if (bS == 4) { s[-across] = (uint8_t)((2 * p1 + p0 + q1 + 2) >> 2); s[0] = (uint8_t)((2 * q1 + q0 + p1 + 2) >> 2); return; }
        int t0 = ia < 17 ? 0 : (bS == 1 ? D_tc0_1 : bS == 2 ? D_tc0_2 : D_tc0_3)[ia - 17];
        int tc = t0 + 1, dl = (((q0 - p0) * 4) + (p1 - q1) + 4) >> 3;
        dl = dl < -tc ? -tc : dl > tc ? tc : dl;
        s[-across] = (uint8_t)u8clip(p0 + dl); s[0] = (uint8_t)u8clip(q0 - dl);
        return;
    }
    int p2 = s[-3 * across], q2 = s[2 * across];
    int ap = abs(p2 - p0) < beta, aq = abs(q2 - q0) < beta;
    if (bS == 4) {
        int p3 = s[-4 * across], q3 = s[3 * across], strong = abs(p0 - q0) < (alpha >> 2) + 2;
        if (ap && strong) {
            s[-across] = (uint8_t)((p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3);
            s[-2 * across] = (uint8_t)((p2 + p1 + p0 + q0 + 2) >> 2);
            s[-3 * across] = (uint8_t)((2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3);
        } else s[-across] = (uint8_t)((2 * p1 + p0 + q1 + 2) >> 2);
        if (aq && strong) {
            s[0] = (uint8_t)((p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3);
            s[across] = (uint8_t)((p0 + q0 + q1 + q2 + 2) >> 2);
            s[2 * across] = (uint8_t)((2 * q3 + 3 * q2 + q1 + q0 + p0 + 4) >> 3);
        } else s[0] = (uint8_t)((2 * q1 + q0 + p1 + 2) >> 2);
        return;
    }
    int t0 = ia < 17 ? 0 : (bS == 1 ? D_tc0_1 : bS == 2 ? D_tc0_2 : D_tc0_3)[ia - 17];
    int tc = t0 + ap + aq, dl = (((q0 - p0) * 4) + (p1 - q1) + 4) >> 3;
    dl = dl < -tc ? -tc : dl > tc ? tc : dl;
    s[-across] = (uint8_t)u8clip(p0 + dl); s[0] = (uint8_t)u8clip(q0 - dl);
    int avg = (p0 + q0 + 1) >> 1;
    if (ap) { int v = (p2 + avg - 2 * p1) >> 1; v = v < -t0 ? -t0 : v > t0 ? t0 : v; s[-2 * across] = (uint8_t)(p1 + v); }
    if (aq) { int v = (q2 + avg - 2 * q1) >> 1; v = v < -t0 ? -t0 : v > t0 ? t0 : v; s[across] = (uint8_t)(q1 + v); }
}
static int strength(const dmb_t *P, int pi, const dmb_t *Q, int qi, int mbedge) { /* pi, qi: raster 4x4 index */
    if (P->kind == 0 || Q->kind == 0) return mbedge ? 4 : 3;
    if (((P->coded >> pi) & 1) || ((Q->coded >> qi) & 1)) return 2;
    const int pq = ((pi >> 2) >> 1) * 2 + ((pi & 3) >> 1), qq = ((qi >> 2) >> 1) * 2 + ((qi & 3) >> 1); /* the 8x8 quadrants of the two blocks */
    if (abs(P->qmv[pq][0] - Q->qmv[qq][0]) >= 4 || abs(P->qmv[pq][1] - Q->qmv[qq][1]) >= 4) return 1;
    return 0;
}
static void deblock_picture(orc_dec_t *d) {
    size_t st = (size_t)d->mbw * 16;
    for (int my = 0; my < d->mbh; my++)
        for (int mx = 0; mx < d->mbw; mx++) {
            dmb_t *Q = &d->mb[my * d->mbw + mx];
            if (Q->dbf_idc == 1) continue;
            for (int dir = 0; dir < 2; dir++) /* 0: vertical edges, 1: horizontal edges */
                for (int e = 0; e < 4; e++) {
                    const dmb_t *P = Q;
                    if (e == 0) {
                        int nx = dir ? mx : mx - 1, ny = dir ? my - 1 : my;
                        if (nx < 0 || ny < 0) continue;
                        P = &d->mb[ny * d->mbw + nx];
                        if (Q->dbf_idc == 2 && P->slice != Q->slice) continue;
                    }
                    for (int k = 0; k < 16; k++) {
                        int seg = k >> 2;
                        if (Q->t8 && (e & 1)) break; /* 8.7: with transform_size_8x8_flag only 8x8 block edges are filtered */
                        int qi = dir ? e * 4 + seg : seg * 4 + e;
                        int pi = e ? (dir ? (e - 1) * 4 + seg : seg * 4 + e - 1) : (dir ? 12 + seg : seg * 4 + 3);
                        int bS = strength(P, pi, Q, qi, e == 0);
                        if (!bS) continue;
                        uint8_t *s = dir ? d->cur_y + (size_t)(my * 16 + e * 4) * st + mx * 16 + k
                                         : d->cur_y + (size_t)(my * 16 + k) * st + mx * 16 + e * 4;
                        edge_filter(s, dir ? (ptrdiff_t)st : 1, bS, (P->qp + Q->qp + 1) >> 1, Q->dbf_a, Q->dbf_b, 0);
                        if (!(e & 1) && !(k & 1)) { /* chroma: edges 0 and 2, one chroma line per two luma lines */
                            for (int c = 0; c < 2; c++) {
                                int kc = k >> 1;
                                uint8_t *sc = dir ? d->cur_uv + (size_t)(my * 8 + e * 2) * st + 2 * (mx * 8 + kc) + c
                                                  : d->cur_uv + (size_t)(my * 8 + kc) * st + 2 * (mx * 8 + e * 2) + c;
                                edge_filter(sc, dir ? (ptrdiff_t)st : 2, bS, (P->qpc + Q->qpc + 1) >> 1, Q->dbf_a, Q->dbf_b, 1);
                            }
                        }
                    }
                }
        }
}

/* ---------------------------------------------------------------- slice + access unit */
static int decode_slice(orc_dec_t *d, int nal_type, int ref_idc) {
    if (!d->have_sps || !d->have_pps) return fail(d, "slice before parameter sets");
    int first = (int)rd_ue(d), st = (int)rd_ue(d) % 5;
    rd_ue(d);
    int is_p = st == 0;
    if (st != 0 && st != 2) return fail(d, "slice_type %d unsupported", st);
    rd_bits(d, d->log2_fn);
    int idr = nal_type == 5;
    if (idr) rd_ue(d);
    if (d->poc_type == 0) { rd_bits(d, d->log2_poc_lsb); if (d->bottom_field_poc) rd_se(d); }
    else if (d->poc_type == 1 && !d->delta_pic_order_always_zero) { rd_se(d); if (d->bottom_field_poc) rd_se(d); }
    if (d->redundant_cnt) rd_ue(d);
    if (is_p) {
        if (rd_bit(d)) { if (rd_ue(d) != 0) return fail(d, "num_ref_idx_active > 1 unsupported"); }
        else if (d->num_ref_default > 1) return fail(d, "num_ref_idx_active > 1 unsupported");
        if (rd_bit(d)) return fail(d, "ref_pic_list_modification unsupported");
    }
    if (ref_idc) {
        if (idr) { rd_bit(d); rd_bit(d); }
        else if (rd_bit(d)) return fail(d, "MMCO unsupported");
    }
    int qp = d->pic_init_qp + rd_se(d);
    int idc = 0, oa = 0, ob = 0;
    if (d->dbf_present) { idc = (int)rd_ue(d); if (idc != 1) { oa = 2 * rd_se(d); ob = 2 * rd_se(d); } }
    if (d->rb_fail || qp < 0 || qp > 51) return fail(d, "bad slice header");
    if (is_p && !d->have_ref) return fail(d, "P slice without a reference picture");
    if (first == 0 || !d->pic_open) {
        for (int i = 0; i < d->mbw * d->mbh; i++) d->mb[i].kind = -1;
        d->pic_open = 1; d->n_slices = 0;
    }
    int slice = d->n_slices++;
    int n = d->mbw * d->mbh, addr = first;
    while (addr < n) {
        if (is_p) {
            unsigned run = rd_ue(d);
            if (d->rb_fail || addr + (int)run > n) return fail(d, "bad mb_skip_run");
            while (run--) {
                dmb_t *m = &d->mb[addr]; m->dbf_idc = (int8_t)idc; m->dbf_a = (int8_t)oa; m->dbf_b = (int8_t)ob;
                if (decode_mb(d, addr % d->mbw, addr / d->mbw, slice, 1, 1, &qp)) return -1;
                addr++;
            }
            if (!more_data(d)) break;
            if (addr >= n) return fail(d, "data after last macroblock");
        }
        dmb_t *m = &d->mb[addr]; m->dbf_idc = (int8_t)idc; m->dbf_a = (int8_t)oa; m->dbf_b = (int8_t)ob;
        if (decode_mb(d, addr % d->mbw, addr / d->mbw, slice, is_p, 0, &qp)) return -1;
        if (d->rb_fail) return fail(d, "bitstream exhausted inside MB %d", addr);
        addr++;
        if (!more_data(d)) break;
    }
    if (more_data(d)) return fail(d, "%zu unread bits at end of slice", d->rb_bits - d->pos);
    return 0;
}

orc_dec_t *orc_dec_open(void) { return (orc_dec_t *)calloc(1, sizeof(orc_dec_t)); }
void orc_dec_set_capture(orc_dec_t *d, orc_mbinfo_t *mbinfo, int16_t *levels) { d->cap_mbi = mbinfo; d->cap_lev = levels; }
void orc_dec_close(orc_dec_t *d) {
    if (!d) return;
    free(d->cur_y); free(d->cur_uv); free(d->ref_y); free(d->ref_uv); free(d->mb); free(d->rbsp); free(d);
}
int orc_dec_decode(orc_dec_t *d, const uint8_t *au, size_t len) {
    d->err[0] = 0;
    int got_slice = 0;
    size_t i = 0;
    while (i + 3 <= len) { /* Annex B: locate 00 00 01 */
        if (!(au[i] == 0 && au[i + 1] == 0 && au[i + 2] == 1)) { i++; continue; }
        size_t s = i + 3, e = s;
        while (e + 3 <= len && !(au[e] == 0 && au[e + 1] == 0 && (au[e + 2] == 1 || au[e + 2] == 0))) e++;
        if (e + 3 > len) e = len;
        if (e > s) {
            int hdr = au[s], type = hdr & 31, ref_idc = (hdr >> 5) & 3;
            if (hdr & 0x80) return fail(d, "forbidden_zero_bit set");
            if (type == 7 || type == 8 || type == 1 || type == 5) {
                if (set_rbsp(d, au + s + 1, e - s - 1)) return fail(d, "empty NAL");
                int r = type == 7 ? parse_sps(d) : type == 8 ? parse_pps(d) : decode_slice(d, type, ref_idc);
                if (r) return r;
                if (type == 1 || type == 5) got_slice = 1;
            }
        }
        i = e;
    }
    if (!got_slice) return 0;
    for (int k = 0; k < d->mbw * d->mbh; k++) if (d->mb[k].kind < 0) return fail(d, "macroblock %d missing from the access unit", k);
    deblock_picture(d);
    uint8_t *t = d->cur_y; d->cur_y = d->ref_y; d->ref_y = t;
    t = d->cur_uv; d->cur_uv = d->ref_uv; d->ref_uv = t;
    d->have_ref = 1; d->pic_open = 0;
    return 1;
}
int orc_dec_width(const orc_dec_t *d) { return d->mbw * 16 - d->crop_l - d->crop_r; }
int orc_dec_height(const orc_dec_t *d) { return d->mbh * 16 - d->crop_t - d->crop_b; }
int orc_dec_coded_width(const orc_dec_t *d) { return d->mbw * 16; }
int orc_dec_coded_height(const orc_dec_t *d) { return d->mbh * 16; }
const uint8_t *orc_dec_y(const orc_dec_t *d) { return d->ref_y; }   /* last completed picture */
const uint8_t *orc_dec_uv(const orc_dec_t *d) { return d->ref_uv; }
const char *orc_dec_error(const orc_dec_t *d) { return d->err; }
