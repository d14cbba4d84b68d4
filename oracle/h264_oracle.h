/*
 * oracle/h264_oracle.h -- TEST INFRASTRUCTURE ONLY (CPU oracle for the H.264 hot path).
 *
 * PARITY UNPINNED at the codec boundary: the reference (CERALIVE/ceracoder) holds no
 * codec arithmetic -- its encoder is the token `x264enc speed-preset=2 key-int-max=60`
 * (/root/reference/pipeline/generic/x264_superfast_camlink:5, resolved by
 * /root/reference/src/io/pipeline_loader.c:59) backed by the un-vendored, un-pinned
 * libx264 -- and it has no golden bitstreams.  This oracle is a
 * scalar restatement of ITU-T H.264 (sections cited per function) plus this repo's own
 * non-normative encoder decisions (motion search rule, mode decision, quantiser dead
 * zone).  It is pinned by (i) known-answer tests of every constant table and of the
 * Exp-Golomb / emulation-prevention / transform identities, and (ii) an independently
 * written decoder (h264_dec_oracle.c) whose output must equal the encoder's
 * reconstruction bit for bit.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 */
#ifndef H264_ORACLE_H
#define H264_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Per-macroblock side record, 16 bytes (SURVEY.md section 8d "16 B/MB info"). */
typedef struct {
    int16_t mvx, mvy;     /* luma motion vector, quarter-sample units (P macroblocks)  */
    uint8_t mb_type;      /* 0 = I16x16, 1 = P_L0_16x16 (P_Skip is an entropy decision),
                             2 = I4x4 (its 16 modes sit in levels[ORC_L_LDC + blkIdx])    */
    uint8_t i16_mode;     /* Intra16x16PredMode 0 V, 1 H, 2 DC, 3 Plane                 */
    uint8_t chroma_mode;  /* intra_chroma_pred_mode 0 DC, 1 H, 2 V, 3 Plane             */
    uint8_t qp;           /* QP_Y of this macroblock                                    */
    uint32_t nzmask;      /* b0-15 luma blkIdx (AC only for I16x16), b16-19 Cb AC,
                             b20-23 Cr AC, b24 luma DC (I16x16), b25 Cb DC, b26 Cr DC   */
    uint32_t cost;        /* motion-search cost (P) or intra SAD luma+chroma (I)        */
} orc_mbinfo_t;

/* Levels per macroblock, int16, coding (zig-zag) order:
 *   [  0..255] luma blkIdx 0..15 x 16   ([0] of each block is 0 for I16x16)
 *   [256..271] Intra16x16 DC levels
 *   [272..275] Cb DC, [276..279] Cr DC
 *   [280..407] chroma AC: Cb blk 0..3, Cr blk 0..3, x 16 ([0] unused = 0)      */
#define ORC_LEVELS_PER_MB 408
#define ORC_L_LUMA   0
#define ORC_L_LDC    256
#define ORC_L_CDC    272
#define ORC_L_CAC    280

#define ORC_NZ_LDC   (1u << 24)
#define ORC_NZ_CBDC  (1u << 25)
#define ORC_NZ_CRDC  (1u << 26)
#define ORC_NZ_T8    (1u << 27) /* P macroblock coded with the 8x8 transform (transform_size_8x8_flag) */

/* ---- stage functions (each is the checker for one HIP kernel) ------------------- */

/* Full-search integer-pel SAD motion search, +-range, on coded-size luma planes; vectors may leave the picture.  Keeps the
 * SAD of every candidate (ORC_SURF uint16 per macroblock, index (dy+16)*33 + dx+16, 0xFFFF outside the range) and makes a first
 * selection with the vector bits charged against zero.  orc_me_select: one Jacobi iteration of the selection, the bits charged
 * against the median of the neighbours' vectors in `in`.  Per macroblock 8 bytes: whole-sample vector (quarter-sample
 * units, multiples of 4), its SAD, the vector bits it was charged (cost = sad + lambda * bits). */
#define ORC_SURF (33 * 33)
#define ORC_SEL_BONUS 2 /* header bits every candidate but the P_Skip one is charged on top of its vector bits */
typedef struct { int16_t mvx, mvy; uint16_t sad, bits; } orc_imv_t;
void orc_me_frame(const uint8_t *cur_y, const uint8_t *ref_y, int stride, int mbw, int mbh,
                  int range, int qp, uint16_t *surf, orc_imv_t *imv, int threads);
void orc_me_select(const uint16_t *surf, int mbw, int mbh, int range, int qp, const orc_imv_t *in, orc_imv_t *out, int threads);

/* Half- then quarter-sample refinement of the vectors left by orc_me_frame (mvx,mvy,cost updated). */
void orc_subpel_frame(const uint8_t *cur_y, const uint8_t *ref_y, int stride, int mbw, int mbh, int qp,
                      orc_mbinfo_t *mbi, int threads);

/* P picture: motion compensation + residual + 4x4 transform/quant + normative
 * dequant/inverse + reconstruction (pre-deblock) for every macroblock. */
void orc_inter_frame(const uint8_t *src_y, const uint8_t *src_uv, const uint8_t *ref_y,
                     const uint8_t *ref_uv, uint8_t *rec_y, uint8_t *rec_uv, int stride,
                     int mbw, int mbh, int qp, orc_mbinfo_t *mbi, int16_t *levels);

/* SADs of every intra candidate measured against predictions built from SOURCE neighbours (0xFFFF: mode not available). */
typedef struct { uint16_t i16[4], chroma[4], i4[16][9]; } orc_isad_t; /* 304 bytes per macroblock */
void orc_intra_analyse(const uint8_t *src_y, const uint8_t *src_uv, int stride, int mbw, int mbh, orc_isad_t *out);
/* Mode decisions of an I picture from those SADs alone: 16 bytes per macroblock (layout shared with the device) */
typedef struct { uint8_t modes4[16]; uint8_t mode16, cmode, use_i4, pad; uint32_t cost, cost_luma, rsv0; } orc_idec_t; /* 32 bytes; cost = luma + chroma */
void orc_intra_decide(const orc_isad_t *isad, int mbw, int mbh, int qp, int i4x4, orc_idec_t *out);

/* I picture: Intra16x16 + chroma prediction, mode decision by SAD, transform/quant,
 * reconstruction (pre-deblock), macroblocks in raster order. */
void orc_intra_frame(const uint8_t *src_y, const uint8_t *src_uv, uint8_t *rec_y,
                     uint8_t *rec_uv, int stride, int mbw, int mbh, int qp, int drop,
                     orc_mbinfo_t *mbi, int16_t *levels);

/* P macroblocks, fused stage (what the device's pmb_kernel computes): predictor estimates from the whole-sample field `imv`,
 * skip probe, sub-sample refinement (`refine`), intra-or-inter against `idec` (may be NULL: no intra macroblocks), inter
 * residual with coefficient decimation.  `drop`: rate control's ladder below QP 51, 0 .. ORC_DROP_MAX.  Macroblocks decided
 * intra get their record's type and modes only; orc_intra_p_frame reconstructs them afterwards. */
#define ORC_DROP_MAX 12
#define ORC_SKIP_MARGIN_BITS 4 /* the skip probe runs when SAD(ps_est) <= SAD(best whole-sample vector) + lambda * this */
#define ORC_INTRA_GATE(lambda) (768u + 8u * (uint32_t)(lambda)) /* whole-sample search cost below which a P macroblock is never analysed for intra */
enum { ORC_F_MVDCOST = 1, ORC_F_SKIPPROBE = 2, ORC_F_DECIMATE = 4, ORC_F_SATD = 8, ORC_F_INTRAP = 16, ORC_F_I4P = 32 /* intra macroblocks of P pictures may be Intra_4x4 (x264 superfast: partitions i8x8,i4x4) */, ORC_F_PART = 64 /* inter partitions 16x8 / 8x16 / 8x8 (oracle-side groundwork: encoder decision, syntax, decoder; the device does not produce them yet, so it is not part of ORC_F_ALL) */, ORC_F_ALL = 63 };
void orc_set_features(int mask); /* process-wide ablation switches for the rate-distortion tables (default ORC_F_ALL = what the device does) */
int orc_get_features(void);
uint32_t orc_drop_threshold(int drop);
int orc_decimate_score(const int16_t *lev, int first);
uint32_t orc_satd16(const uint8_t *src, int stride, const uint8_t *pred /* 16 x 16, stride 16 */);
void orc_pmb_frame(const uint8_t *src_y, const uint8_t *src_uv, const uint8_t *ref_y, const uint8_t *ref_uv, uint8_t *rec_y, uint8_t *rec_uv,
                   int stride, int mbw, int mbh, int qp, int drop, int refine, const orc_imv_t *imv, const uint16_t *surf, const orc_idec_t *idec,
                   orc_mbinfo_t *mbi, int16_t *levels, int threads);
void orc_intra_p_frame(const uint8_t *src_y, const uint8_t *src_uv, uint8_t *rec_y, uint8_t *rec_uv, int stride, int mbw, int mbh, int qp,
                       const orc_idec_t *idec, orc_mbinfo_t *mbi, int16_t *levels);

void orc_set_transform8x8(int on); /* process-wide: High-profile stream, 8x8 transform for P macroblocks (default off) */
int orc_get_transform8x8(void);
void orc_fdct8(const int in[64], int out[64]);
void orc_idct8(const int in[64], int out[64]);
int orc_quant8(int coef, int qp, int pos, int intra);
int orc_dequant8(int level, int qp, int pos);
int orc_zigzag8(int k);
void orc_set_i4x4(int on); /* process-wide: try Intra_4x4 besides Intra_16x16 (default on) */

/* In-loop deblocking filter, in place, normative macroblock raster order (8.7). */
void orc_deblock_frame(uint8_t *rec_y, uint8_t *rec_uv, int stride, int mbw, int mbh,
                       const orc_mbinfo_t *mbi);

/* ---- whole-encoder wrapper ------------------------------------------------------ */
typedef struct orc_enc orc_enc_t;

orc_enc_t *orc_enc_open(int width, int height, int fps_num, int fps_den, int gop,
                        int me_range, int threads);
void orc_enc_close(orc_enc_t *e);
void orc_enc_set_subpel(orc_enc_t *e, int on);
void orc_enc_set_aq(orc_enc_t *e, int on);      /* adaptive quantisation: a QP offset per macroblock from the source's luma variance (default off) */
void orc_set_aq_map(const int8_t *off);          /* stage functions: the offsets of the picture being coded (NULL: none) */
void orc_aq_offsets(const uint8_t *src_y, int stride, int mbw, int mbh, int8_t *off);
int orc_aq_offset_of(uint32_t sum, uint32_t sum_sq);
void orc_qp_chain_slices(orc_mbinfo_t *mbi, int nmb, int slice_qp, int slice_mbs); /* ... with a new slice every slice_mbs macroblocks (0: one slice) */
void orc_set_i8x8(int on);                      /* process-wide: I pictures of a stream with the 8x8 transform (orc_set_transform8x8) may use Intra_8x8 macroblocks */
int orc_get_i8x8(void);
void orc_intra_decide8(const uint8_t *src_y, int stride, int mbw, int mbh, int qp, orc_idec_t *idec); /* after orc_intra_decide: Intra_8x8 where strictly cheaper (use_i4 = 2, modes in modes4[0..3]) */
void orc_set_part_levels(const int16_t *levels); /* stage functions (deblocking, slice writer): where the vectors of an inter macroblock's partitions 1 .. 3 lie (the levels of the picture; NULL: 16x16 only) */
void orc_set_slice_rows(int rows);              /* stage functions: the picture being coded is cut into slices of `rows` macroblock rows (0: one slice) */
void orc_set_slice_deblock(int idc);            /* stage functions: disable_deblocking_filter_idc of the picture's slices, 0 (across slice boundaries) or 2 (slice-local) */
int orc_get_slice_deblock(void);
int orc_slice_rows_for(int mbh, int n, int local_deblock); /* rows per slice for n slices (0: one slice); a multiple of four with slice-local deblocking */
int orc_get_slice_rows(void);
int orc_auto_intra_slices(int mbh);             /* the default number of slices of an I picture: about 17 rows each, at most 8 */
void orc_enc_set_intra_slices(orc_enc_t *e, int n); /* slices per I picture (0: the default above) */
void orc_enc_set_p_slices(orc_enc_t *e, int n);     /* slices per P picture (0 / 1: one slice, the default) */
void orc_enc_set_slice_deblock(orc_enc_t *e, int local); /* 1: the deblocking filter stops at slice boundaries (disable_deblocking_filter_idc 2; slice heights become multiples of four rows); default 0 */
void orc_qp_chain(orc_mbinfo_t *mbi, int nmb, int slice_qp); /* 7.4.5: macroblocks without mb_qp_delta take the QP_Y of the one before them */
void orc_enc_set_scenecut(orc_enc_t *e, int on); /* default on */
void orc_enc_set_sc_lag(orc_enc_t *e, int lag);  /* scene-cut recovery lands on picture k + lag (default 2; the device: pipeline_depth + 1 from depth 2 on) */
void orc_enc_set_me_iters(orc_enc_t *e, int n);  /* orc_me_select iterations after the first selection (default ORC_ME_ITERS) */
#define ORC_ME_ITERS 3
/* Encode one NV12 frame at a caller-chosen QP.  Returns 0, or <0 on error. */
int orc_enc_frame(orc_enc_t *e, const uint8_t *y, int y_stride, const uint8_t *uv,
                  int uv_stride, int qp, int force_idr, uint8_t *out, size_t out_cap,
                  size_t *out_len, int *is_idr);
/* The same with rate control's ladder below QP 51: drop 0 .. ORC_DROP_MAX for P pictures, ORC_DROP_SKIP = the whole picture
 * as one run of P_Skip macroblocks (the input planes are not read and may be NULL). */
#define ORC_DROP_SKIP 255
int orc_enc_frame2(orc_enc_t *e, const uint8_t *y, int y_stride, const uint8_t *uv, int uv_stride,
                   int qp, int drop, int force_idr, uint8_t *out, size_t out_cap, size_t *out_len, int *is_idr);
const orc_imv_t *orc_enc_imv(const orc_enc_t *e);
const orc_idec_t *orc_enc_idec(const orc_enc_t *e);
/* Views into the last encoded frame (coded size, stride = 16*mbw). */
const uint8_t *orc_enc_recon_y(const orc_enc_t *e);
const uint8_t *orc_enc_recon_uv(const orc_enc_t *e);
const uint8_t *orc_enc_prefilter_y(const orc_enc_t *e);
const uint8_t *orc_enc_prefilter_uv(const orc_enc_t *e);
const orc_mbinfo_t *orc_enc_mbinfo(const orc_enc_t *e);
const int16_t *orc_enc_levels(const orc_enc_t *e);
int orc_enc_mbw(const orc_enc_t *e);
int orc_enc_mbh(const orc_enc_t *e);

/* Parameter sets + slice entropy coding, usable on externally produced records
 * (this is how the product's device output is turned into the expected bytes). */
size_t orc_write_headers(uint8_t *out, size_t cap, int width, int height, int fps_num,
                         int fps_den);
size_t orc_write_slice(uint8_t *out, size_t cap, int mbw, int mbh, int is_idr,
                       int frame_num, int idr_pic_id, int qp, const orc_mbinfo_t *mbi,
                       const int16_t *levels);

/* ---- independent decoder (h264_dec_oracle.c) ------------------------------------ */
typedef struct orc_dec orc_dec_t;
orc_dec_t *orc_dec_open(void);
void orc_dec_close(orc_dec_t *d);
/* Feed one access unit (Annex B).  Returns number of pictures output (0/1), <0 error. */
int orc_dec_decode(orc_dec_t *d, const uint8_t *au, size_t len);
int orc_dec_width(const orc_dec_t *d);        /* cropped */
int orc_dec_height(const orc_dec_t *d);
int orc_dec_coded_width(const orc_dec_t *d);
int orc_dec_coded_height(const orc_dec_t *d);
const uint8_t *orc_dec_y(const orc_dec_t *d);  /* coded-size planes, stride = coded width */
const uint8_t *orc_dec_uv(const orc_dec_t *d); /* interleaved CbCr                         */
const char *orc_dec_error(const orc_dec_t *d);
/* Tests: have the decoder write the syntax it parses from the next access units into caller-owned arrays in the encoder's
 * layout (coded-size mbw*mbh records / mbw*mbh*ORC_LEVELS_PER_MB levels; cost = 0, nzmask from the parsed levels). */
void orc_dec_set_capture(orc_dec_t *d, orc_mbinfo_t *mbinfo, int16_t *levels);
/* One CAVLC residual block through the encoder oracle's cavlc_block (9.2.1-9.2.3): coef in scan order, maxnum 16 / 15 / 4,
 * nC as 9.2.1 derives it (-1: chroma DC).  Writes the bits MSB-first into out, returns their number. */
int orc_cavlc_block_bits(const int16_t *coef, int maxnum, int nC, uint8_t *out, size_t cap);

/* ---- small known-answer helpers exported for tests ------------------------------ */
void orc_fdct4(const int16_t in[16], int16_t out[16]);
void orc_idct4_add(const int32_t coef[16], uint8_t *dst, int stride);
int orc_quant4(int coef, int qp, int pos, int intra);
int orc_dequant4(int level, int qp, int pos);
size_t orc_nal_escape(const uint8_t *rbsp, size_t n, uint8_t *out, size_t cap);
int orc_ue_bits(uint32_t v, uint32_t *code);
uint32_t orc_table_checksum(int which);
int orc_me_lambda(int qp);

#ifdef __cplusplus
}
#endif
#endif
