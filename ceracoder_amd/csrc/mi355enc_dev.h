/* Internal: types shared by the host orchestration (enc_*.cpp), the host entropy coder
 * (h264_host.c) and the HIP kernels (k_*.hip, kernels_common.hpp).  Not part of the C ABI. */
#ifndef MI355ENC_DEV_H
#define MI355ENC_DEV_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 16-byte per-macroblock record (DESIGN.md "HBM layout") */
typedef struct {
    int16_t mvx, mvy;     /* luma motion vector, quarter-sample units               */
    uint8_t mb_type;      /* 0 I16x16, 1 P_L0_16x16, 2 I4x4 (modes in levels[L_LDC..]) */
    uint8_t i16_mode;     /* 0 V 1 H 2 DC 3 Plane                                   */
    uint8_t chroma_mode;  /* 0 DC 1 H 2 V 3 Plane                                   */
    uint8_t qp;
    uint32_t nzmask;      /* b0-15 luma blkIdx, b16-23 chroma AC, b24 luma DC, b25 Cb DC, b26 Cr DC */
    uint32_t cost;
} mb_info_t;

/* 8-byte result of the whole-sample motion search per macroblock (me_kernel / me_select_kernel) */
typedef struct {
    int16_t mvx, mvy;     /* whole-sample vector in quarter-sample units (multiples of 4)      */
    uint16_t sad;         /* its SAD                                                            */
    uint16_t bits;        /* the vector + header bits it was charged (cost = sad + lambda*bits) */
} imv_t;

#define MB_LEVELS 408
#define ISAD_PER_MB 152
#define IDEC_BYTES 32 /* intra decisions per macroblock: u8 modes4[16] (by blkIdx); u8 mode16, cmode, use_i4, 0; u32 cost (luma + chroma), cost_luma, 0 */
/* SAD surface of the motion search: per macroblock 35 rows (dy = -16 .. 18) x 36 columns (dx = -16 .. 19) of uint16; the search
 * range is the 33 x 33 upper-left part, the rest is what the lanes' 5 x 4 candidate tiles compute beyond it */
#define SURF_COLS 36
#define SURF_ROWS 35
#define SURF_U16 (SURF_ROWS * SURF_COLS)
#ifndef MI355_BAND_ROWS
#define MI355_BAND_ROWS 4 /* macroblock rows per band of the band deblocker (the intra band kernel has bands of its own height, k_intra_band_rows()) */
#endif
/* Words that one workgroup publishes and others poll (pmb_kernel's row counts, intra_p_kernel's progress words) lie this many words apart: one
 * per 4 KB, i.e. on different memory channels -- dozens of waves polling neighbouring words of one cache line make its channel a hot spot. */
#ifndef MI355_PROG_STRIDE
#define MI355_PROG_STRIDE 1024
#endif
#define ME_ITERS 3          /* Jacobi iterations of the vector selection after the search's own (oracle: ORC_ME_ITERS) */
#define SEL_BONUS 2         /* oracle: ORC_SEL_BONUS */
#define SKIP_MARGIN_BITS 4  /* oracle: ORC_SKIP_MARGIN_BITS */
#define INTRA_GATE(lambda) (768u + 8u * (unsigned)(lambda)) /* oracle: ORC_INTRA_GATE */
#define DROP_MAX 12
#define DROP_SKIP 255
#define L_LUMA 0
#define L_LDC 256
#define L_CDC 272
#define L_CAC 280
#define NZ_LDC (1u << 24)
#define NZ_CBDC (1u << 25)
#define NZ_CRDC (1u << 26)
#define NZ_T8 (1u << 27) /* transform_size_8x8_flag of a P macroblock */
#define PACK_BLOCKS_MAX 27 /* 32-byte blocks a macroblock can contribute to the packed level stream */

/* Device-resident description of the picture being encoded; kernels read it through one
 * pointer so that the per-picture launch sequence can be replayed as a hipGraph. */
typedef struct {
    const uint8_t *src_y, *src_uv; /* source NV12 (visible rows only; rows clamp to vis_h-1) */
    const uint8_t *ref_y, *ref_uv; /* previous reconstructed, deblocked picture (coded size) */
    uint8_t *rec_y, *rec_uv;       /* picture being reconstructed (coded size)               */
    mb_info_t *mbi;
    int16_t *levels;
    uint16_t *isad;                /* intra analysis: 152 u16 per macroblock {i16[4], chroma[4], i4[16][9]}, 0xFFFF = mode unavailable */
    uint8_t *idec;                 /* intra decisions of intra_analyse_kernel, IDEC_BYTES per macroblock */
    uint8_t *dbrec;                /* deblocking: 64 B per macroblock {bS nibbles V, H; alpha/beta/tc0 of 6 edge classes} */
    int32_t src_stride;            /* bytes per source luma row (= chroma row, NV12)          */
    int32_t stride;                /* coded-surface stride = 16*mbw                           */
    int32_t mbw, mbh, vis_h;
    int32_t qp, me_range, lambda;
    int32_t i4x4;                  /* try Intra_4x4 in I pictures */
    int32_t t8;                    /* P macroblocks use the 8x8 transform (High profile stream) */
    int32_t all_intra;             /* every macroblock of the picture is intra (IDR): the deblocker runs all edges without per-edge tests */
    /* P pictures */
    const uint8_t *me_ref_y;       /* what the whole-sample search runs against: the padded SOURCE luma of the last coded picture (coded size, stride `stride`) */
    uint8_t *psrc_out;             /* where this picture's padded source luma goes for the next picture's search (written by me_kernel / copy_luma_kernel) */
    uint16_t *surf;                /* SAD surfaces, SURF_U16 per macroblock */
    imv_t *imv_a, *imv_b, *imv_c;  /* whole-sample vector fields: the search writes imv_a, selection iteration k reads field k % 3 and writes (k + 1) % 3 */
    uint32_t epoch;                /* picture stamp (never 0): tags the progress words of intra_p_kernel so that nothing needs clearing */
    uint32_t drop_sad;             /* rate control's ladder below QP 51: 0 off, else the SAD below which a P macroblock carries no residual / takes the skip vector */
    int32_t iac_drop;              /* I pictures on rate control's ladder: 0 off, else the sum of level magnitudes up to which a macroblock's luma / chroma
                                      residual is not sent */
    int32_t intra_p;               /* P macroblocks may be intra (the analysis of this picture's source is in isad / idec) */
    const int8_t *qp_off;          /* adaptive quantisation: one QP offset per macroblock (aq_kernel), or null: one QP per picture */
    int32_t partitions;            /* P macroblocks may be split (shape in the record's i16_mode: 1 16x8, 2 8x16, 3 8x8; the vectors of partitions 1 .. 3 in the
                                      luma-DC slot of the macroblock's levels); the deblocker then takes boundary strengths per 8x8 quadrant */
    int32_t slice_rows;            /* a new slice every so many macroblock rows (0: one slice); the row above a slice's first row is not available (6.4.8): intra prediction,
                                      mode and vector predictors, nC, QP_Y,PRED */
    int32_t slice_dbf;             /* disable_deblocking_filter_idc of the picture's slices: 0 the deblocking filter runs across slice boundaries, 2 it stops at them
                                      (slice_rows is then a multiple of MI355_BAND_ROWS: a band of the deblocker never spans two slices) */
    int32_t i8;                    /* I pictures: try Intra_8x8 (High profile; intra_mode 0 only: the macroblock above-right has to be complete) */
} frame_ctx_t;

#ifdef __cplusplus
}
#endif

#ifdef __HIPCC__
#include <hip/hip_runtime.h>
/* launchers (k_*.hip); all asynchronous on `s`.  h_ctx: HOST copy of the context, passed to the kernel by value
 * (kernarg segment); d_ctx: device copy, for the kernels that are replayed from a hipGraph. */
void k_launch_me(const frame_ctx_t *h_ctx, int mbw, int row0, int row1, hipStream_t s);
void k_launch_copy_luma(const frame_ctx_t *h_ctx, hipStream_t s); /* I pictures: source luma -> psrc_out (rows beyond the visible picture repeat its last row) */
void k_launch_me_select(const frame_ctx_t *h_ctx, int mbw, int row0, int row1, const imv_t *in, imv_t *out, const imv_t *prev, int mode, hipStream_t s); /* mode 0: recompute all; 1 / 2: copy where the predictors equal zeros / those of `prev` */
void k_launch_me_select_all(const frame_ctx_t *h_ctx, int mbw, int row0, int row1, hipStream_t s); /* the ME_ITERS iterations */
void k_launch_intra_p(const frame_ctx_t *h_ctx, int mbw, int mbh, unsigned *d_progress /* one word per macroblock row */, uint8_t *d_strips /* 32 bytes per macroblock */,
                      unsigned *d_err, hipStream_t s);
const imv_t *k_final_imv(const frame_ctx_t *h_ctx); /* where the last selection iteration leaves the field */
void k_launch_imv_to_mbi(const frame_ctx_t *h_ctx, int mbw, int row0, int row1, hipStream_t s); // whole-sample field -> records, for the two-kernel (8x8 transform) path
void k_launch_subpel(const frame_ctx_t *h_ctx, int mbw, int row0, int row1, hipStream_t s);
// gate_done (may be null): the reference picture's band deblocker may still be running; a wave waits until the band that holds macroblock row r + 2 carries ref_epoch
void k_launch_pmb(const frame_ctx_t *h_ctx, int mbw, int row0, int row1, int refine, const unsigned *gate_done, unsigned ref_epoch, unsigned *d_err,
                  unsigned *d_row_done /* gated launches: one count per macroblock row, + mbw per launch */, hipStream_t s); // fused refinement + inter (4x4 transform)
void k_launch_inter(const frame_ctx_t *h_ctx, int mbw, int row0, int row1, hipStream_t s);
void k_launch_intra_analyse(const frame_ctx_t *h_ctx, int mbw, int mbh, int gate_p, hipStream_t s); /* gate_p: P picture -- only macroblocks whose search cost reaches INTRA_GATE */
void k_launch_intra_diag(const frame_ctx_t *d_ctx, int mbw, int mbh, int diag, hipStream_t s);
void k_launch_deblock_diag(const frame_ctx_t *d_ctx, int mbw, int mbh, int diag, hipStream_t s);
int k_deblock_bands16(int mbh);
 // flags: per-band "has work" words of this picture's set
// d_ip_progress (may be null): intra_p_kernel of the same picture is still running; the band kernel follows its per-row progress words
// d_iband_done (may be null; all-intra pictures): the intra band kernel of the same picture is still running; a band waits for its flags
/* returns the number of workgroups that will count themselves in *d_started */
int k_launch_deblock_bands(const frame_ctx_t *h_ctx, int mbh, int band0, int band1, unsigned *d_err, uint2 *d_gran, unsigned *d_partab, const unsigned *d_ip_progress,
                            const unsigned *d_iband_done, int ib_rows /* rows per intra band */,
                            unsigned *d_band_done /* may be null: DB_DONE_COPIES x {luma, chroma} per band = the picture's epoch once the band is final in memory */,
                            unsigned *d_started /* may be null: counts the workgroups placed */,
                            const unsigned *d_row_done /* may be null: pmb_kernel<GATED> of this picture is still running; per macroblock row, it counts up to row_need */, unsigned row_need,
                            uint8_t *d_ip_strips, unsigned *d_ip_done /* both non-null (with d_ip_progress and d_row_done): the picture's intra macroblock rows run as the launch's
                                                                            leading workgroups; each counts itself in d_ip_done when its records and levels are in memory */,
                            unsigned *d_qpc, unsigned qpc_base /* adaptive quantisation (h_ctx->qp_off): the launch's first workgroup resolves the QP_Y chain and counts the rows in *d_qpc from qpc_base */,
                            unsigned *d_part_cnt /* may be null; 2 words per band, zeroed once: P pictures walk every band as two workgroups, cut at a column where the filter does nothing;
                                                    each part adds one when its lines are in memory (the counts only grow: two per band, plane and launch) */, hipStream_t s);
int k_intra_band_rows(void);
size_t k_deblock_partab_bytes(int mbw, int mbh); // scratch of the band kernel: one parameter word per (edge, segment) of every macroblock
size_t k_deblock_gran_bytes(int mbw, int mbh);
void k_launch_wait_started(const unsigned *d_started, unsigned count, unsigned *d_err, hipStream_t s);
size_t k_deblock_done_bytes(void); /* band-done words of one picture (all copies) */ // the strips between bands: 8-byte {samples, epoch} granules
void k_launch_pad(uint8_t *y, uint8_t *uv, int stride, int vis_w, int vis_h, int W, int H, hipStream_t s);
int k_intra_diags(int mbw, int mbh);
int k_intra_bands(int mbh);
void k_launch_intra_band(const frame_ctx_t *h_ctx, int mbh, uint2 *d_gran, unsigned *d_err, unsigned *d_band_done, hipStream_t s);
/* I pictures, dataflow at 4x4-block granularity: one workgroup per macroblock row; d_gran: 8 granules per macroblock; d_row_done (may be null): one tagged word per row */
void k_launch_intra_rows(const frame_ctx_t *h_ctx, int mbh, uint2 *d_gran, unsigned *d_err, unsigned *d_row_done, hipStream_t s);
int k_launch_csc(int fmt, const uint8_t *p0, const uint8_t *p1, const uint8_t *p2, int s0, int s1, int s2, uint8_t *dy, uint8_t *duv,
                 int vw, int vh, int W, int H, hipStream_t s);
void k_launch_pack(const mb_info_t *d_mbi, const int16_t *d_levels, int nmb, int mbw, unsigned *d_off, mb_info_t *h_mbi, int16_t *h_packed,
                   unsigned *h_hdr, const unsigned *d_err, hipStream_t s);
int k_deblock_diags(int mbw, int mbh);
/* adaptive quantisation: per-macroblock QP offsets from the source's luma variance (oracle: orc_aq_offsets), and -- once a picture's records are
 * final -- the QP_Y of macroblocks that send no mb_qp_delta (7.4.5: that of the macroblock before them), which the deblocker reads (orc_qp_chain) */
void k_launch_aq(const frame_ctx_t *h_ctx, int8_t *d_off, hipStream_t s);
void k_launch_qp_chain(mb_info_t *d_mbi, int nmb, int slice_qp, int slice_mbs, hipStream_t s);
#endif
#endif
