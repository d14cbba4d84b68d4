/* Host-side bitstream writer of libmi355enc (internal). */
#ifndef H264_HOST_H
#define H264_HOST_H
#include <stddef.h>
#include <stdint.h>

#include "mi355enc_dev.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct h264_writer h264_writer_t;
h264_writer_t *h264_writer_new(int mbw, int mbh, int transform8x8);
void h264_writer_free(h264_writer_t *w);
size_t h264_max_au_bytes(int mbw, int mbh);
/* SPS + PPS (Annex B).  Returns bytes written, 0 if `cap` is too small. */
size_t h264_write_headers(uint8_t *out, size_t cap, int width, int height, int fps_num, int fps_den, int transform8x8);
/* One slice NAL covering the whole picture.  Returns bytes written, 0 if out of room. */
size_t h264_write_slice(h264_writer_t *w, uint8_t *out, size_t cap, int is_idr, int frame_num, int idr_pic_id,
                        int slice_qp, const mb_info_t *mbi, const int16_t *levels);
/* same, from the packed level stream the device writes (k_handover.hip, levels_pack_kernel) */
size_t h264_write_slice_packed(h264_writer_t *w, uint8_t *out, size_t cap, int is_idr, int frame_num, int idr_pic_id,
                               int slice_qp, const mb_info_t *mbi, const int16_t *packed);

/* Row-parallel coding of the same single slice: `threads` workers (the caller is one of them) code ranges of macroblock
 * rows concurrently and the result is concatenated bit-exactly.  row_off[r] = index of the first 32-byte block of
 * macroblock row r in the packed stream (written by the device's scan kernel).  threads <= 1: everything on the caller. */
int h264_writer_set_threads(h264_writer_t *w, int threads);
/* I pictures written from now on are cut into slices of `rows` macroblock rows, one NAL unit each (0: one slice); P pictures stay one slice */
void h264_writer_set_slice_rows(h264_writer_t *w, int rows);
void h264_writer_set_p_slices(h264_writer_t *w, int rows, int dbf_idc); /* P pictures: rows per slice (0: one slice); disable_deblocking_filter_idc of every slice header (I and P): 0 or 2 */
size_t h264_write_slice_packed_rows(h264_writer_t *w, uint8_t *out, size_t cap, int is_idr, int frame_num, int idr_pic_id,
                                    int slice_qp, const mb_info_t *mbi, const int16_t *packed, const uint32_t *row_off);

/* one CAVLC residual block (maxnum 16, 15 or 4; nC as 9.2.1 derives it, ignored for 4): bits MSB-first into out (cap >= 64), returns their number */
int h264_cavlc_block_bits(const int16_t *coef, int maxnum, int nC, uint8_t *out, size_t cap);

/* Host statement of the device's hand-over (levels_scan_kernel + levels_pack_kernel): dense levels -> packed stream of 32-byte
 * blocks + first block of every macroblock row.  `packed` must hold mbw*mbh*PACK_BLOCKS_MAX*16 int16.  Returns the block count. */
size_t h264_pack_levels(int mbw, int mbh, const mb_info_t *mbi, const int16_t *levels, int16_t *packed, uint32_t *row_off);

/* ---- rate control (ratecontrol.c): one QP (+ below QP 51 a drop level) per picture from a bits/s setpoint ---- */
typedef struct {
    double fps;
    int gop, qp_min, qp_max, vbv_ms;
    double target_bps;
    double cplx_i, cplx_p;     /* bits * qstep(virtual QP) of recent I / P pictures */
    double gop_bits;           /* what the current GOP may still spend (planned sizes of pictures in flight already taken off) */
    int gop_left, started;     /* pictures of the current GOP still to be picked    */
    double vbv;                /* leaky bucket at the setpoint's rate, bits         */
    double last_bits_p, last_target_p; /* the last P picture whose size is known: what it took, what it was given */
    int since_real;            /* P pictures coded as skip runs since the last really coded one */
    double cliff_bits; int cliff_vqp, cliff_age; /* the quantiser at which a P picture last cost several times its target, what it cost, pictures left to remember it */
    int last_vqp_i, last_vqp_p;
    int have_i, have_p;
    double plan[4];            /* planned bits of the pictures picked but not yet updated (three pictures in flight at most) */
    double plan_k[4];          /* ... the factor the setpoint has moved by since they were picked (1: no change) */
    short plan_gap[4];         /* ... how many all-skip pictures preceded them */
    double cplx_gap;           /* the same, averaged over the samples the P tracker holds: a picture after a skip run costs more than one after a coded picture */
    short plan_vqp[4];         /* ... and the virtual quantiser they were given (P pictures; 0x7FFF: an IDR or all-skip picture) */
    unsigned plan_gop[4];      /* ... and the GOP they belong to */
    unsigned gop_serial;       /* the current GOP */
    double prev_rest, carry_used; /* what the previous GOP left (uncapped) when this one was granted, and the carry granted from it: pictures of the previous
                                * GOP whose sizes arrive after the grant settle through the same cap */
    int upd_drop_p, catchup;   /* ladder level of the last P picture whose size is known; pictures still to come whose size is a catch-up transient */
    int known_vqp_p;           /* the virtual quantiser of the last P picture whose size is known */
    unsigned n_pick, n_upd;
    unsigned char plan_reg[4], regime; /* the tracker's regime when a picture was picked: a rise out of the ladder starts a new one, and the sizes of pictures picked before it
                                        * (still on the ladder) no longer move the tracker */
    int cliff_doubt;           /* pictures in a row, coded one step above a remembered cliff, that came out far below their target: the cliff is tried again after eight */
    double cplx_q; int have_q; /* the P tracker's last value while the stream lived on real quantisers (no ladder level, no all-skip cadence): what a rise of the setpoint out of
                                * the ladder starts from -- on the ladder bits * qstep(virtual QP) says little about what a quantiser below 51 will cost */
} rc_state_t;
void rc_init(rc_state_t *rc, double fps, int gop, uint32_t bps, int qp_min, int qp_max);
void rc_set_vbv(rc_state_t *rc, int vbv_ms);
void rc_set_bitrate(rc_state_t *rc, uint32_t bps);
int rc_pick_qp(rc_state_t *rc, int is_idr);
/* QP and, once qp_max is not enough, the drop level of the ladder below it (P pictures: 0 .. DROP_MAX, DROP_SKIP = all-skip picture).
 * Every rc_pick() is followed (possibly one picture later) by one rc_update() for the same picture, in the same order. */
void rc_pick(rc_state_t *rc, int is_idr, int *qp, int *drop);
void rc_update(rc_state_t *rc, int is_idr, int qp, int drop, size_t bytes);
void rc_cancel(rc_state_t *rc);

#ifdef __cplusplus
}
#endif
#endif
