/*
 * include/mi355enc.h -- C ABI of the MI355X-native H.264 encoder (libmi355enc.so).
 *
 * This is the drop-in boundary under ceracoder's GStreamer graph.  It replaces what the
 * reference obtains from the third-party `x264enc` element named in its pipeline text
 * (/root/reference/pipeline/generic/x264_superfast_camlink:5,
 *  /root/reference/pipeline/generic/x264_superfast_v4l_mjpeg_720p30:6,
 *  /root/reference/bindings/typescript/src/pipeline/generic-builder.ts:50-55),
 * instantiated by gst_parse_launch at /root/reference/src/io/pipeline_loader.c:59.
 * The reference has no FFI of its own for this path: the binding a maintainer adds is the
 * GStreamer element in ceracoder_amd/csrc/gstmi355h264enc.c, which
 * calls exactly these entry points.
 *
 * Conventions follow the reference's C modules (int return, 0 = ok, negative = error,
 * message on stderr; cf. /root/reference/src/gst/encoder_control.h:42-50,
 * /root/reference/src/net/srt_client.c:40-55): no exceptions, no abort(), plain pointers
 * and sizes only.  There is NO CPU fallback: without a usable HIP device
 * mi355enc_open() fails with MI355ENC_ERR_NO_DEVICE.
 */
#ifndef MI355ENC_H
#define MI355ENC_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI355ENC_ABI_VERSION 4

enum {
    MI355ENC_OK = 0,
    MI355ENC_ERR_ARG = -1,        /* bad argument / unsupported geometry            */
    MI355ENC_ERR_NO_DEVICE = -2,  /* no HIP device, or device-id out of range       */
    MI355ENC_ERR_HIP = -3,        /* a HIP runtime call failed (text on stderr)     */
    MI355ENC_ERR_NOMEM = -4,
    MI355ENC_ERR_OVERFLOW = -5,   /* caller's output buffer too small               */
    MI355ENC_ERR_STATE = -6,      /* call order violated (e.g. collect with nothing pending) */
};

typedef struct mi355enc mi355enc_t; /* opaque; owns all device memory, streams, graphs */

typedef struct {
    int width, height;        /* visible picture, even, 16..8192                          */
    int fps_num, fps_den;
    int gop;                  /* IDR period; x264enc `key-int-max` (pipelines pass 60)     */
    int me_range;             /* full-search radius in integer pels, 1..16                 */
    uint32_t bitrate_bps;     /* initial target; what encoder_control.c:53 later rewrites  */
    int device_id;            /* HIP device ordinal (one stream per GPU)        */
    int fixed_qp;             /* >= 0: constant QP, rate control off (tests, bench); -1: CBR */
    int qp_min, qp_max;       /* rate-control clamp; 0,0 -> defaults 10..51                */
    int pipeline_depth;       /* 0: encode() returns this frame's AU; 1: host entropy coding
                                 of frame n overlaps device work of frame n+1; 2: ... and the
                                 device never waits for the host (three pictures in flight; rate
                                 control sees a picture's size two pictures later, scene-cut
                                 recovery lands one picture later than at depth 0/1)       */
    int profile_events;       /* k > 0: bracket the kernel stages of every k-th picture (and every IDR) with HIP events
                                 for the stage statistics; an event record costs ~5 us of queue time, so k = 1 slows
                                 the stream by several per cent */
    int use_graphs;           /* 1: replay the per-picture launch sequence as a hipGraph  */
    int keep_prefilter;       /* 1: keep a copy of the picture before deblocking (tests)  */
    int transform8x8;         /* 1: High-profile stream, P macroblocks use the 8x8 transform; 0 (default): Constrained Baseline */
    int i4x4;                 /* 1 (default): try Intra_4x4 besides Intra_16x16 in I pictures */
    int subpel;               /* 1 (default): half- then quarter-sample refinement after the integer search */
    int deblock_mode;         /* 0: persistent band kernel (x+y order, three waves per macroblock row; boundary strengths in its prologue);
                                 1: one launch per x+2y wavefront (plain form, kept as a cross-check) */
    int intra_in_p;           /* 1 (default): macroblocks of P pictures may be coded intra (Intra_16x16) for uncovered regions and partial scene
                                 changes: decided in the fused P stage from the open-loop intra analysis, reconstructed by a short dependent pass
                                 after it.  2: Intra_4x4 as well (with i4x4) -- the part of x264 superfast's partition search that survives,
                                 `partitions i8x8,i4x4`; rate-distortion neutral on the synthetic clips (-0.2 % / +0.4 % BD-rate), ten dependent
                                 sub-steps per such macroblock.  0: P pictures hold inter macroblocks only */
    int cavlc_threads;        /* host threads that code the slice (ranges of macroblock rows, concatenated bit-exactly into the
                                 same single slice); 1: the calling thread only; 0 (default, like x264enc's threads=0): chosen
                                 from the machine -- a quarter of the online CPUs, between 1 and 8 (1 for pictures under
                                 1000 macroblocks, where waking workers costs more than it saves); mi355enc_stats_t reports it */
    int intra_mode;           /* 0 (default): one persistent launch, a workgroup per macroblock row, macroblocks overlapping at 4x4-block
                                 granularity (dataflow); 1: one launch per anti-diagonal replayed as a hipGraph (plain form, kept as a
                                 cross-check); 2: the lock-step band kernel of rounds 1-2 (x + y order, one barrier per step; kept for A/B) */
    int vbv_ms;               /* rate control's buffer model in ms of stream at the setpoint (default 600: x264enc's vbv-buf-capacity): an IDR
                                 picture is planned at most half of it, and P pictures are all-skip while the bucket is nearly full */
    int scenecut;             /* 1 (default, like x264's scenecut): when the summed motion cost of a P picture exceeds three times
                                 the mean of the P pictures since the last IDR (at least two of them), the picture two
                                 positions later is coded as IDR -- recovery within two pictures instead of a GOP, with no wait
                                 on the device (the sum arrives with the picture's hand-over).  0: IDR only every `gop`
                                 pictures or on request */
    int exclusive_device;     /* 0 (default): other processes may use the same GPU.  1: this encoder has the GPU to itself (one
                                 stream per GPU, BASELINE configs[4]; what bench.py sets): a P picture's fused stage is launched
                                 beside the deblocking of the picture before it and its workgroups wait ON the device for the
                                 bands they read -- a chip full of waiting workgroups.  With another process on the same GPU
                                 such launches can keep each other's kernels off the chip (seen: one of two processes ran into
                                 the bound of its wait), so it is opt-in.  Same stream either way */
    int aq_mode;              /* 0 (default): one QP per picture.  1: adaptive quantisation -- a QP offset of -4 .. +4 per macroblock from the luma
                                 variance of its source samples (flat areas finer, busy texture coarser), coded with mb_qp_delta.  The QP_Y of
                                 macroblocks that send no mb_qp_delta is that of the macroblock before them (7.4.5), which the deblocker reads: a
                                 chain over the whole picture, resolved row by row by the first workgroup of the deblocking launch behind the same
                                 progress its bands wait for (r03; about 2 % fewer frames/s).  With intra_in_p = 2 the
                                 kernels of a picture run in stream order */
    int single_stream;        /* 0 (default): four HIP streams per encoder (front / main / intra / hand-over), so that a picture's independent stages and
                                 consecutive pictures overlap.  1: everything on ONE stream, in order -- for many encoders on one GPU (several in a
                                 process, or many processes): the GPU has a handful of hardware queues, and 8 encoders x 4 streams made the driver
                                 time-slice them (8 streams in one process: 409 frames/s in ALL; with single_stream each encoder keeps one queue busy) */
    int intra_slices;         /* slices per I picture, each its own NAL unit (x264enc: the `slices` option of libx264).  0 (default): about 17
                                 macroblock rows per slice, at most 8 slices (1080p: 4, 720p: 2, below 34 rows: 1).  Intra prediction cannot cross a
                                 slice boundary (6.4.8), so the slices of a picture are independent chains for the intra wavefront: an IDR picture's
                                 reconstruction takes about 1/n of the time (the deblocking filter still runs across the boundaries); the cost is the
                                 prediction lost along n - 1 rows, +0.5 % on the IDR pictures' bytes at 1080p with 4.  P pictures are one slice */
    int partitions;           /* 0 (default): every inter macroblock is one 16x16 partition (what x264's superfast preset searches).  1: P macroblocks may be
                                 split into 16x8, 8x16 or 8x8 partitions (x264enc: analyse / `partitions`): every partition chooses among the vectors the
                                 macroblock's sub-sample refinement visits; rate-distortion neutral on the test clips, about 3 % fewer frames/s.  Not with
                                 transform8x8, deblock_mode 1 or adaptive quantisation's in-order cases */
    int profile_overlap;      /* with profile_events: 0 (default) a sampled picture runs its stages strictly in order, so that every timer is one kernel
                                 alone (the picture costs the stream about two periods).  1: sampled P pictures keep the free-running schedule -- the
                                 event pairs sit on the streams the kernels are launched on, and a launch that waits on the device for another kernel's
                                 rows is timed with that wait, as a kernel trace would show it.  IDR pictures are always sampled in order */
    int i8x8;                 /* 0 (default): off.  1: with transform8x8, the macroblocks of I pictures may be Intra_8x8 (x264enc: dct8x8 brings the transform
                                 and the intra type together) at picture quantisers up to 37; needs intra_mode 0 (the macroblock above-right has to be
                                 complete), otherwise ignored.  Measured at 1080p: IDR pictures 1.4 - 3.9 % smaller at QP 22 - 34 at equal PSNR; an
                                 Intra_8x8 macroblock is four dependent 8x8 blocks on the intra wavefront (its neighbour to the right starts half a
                                 macroblock behind), so a stream with key-int 60 runs 1 - 3 % slower, an all-intra one by a third */
    int slices;               /* slices per P picture, each its own NAL unit (r04; what x264enc's threads do to a picture behind
                                 /root/reference/pipeline/generic/x264_superfast_camlink:5).  0 (default): as many as an I picture gets by default -- about 17
                                 macroblock rows each (1080p: 4, 720p: 2, 2160p: 7, below 34 rows: 1).  1: one slice.  n > 1: n slices of ceil(rows / n) rows.
                                 Motion-vector prediction, the P_Skip inference, intra prediction, nC and QP_Y,PRED stop at a slice's first row (6.4.8): on
                                 panning content the rows below a seam cannot be P_Skip (the inferred vector is zero there) -- Bjontegaard rate +2.4 % (S2 clip)
                                 / +3.9 % (panning S4) at 1080p with 5 slices, nothing on still content (profiles/r04_rd_slices_*.txt) */
    int slice_deblock;        /* 1 (default): the deblocking filter stops at slice boundaries (disable_deblocking_filter_idc 2, in the slices of I and P
                                 pictures alike; slice heights are then multiples of four macroblock rows, the height of the deblocker's bands): every slice is
                                 an independent dependency chain for the deblocking launch -- the launch that sets the picture period -- whose length falls from
                                 columns + rows - 1 steps to columns + rows per slice - 1 (1080p: the launch alone 146 -> 85 us; +17...20 % frames/s; the seam rows
                                 lose 0.0 - 0.2 dB at QP 36 - 42, nothing measurable below).  0: the filter runs across slice boundaries (idc 0) */
} mi355enc_cfg_t;

typedef struct {
    uint64_t frames, idr_frames, bytes;   /* pictures collected, IDR pictures among them, bytes produced */
    uint32_t last_qp, last_bytes, target_bps;
    /* accumulated device time per stage in ms and sample counts (profile_events > 0) */
    double ms_me, ms_inter, ms_intra, ms_deblock, ms_total_gpu;
    double ms_subpel;
    uint64_t n_me, n_inter, n_intra, n_deblock;
    double ms_entropy;        /* host CAVLC wall time */
    double ms_wait;           /* host time blocked on the device */
    uint64_t n_total_gpu;     /* pictures sampled into ms_total_gpu */
    double ms_deblock_idr;    /* the part of ms_deblock / n_deblock that came from IDR pictures */
    uint64_t n_deblock_idr;
    uint32_t cavlc_threads;   /* host threads in use for entropy coding (cfg.cavlc_threads resolved) */
    uint32_t last_drop;       /* drop level of the last collected picture (0: none; 1 .. 12: the ladder below QP 51; 255: all-skip picture) */
    double ms_select;         /* P pictures: the ME_ITERS vector-selection iterations (not part of ms_me), n_me samples */
    double ms_analyse_p;      /* P pictures: intra analysis of the gated macroblocks, n_inter samples (also contained in ms_inter) */
    double ms_intra_p;        /* P pictures: reconstruction of the intra macroblocks, n_inter samples (also contained in ms_inter) */
    uint64_t skip_pictures;   /* pictures coded as one P_Skip run (rate control's last resort) */
    double ms_open;           /* wall time mi355enc_open() took (device selection, allocations, stream creation): must stay far below
                                 the 1 s tick of the reference's stall watchdog, /root/reference/src/ceracoder.c:152-200; survives reset_stats */
    uint32_t recoveries;      /* times a bounded device-side wait ran out and the pictures in flight were re-encoded from an IDR picture
                                 (survives reset_stats); last_error_word: which wait it was (text on stderr); safe_level: 0 kernels may wait
                                 on the device for each other, 1 stream order only, 2 one launch per wavefront step (no device-side wait left) */
    uint32_t last_error_word, safe_level;
    uint64_t pinned_inputs;   /* pictures submitted from mi355enc_host_alloc() memory (DMA'd in place, no staging copy) */
} mi355enc_stats_t;

/* Fill cfg with the defaults of the element (gop 60, me_range 16, 2048 kbit/s like x264enc). */
void mi355enc_default_cfg(mi355enc_cfg_t *cfg, int width, int height, int fps_num, int fps_den);

int mi355enc_open(const mi355enc_cfg_t *cfg, mi355enc_t **out);
void mi355enc_close(mi355enc_t *h);

/* Thread-safe against a concurrent encode (atomic store, latched at the next picture);
 * never touches the GPU.  Mirrors g_object_set(elem, "bps", v) at encoder_control.c:53. */
int mi355enc_set_bitrate(mi355enc_t *h, uint32_t bps);
uint32_t mi355enc_get_bitrate(const mi355enc_t *h);
/* Constant-QP override for the next pictures (tests/bench); -1 returns to rate control. */
int mi355enc_set_fixed_qp(mi355enc_t *h, int qp);
/* With a constant QP: the level of rate control's ladder below QP 51 for the next P pictures (tests): 0 off .. 12: P macroblocks
 * whose prediction error is small enough carry no residual / take the P_Skip vector; 255: whole pictures as one P_Skip run. */
int mi355enc_set_fixed_drop(mi355enc_t *h, int drop);

/* Synchronous: one NV12 picture in host memory -> one Annex-B access unit
 * (SPS+PPS precede every IDR).  Borrowed input, caller-owned output. */
int mi355enc_encode(mi355enc_t *h, const uint8_t *y, int y_stride, const uint8_t *uv, int uv_stride,
                    int64_t pts, int force_idr, uint8_t *out, size_t out_cap, size_t *out_len,
                    int *is_keyframe);

/* Pinned host memory for input pictures.  mi355enc_submit() copies a picture from ordinary (pageable) memory into a pinned staging buffer
 * first -- one pass of the calling thread over the picture, 3.1 MB at 1080p.  A picture that lies in memory obtained here is transferred
 * from where it is, asynchronously: submit() returns at once, and the memory must then stay untouched until the matching collect()
 * (the element offers such memory to its upstream through the ALLOCATION query, so a source writes its pictures straight into it).
 * Process-wide; usable on every device.  NULL when there is no HIP device or no memory. */
void *mi355enc_host_alloc(size_t bytes);
void mi355enc_host_free(void *p);

/* Split form.  submit() enqueues all device work of a picture and returns; collect()
 * entropy-codes the oldest submitted picture.  At most pipeline_depth+1 pictures may be
 * outstanding.  submit_device() takes planes already resident in this GPU's memory
 * (the bench's timed region; they must stay valid until the matching collect()). */
int mi355enc_submit(mi355enc_t *h, const uint8_t *y, int y_stride, const uint8_t *uv, int uv_stride,
                    int64_t pts, int force_idr);
/* Raw input formats other than NV12 are converted on the device (no `videoconvert` hop): I420 (planes Y, U, V), and
 * packed 4:2:2 YUY2 / UYVY (plane 0 only; chroma rows are averaged pairwise with rounding to reach 4:2:0). */
enum { MI355ENC_FMT_NV12 = 0, MI355ENC_FMT_I420 = 1, MI355ENC_FMT_YUY2 = 2, MI355ENC_FMT_UYVY = 3 };
/* like mi355enc_submit, from host memory in `fmt`; planes[]/strides[]: as many entries as the format has planes */
int mi355enc_submit_fmt(mi355enc_t *h, int fmt, const uint8_t *const planes[3], const int strides[3], int64_t pts, int force_idr);
/* conversion stage alone (tests): writes the coded-size NV12 surfaces (16*mbw x 16*mbh luma, then interleaved chroma) */
int mi355enc_stage_csc(mi355enc_t *h, int fmt, const uint8_t *const planes[3], const int strides[3], uint8_t *out_y, uint8_t *out_uv);
int mi355enc_submit_device(mi355enc_t *h, const void *d_y, int y_stride, const void *d_uv,
                           int uv_stride, int64_t pts, int force_idr);
int mi355enc_pending(const mi355enc_t *h);
int mi355enc_collect(mi355enc_t *h, uint8_t *out, size_t out_cap, size_t *out_len, int *is_keyframe,
                     int64_t *pts, int *qp);

/* Fault injection (tests): behaves as if a kernel's bounded wait on the device had just run out with error word `code` (> 0): waits
 * for the device to drain, then sets the sticky word every waiting kernel reports through.  The pictures submitted next come back from
 * collect() through the recovery path: re-encoded in stream order, starting with an IDR picture; stats.recoveries counts it. */
int mi355enc_debug_trip_wait(mi355enc_t *h, unsigned code);

int mi355enc_get_stats(mi355enc_t *h, mi355enc_stats_t *st);
void mi355enc_reset_stats(mi355enc_t *h);
size_t mi355enc_max_au_bytes(const mi355enc_t *h);
const char *mi355enc_strerror(int code);
int mi355enc_abi_version(void);

/* ---- inspection of the last collected picture (parity tests) ----------------------
 * Copies device state to host: coded-size planes (stride = 16*mb_width), the 16-byte
 * per-macroblock records and the 408 int16 levels per macroblock (layout: DESIGN.md).
 * MI355ENC_FETCH_MBINFO returns the records as they were handed to the entropy coder.  With adaptive quantisation (aq_mode 1) the `qp` byte of a macroblock that sends
 * no mb_qp_delta (P_Skip, or no coded block and not Intra_16x16) is NOT defined there: the deblocking launch's QP_Y chain rewrites it on the device (7.4.5: the QP_Y of the
 * macroblock before it) while the hand-over copies the records, and the entropy coder never reads it -- take `qp` only from macroblocks that send a delta. */
enum { MI355ENC_FETCH_RECON_Y = 0, MI355ENC_FETCH_RECON_UV = 1, MI355ENC_FETCH_PREFILTER_Y = 2,
       MI355ENC_FETCH_PREFILTER_UV = 3, MI355ENC_FETCH_MBINFO = 4, MI355ENC_FETCH_LEVELS = 5 };
int mi355enc_fetch(mi355enc_t *h, int what, void *dst, size_t dst_bytes);
int mi355enc_mb_width(const mi355enc_t *h);
int mi355enc_mb_height(const mi355enc_t *h);

/* ---- single-stage entry points (parity tests; same kernels the encoder launches) ----
 * All planes are host pointers to coded-size (multiple-of-16) surfaces with stride 16*mbw;
 * mbinfo is mbw*mbh 16-byte records, levels mbw*mbh*408 int16. */
/* Whole-sample motion search: SAD surfaces (35 x 36 uint16 per macroblock: row dy+16, column dx+16; the 33 x 33 upper-left
 * part is the +-16 search range; surf_out may be NULL) and the first selection, 8 bytes per macroblock {i16 mvx, mvy
 * (quarter-sample units, multiples of 4); u16 sad, bits}. */
int mi355enc_stage_me(mi355enc_t *h, const uint8_t *cur_y, const uint8_t *ref_y, int qp, uint16_t *surf_out, void *imv_out);
/* One more selection over the surfaces, bits charged against the median of the neighbours' vectors in imv_in. */
int mi355enc_stage_me_select(mi355enc_t *h, const uint16_t *surf, const void *imv_in, int qp, void *imv_out);
/* the same iteration in the form the encoder's later passes use (macroblocks whose predictors are unchanged against imv_prev -- the field imv_in was selected from -- are copied) */
int mi355enc_stage_me_select_next(mi355enc_t *h, const uint16_t *surf, const void *imv_in, const void *imv_prev, int qp, void *imv_out);
/* Two-kernel form of the P stage (High-profile path): refine the vectors in mbinfo (mvx, mvy in quarter-sample units, cost) in place ... */
int mi355enc_stage_subpel(mi355enc_t *h, const uint8_t *cur_y, const uint8_t *ref_y, int qp, void *mbinfo_inout);
/* ... and prediction, residual (4x4 or 8x8 transform), reconstruction for the vectors in mbinfo */
int mi355enc_stage_inter(mi355enc_t *h, const uint8_t *src_y, const uint8_t *src_uv, const uint8_t *ref_y,
                         const uint8_t *ref_uv, int qp, void *mbinfo_inout, uint8_t *rec_y, uint8_t *rec_uv,
                         int16_t *levels);
/* The fused P-macroblock stage the encoder runs, from the final whole-sample field `imv` and the surfaces: predictor estimates,
 * skip probe, sub-sample refinement (if `refine`), intra-or-inter against `idec` (32 bytes per macroblock as written by
 * mi355enc_stage_intra_analyse; NULL: inter only), residual with coefficient decimation; `drop`: 0 .. 12.  With run_intra_p the
 * macroblocks decided intra are reconstructed afterwards (intra_p_kernel); without, they only carry type and modes. */
int mi355enc_stage_pmb(mi355enc_t *h, const uint8_t *src_y, const uint8_t *src_uv, const uint8_t *ref_y, const uint8_t *ref_uv,
                       int qp, int drop, int refine, const void *imv, const uint16_t *surf, const void *idec, int run_intra_p,
                       void *mbinfo_out, uint8_t *rec_y, uint8_t *rec_uv, int16_t *levels);
/* I picture: analysis + reconstruction wavefront; `drop`: rate control's ladder for I pictures, 0 off .. 12 (Intra_16x16 only, and a
 * macroblock's luma / chroma levels are not sent when their magnitudes sum to no more than the level's threshold) */
int mi355enc_stage_intra(mi355enc_t *h, const uint8_t *src_y, const uint8_t *src_uv, int qp, int drop, void *mbinfo_out,
                         uint8_t *rec_y, uint8_t *rec_uv, int16_t *levels);
/* open-loop intra analysis only: 152 uint16 per macroblock {i16[4], chroma[4], i4[16][9]}, 0xFFFF = mode unavailable; and
 * (idec_out may be NULL) the decisions taken from them at `qp`: 32 bytes per macroblock {u8 modes4[16] by luma4x4BlkIdx;
 * u8 mode16, chroma mode, use_i4, 0; u32 cost (luma + chroma), cost_luma, 0} */
int mi355enc_stage_intra_analyse(mi355enc_t *h, const uint8_t *src_y, const uint8_t *src_uv, int qp, uint16_t *isad_out, void *idec_out);
int mi355enc_stage_deblock(mi355enc_t *h, uint8_t *rec_y, uint8_t *rec_uv, const void *mbinfo);
/* Time `iters` back-to-back launches of one stage on the handle's stream with HIP events;
 * stage: 0 ME, 1 inter, 2 intra (whole wavefront), 3 deblock (whole wavefront), 4 sub-sample refinement,
 * 5 / 6 / 7 input conversion from I420 / YUY2 / UYVY, 8 one vector-selection iteration, 9 fused P stage, 10 intra macroblocks of a P picture.
 * Uses whatever the handle's surfaces currently hold.  Returns average ms per launch. */
int mi355enc_time_stage(mi355enc_t *h, int stage, int iters, double *avg_ms);


/* ---- host-only stages (no device needed; what collect() runs once the packed hand-over has landed) ----
 * SPS+PPS, and one CAVLC slice NAL from macroblock records + levels.  *out_len = bytes. */
int mi355enc_host_write_headers(int width, int height, int fps_num, int fps_den, int transform8x8, uint8_t *out, size_t out_cap,
                                size_t *out_len);
int mi355enc_host_write_slice(int mb_width, int mb_height, int is_idr, int frame_num, int idr_pic_id, int slice_qp, int transform8x8,
                              const void *mbinfo, const int16_t *levels, uint8_t *out, size_t out_cap, size_t *out_len);
/* process-wide, for the two host stage functions below: I pictures are written as slices of `rows` macroblock rows (0, the default: one slice) */
void mi355enc_host_set_slice_rows(int rows);
/* ... P pictures as slices of `rows` rows (0: one slice), and the disable_deblocking_filter_idc every slice header (I and P) carries: 0 or 2 */
void mi355enc_host_set_p_slices(int rows, int dbf_idc);
/* the single-stage entry points of a handle work on one-slice pictures unless told otherwise (the encoder itself follows cfg.intra_slices) */
int mi355enc_stage_set_slice_rows(mi355enc_t *h, int rows);
int mi355enc_slice_rows(const mi355enc_t *h); /* rows per slice of this handle's I pictures (0: one slice) */
int mi355enc_p_slice_rows(const mi355enc_t *h); /* ... and of its P pictures */
int mi355enc_stage_set_slice_deblock(mi355enc_t *h, int idc); /* the single-stage entry points: disable_deblocking_filter_idc of the picture's slices, 0 (default) or 2 */
/* the same slice through the packed hand-over format and `threads` row-parallel host threads (bit-identical result) */
int mi355enc_host_write_slice_packed(int mbw, int mbh, int is_idr, int frame_num, int idr_pic_id, int qp, int t8, int threads, const void *mbinfo,
                                     const int16_t *levels, uint8_t *out, size_t cap, size_t *out_len);
/* One CAVLC residual block (9.2) through the slice writer's block coder, for known-answer tests: coef in scan order, maxnum 16
 * (whole 4x4 block), 15 (AC of Intra16x16 / chroma) or 4 (chroma DC); nC as 9.2.1 derives it.  Bits MSB-first into out
 * (cap >= 64 bytes); returns the number of bits, negative on error. */
int mi355enc_host_cavlc_block(const int16_t *coef, int maxnum, int nC, uint8_t *out, size_t cap);
/* Rate-control model on its own: per picture one pick (QP and, below QP 51, the drop level: 0 .. 12, 255 = all-skip picture)
 * and -- possibly one picture later, as with pipeline_depth 1 -- one update with the bytes it produced, in the same order.
 * rc is an opaque block of MI355ENC_RC_BYTES bytes owned by the caller. */
#define MI355ENC_RC_BYTES 512
void mi355enc_rc_init(void *rc, double fps, int gop, uint32_t bps, int qp_min, int qp_max);
void mi355enc_rc_set_bitrate(void *rc, uint32_t bps);
void mi355enc_rc_pick(void *rc, int is_idr, int *qp, int *drop);
void mi355enc_rc_update(void *rc, int is_idr, int qp, int drop, size_t bytes);

#ifdef __cplusplus
}
#endif
#endif
