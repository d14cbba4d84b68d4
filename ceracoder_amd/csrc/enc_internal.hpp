// enc_internal.hpp -- what the translation units of the C-ABI shim share (not part of the ABI):
// the handle, the per-picture slot, and the helpers used by more than one of
//   enc_handle.cpp    open / close / setters / statistics / fetch
//   enc_schedule.cpp  the picture pipeline: submit*, the stream schedule of one picture, collect
//   enc_stages.cpp    single-stage entry points (parity tests, probes) and the host-only stages
#ifndef MI355_ENC_INTERNAL_HPP
#define MI355_ENC_INTERNAL_HPP
#include "../../include/mi355enc.h"

#include <hip/hip_runtime.h>

#include <atomic>
#include <condition_variable>
#include <mutex>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <thread>

#include "h264_host.h"
#include "mi355enc_dev.h"

#define HIPCHK(expr)                                                                                 \
    do {                                                                                             \
        hipError_t e_ = (expr);                                                                      \
        if (e_ != hipSuccess) {                                                                      \
            fprintf(stderr, "mi355enc: %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return MI355ENC_ERR_HIP;                                                                 \
        }                                                                                            \
    } while (0)

static const uint8_t k_lambda[52] = {1,  1,  1,  1,  1,  1,  1,  1,  1,  1,  1,  1,  1,  1,  1,  1,  2,  2,
                                     2,  2,  3,  3,  3,  4,  4,  4,  5,  6,  6,  7,  8,  9,  10, 11, 13, 14,
                                     16, 18, 20, 23, 25, 29, 32, 36, 40, 45, 51, 57, 64, 72, 81, 91};

#define NSLOT 3 /* pictures in flight: pipeline_depth + 1 */
#define NSET 3  /* device-side sets of what one picture's stages hand to each other (and to the host): one per picture in flight */
#define SURF_PAD 256 /* bytes past each surface: unaligned-pair loads may touch 4 bytes beyond */


struct slot_t {
    frame_ctx_t *h_ctx;   // pinned
    mb_info_t *h_mbi;     // pinned
    int16_t *h_levels;    // pinned: packed level stream, written by levels_pack_kernel over PCIe (no D2H copy)
    unsigned *h_hdr;      // pinned: [0] blocks in the stream, [1] error word of the band deblocker, [2 + r] first block of macroblock row r
    uint8_t *h_src;              // pinned staging for pictures submitted from pageable host memory (allocated on first use): rows at stride W
    uint8_t *d_src_y, *d_src_uv; // staging for host / unaligned input
    uint8_t *d_raw;              // staging of non-NV12 input before the conversion kernel (allocated on first use)
    hipEvent_t done, gpu_done, ev[12];
    hipEvent_t ev_up;          // the source has arrived (upload stream; only when that is a stream of its own)
    hipEvent_t ev_front;       // the front stream's part of the picture is done (source in place, search + selection + analysis)
    int prof, fused;
    int all_skip;              // the picture is one run of P_Skip macroblocks: written by the host alone, no device work
    uint64_t index;            // position of the picture in the stream
    int is_idr, qp, drop, frame_num, idr_pic_id, rec_index, set;
    int rc_picked;             // rate control booked this picture (rc_pick): its size is reported back (rc_update) or the booking taken back (rc_cancel)
    int64_t pts;
    // entropy coding on the handle's worker thread (pipeline_depth >= 1): the access unit is coded here while the caller submits the next picture
    uint8_t *au; size_t au_len; int au_state; // 0 not submitted to the worker, 1 queued / being coded, 2 coded (au_len 0: did not fit), 3 the hand-over carried an error word
    double au_ms;
    const uint8_t *src_y, *src_uv; int src_stride, force_idr; // what enqueue_picture() was given: a recovery re-enqueues the pictures in flight from here
};

struct mi355enc {
    mi355enc_cfg_t cfg;
    int mbw, mbh, W, H, nmb;
    size_t ysz, csz;
    hipStream_t stream;                  // "back" stream: everything of a picture that needs the picture before it -- fused P stage / intra wavefront, deblocking
    hipStream_t fstream;                 // "front" stream: source upload / conversion, and for P pictures the whole-sample search, the vector selection
                                         // and the gated intra analysis (source against source: nothing of the previous picture's coding is needed, so
                                         // they run beside its deblocking)
    frame_ctx_t *d_ctx, *d_ctx2[NSET];   // one context per picture in flight; d_ctx = d_ctx2[0]
    slot_t *prev_slot;                   // slot of the picture enqueued last
    mb_info_t *d_mbi, *d_mbi_set[NSET];  // record/level sets: the hand-over of picture n overlaps the kernels of n+1 (and n+2)
    int16_t *d_levels, *d_levels_set[NSET];
    hipStream_t cstream;                 // hand-over stream (scan + pack into pinned host memory)
    bool pgate;                          // the fused P stage runs beside the previous picture's deblocking launch, gated per band (pgate_on(), latched at open())
    bool fip_rows;                       // the intra macroblock rows of a P picture ride in its deblocking launch (fip_on(), latched at open())
    uint64_t n_submitted;
    uint64_t sc_sum, sc_force_at; int sc_cnt, sc_prev_skip; // scene-cut recovery: summed cost / number of the P pictures since the last IDR; picture to force
    uint8_t *d_rec_y[2], *d_rec_uv[2], *d_pre_y, *d_pre_uv;
    uint8_t *d_dbrec;     // deblocking records, 64 B per macroblock
    uint8_t *d_idec;      // intra decisions, IDEC_BYTES per macroblock
    uint16_t *d_isad;     // intra analysis SADs, ISAD_PER_MB u16 per macroblock
    uint2 *d_ib_gran;     // the intra band kernel's bottom lines between bands (tagged granules)
    unsigned *d_iband_done; // ... and its per-band completion flags (the band deblocker's gate on IDR pictures)
    hipEvent_t ev_dbI[2];  // [reconstruction buffer]: the deblocking of an IDR picture that ran beside its intra wavefront on the intra stream has finished
    int dbI_busy[2];
    unsigned *d_db_par;   // the band deblocker's table of per-edge parameter words (written by its prologue, read by its movers)
    unsigned *d_db_part;  // ... and, per band and plane, the count of band parts that have finished (P pictures walk every band as two workgroups: k_deblock.hip, "the cut"); null: bands are walked whole (MI355ENC_NO_SPLIT)
    unsigned *d_db_done;  // per reconstruction buffer: one word per band and plane, = the epoch of the picture whose deblocking of that band is complete
    unsigned *d_row_done;      // per macroblock row: macroblocks the gated P-stage launches have completed so far (the picture's deblocking launch waits for its rows)
    uint32_t pmb_rows_total;   // ... and what each of those counts reaches with the last gated launch enqueued
    uint32_t db_started_total; // workgroups of all band-deblocking launches so far (the device counts them as they are placed: d_progress[1])
    uint32_t qpc_total;        // macroblock rows whose QP_Y chain the deblocking launches have resolved so far (adaptive quantisation; the device counts them: d_progress[3])
    uint32_t ip_done_total;    // intra macroblock rows of all fused launches so far (the device counts them as they complete: d_progress[2])
    uint32_t rec_epoch[2]; // ... and the epoch those words carry once the buffer's picture is done (0: no flags for it)
    uint2 *d_db_gran;     // strips between deblocking bands, as epoch-tagged granules (never cleared)
    unsigned *d_progress; // [0] the sticky error word of the persistent kernels (bounded spins report here), [1] workgroups of band-deblocking launches placed
    unsigned *d_off;      // per-macroblock block offsets of the packed stream (scan kernel -> pack kernel)
    uint16_t *d_surf[NSET]; // SAD surfaces of the motion search, SURF_U16 per macroblock; one set per picture in flight: the front stages of picture n+1 (n+2) run beside the back stages of n
    imv_t *d_imv[NSET][3];   // whole-sample vector fields (search result / selection iterations alternate), per set
    uint8_t *d_idec2[NSET];  // intra decisions per set (d_idec = set 0)
    int8_t *d_qp_off[NSET];  // adaptive quantisation: QP offset per macroblock, per set (null unless cfg.aq_mode)
    uint8_t *d_psrc[2];   // padded source luma of the last two coded pictures: the search runs source against source
    int psrc_cur;         // which of them holds the last coded picture
    unsigned *d_ip_progress; // intra macroblocks of P pictures: one progress word per macroblock row (epoch-tagged, never cleared)
    uint8_t *d_ip_strips;    // ... and the bottom lines they publish for the row below, 32 bytes per macroblock
    uint32_t epoch;
    int islice_rows, stage_slice_rows;   // rows per slice of an I picture (cfg.intra_slices; 0: one slice) / what the single-stage entry points use
    int pslice_rows, slice_dbf, stage_slice_dbf; // ... of a P picture (cfg.slices); disable_deblocking_filter_idc of every slice (cfg.slice_deblock: 0 or 2) / of the single-stage entry points
    hipStream_t ustream;       // host-to-device copies of the source pictures (pipeline_depth >= 1): a copy engine's queue, so that a picture's transfer runs beside the
                               // previous picture's search instead of in front of this one's; nullptr: the front stream carries them
    hipStream_t istream;       // intra_p_kernel of a P picture: beside prep + the band deblocker, which follows it row by row
    hipEvent_t ev_pmb;         // the fused P stage of the picture is done
    slot_t slot[NSLOT];
    int head, tail, pending;
    int cur, have_ref, frames_since_idr, idr_count, last_collected_rec;
    slot_t *last_slot;
    hipGraphExec_t g_intra[NSET], g_deblock[NSET]; // per context
    h264_writer_t *writer;
    rc_state_t rc;
    std::atomic<uint32_t> want_bps;
    std::atomic<int> fixed_qp, fixed_drop;
    mi355enc_stats_t st;
    double ms_open;
    uint64_t n_skip_pictures;
    // Degradation ladder of the device-side waits (collect(): recover()).  0: kernels may wait on the device for other kernels' progress
    // (what exclusive_device and a single encoder per process allow); 1: kernels run in stream order, the only waits left are those between the
    // workgroups of ONE persistent launch (bands of the intra wavefront / the deblocker); 2: one launch per wavefront step, no wait on the device at all.
    int safe_level;
    uint32_t n_recoveries, last_error_word;
    // the entropy-coding worker (started by open() when pipeline_depth >= 1)
    std::thread wk;
    std::mutex wk_mu;
    std::condition_variable wk_cv, wk_done_cv;
    int wk_q[NSLOT + 1], wk_qh, wk_qt;
    bool wk_stop, wk_on;
    size_t au_cap;
    // staging helpers (pageable host input at pipeline_depth >= 1): the pieces of a picture are copied into the slot's pinned buffer by the caller and
    // two helper threads side by side, each piece transferred as soon as it is staged; submit() returns when all of them are on their way
    struct stage_job { const uint8_t *src; uint8_t *dst, *dev; int src_stride, rows, width; size_t dst_stride; };
    std::thread stg_th[2];
    std::mutex stg_mu;
    std::condition_variable stg_cv, stg_done_cv;
    stage_job stg_job[8];
    int stg_n, stg_next, stg_done, stg_err;
    unsigned long long stg_gen;
    bool stg_stop, stg_on;
};
void stage_helper(mi355enc_t *h);

static inline double now_ms() {
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}
static inline unsigned *err_word(const mi355enc_t *h) { return h->d_progress; }

// enc_handle.cpp
extern std::atomic<int> g_open_encoders;
bool exclusive_device(const mi355enc_t *h);
bool pgate_on(int nmb);
bool fip_on(int nmb);
bool overlap_allowed(const mi355enc_t *h);
int sync_compute(mi355enc_t *h);
bool host_range_pinned(const void *p, size_t bytes); // inside memory handed out by mi355enc_host_alloc()
// enc_schedule.cpp
int run_intra(mi355enc_t *h, int ci, const frame_ctx_t *hc, unsigned *band_done = nullptr);
int run_deblock(mi355enc_t *h, int ci, const frame_ctx_t *hc, hipStream_t st, const unsigned *ip_progress, const unsigned *iband_done = nullptr,
                unsigned *band_done = nullptr, bool after_gated_pmb = false, unsigned row_need = 0, bool fused_ip = false);
void fill_ctx(mi355enc_t *h, frame_ctx_t *c, int qp, int drop, int idr, int set = 0);
int enqueue_picture(mi355enc_t *h, slot_t *s, const uint8_t *src_y, const uint8_t *src_uv, int src_stride, int64_t pts, int force_idr);
void entropy_worker(mi355enc_t *h);
int upload_and_convert(mi355enc_t *h, slot_t *s, int fmt, const uint8_t *const planes[3], const int strides[3], hipStream_t up);
#endif
