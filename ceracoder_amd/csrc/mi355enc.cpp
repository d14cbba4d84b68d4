// mi355enc.cpp -- C-ABI shim (include/mi355enc.h) over the HIP kernels: owns device
// surfaces, the stream, the captured launch graphs, the picture pipeline and rate control.
//
// Per picture, on one HIP stream:
//   [H2D source] -> IDR: intra analysis (flat) -> intra wavefront (persistent bands)
//                   P  : me_kernel (SAD surfaces + first selection) -> me_select_kernel x ME_ITERS
//                        -> intra analysis of the badly predicted macroblocks -> pmb_kernel -> intra_p_kernel
//                -> deblock (prep + persistent band kernel)
//   and on a second stream, from the moment records and levels are final: scan + pack into pinned host memory -> event
// and on the host, when the event has fired: CAVLC slice coding (h264_host.c).
// With pipeline_depth = 1 the host codes picture n while the device works on n+1.
#include "../../include/mi355enc.h"

#include <hip/hip_runtime.h>

#include <atomic>
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
    uint8_t *d_src_y, *d_src_uv; // staging for host / unaligned input
    uint8_t *d_raw;              // staging of non-NV12 input before the conversion kernel (allocated on first use)
    hipEvent_t done, gpu_done, ev[12];
    hipEvent_t ev_front;       // the front stream's part of the picture is done (source in place, search + selection + analysis)
    int prof, fused;
    int all_skip;              // the picture is one run of P_Skip macroblocks: written by the host alone, no device work
    uint64_t index;            // position of the picture in the stream
    int is_idr, qp, drop, frame_num, idr_pic_id, rec_index, set;
    int64_t pts;
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
    hipStream_t cstream;                 // copy stream for the D2H hand-over
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
    unsigned *d_db_done;  // per reconstruction buffer: one word per band and plane, = the epoch of the picture whose deblocking of that band is complete
    unsigned *d_row_done;      // per macroblock row: macroblocks the gated P-stage launches have completed so far (the picture's deblocking launch waits for its rows)
    uint32_t pmb_rows_total;   // ... and what each of those counts reaches with the last gated launch enqueued
    uint32_t db_started_total; // workgroups of all band-deblocking launches so far (the device counts them as they are placed: d_progress[1])
    uint32_t rec_epoch[2]; // ... and the epoch those words carry once the buffer's picture is done (0: no flags for it)
    uint2 *d_db_gran;     // strips between deblocking bands, as epoch-tagged granules (never cleared)
    unsigned *d_progress; // [0] the sticky error word of the persistent kernels (bounded spins report here), [1] workgroups of band-deblocking launches placed
    unsigned *d_off;      // per-macroblock block offsets of the packed stream (scan kernel -> pack kernel)
    uint16_t *d_surf[NSET]; // SAD surfaces of the motion search, SURF_U16 per macroblock; one set per picture in flight: the front stages of picture n+1 (n+2) run beside the back stages of n
    imv_t *d_imv[NSET][2];   // whole-sample vector fields (search result / selection iterations alternate), per set
    uint8_t *d_idec2[NSET];  // intra decisions per set (d_idec = set 0)
    uint8_t *d_psrc[2];   // padded source luma of the last two coded pictures: the search runs source against source
    int psrc_cur;         // which of them holds the last coded picture
    unsigned *d_ip_progress; // intra macroblocks of P pictures: one progress word per macroblock row (epoch-tagged, never cleared)
    uint8_t *d_ip_strips;    // ... and the bottom lines they publish for the row below, 32 bytes per macroblock
    uint32_t epoch;
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
};

static double now_ms() {
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

extern "C" {

int mi355enc_abi_version(void) { return MI355ENC_ABI_VERSION; }

const char *mi355enc_strerror(int code) {
    switch (code) {
    case MI355ENC_OK: return "ok";
    case MI355ENC_ERR_ARG: return "invalid argument";
    case MI355ENC_ERR_NO_DEVICE: return "no usable HIP device (this encoder has no CPU fallback)";
    case MI355ENC_ERR_HIP: return "HIP runtime error";
    case MI355ENC_ERR_NOMEM: return "out of memory";
    case MI355ENC_ERR_OVERFLOW: return "output buffer too small";
    case MI355ENC_ERR_STATE: return "call order violated";
    default: return "unknown error";
    }
}

void mi355enc_default_cfg(mi355enc_cfg_t *c, int width, int height, int fps_num, int fps_den) {
    memset(c, 0, sizeof *c);
    c->width = width; c->height = height; c->fps_num = fps_num; c->fps_den = fps_den > 0 ? fps_den : 1;
    c->gop = 60; c->me_range = 16; c->bitrate_bps = 2048000; c->device_id = 0; c->fixed_qp = -1;
    c->qp_min = 10; c->qp_max = 51; c->pipeline_depth = 0; c->profile_events = 0; c->use_graphs = 1; c->keep_prefilter = 0; c->deblock_mode = 0; c->subpel = 1; c->i4x4 = 1; c->transform8x8 = 0; c->intra_in_p = 1; c->vbv_ms = 600; c->cavlc_threads = 0; c->intra_mode = 0; c->scenecut = 1; c->exclusive_device = 0;
}

static unsigned *err_word(const mi355enc_t *h) { return h->d_progress; }
// A kernel that follows another kernel's progress (the band deblocker beside intra_p_kernel) needs the two to be able to run at the
// same time.  One encoder's own streams guarantee that (intra_p_kernel is enqueued first).  Several encoders in one process share the
// process's hardware queues, where a waiting kernel of one can sit in front of the kernel another one waits for; and tools that
// serialise dispatches (rocprofv3 --pmc) run one kernel at a time.  So: only with a single open encoder in the process, and not when
// MI355ENC_SERIAL is set (tools/measure_all.sh sets it for the counter passes).  Every wait is bounded and reported anyway.
static std::atomic<int> g_open_encoders{0};
static bool exclusive_device(const mi355enc_t *h) { static const bool env = getenv("MI355ENC_EXCLUSIVE") != nullptr; return h->cfg.exclusive_device != 0 || env; } // cfg.exclusive_device, or the environment for tools
static bool no_pgate() { static const bool off = getenv("MI355ENC_NO_PGATE") != nullptr; return off; } // A/B switch: the fused P stage in stream order behind the deblocking launch
static bool overlap_allowed() { static const bool serial = getenv("MI355ENC_SERIAL") != nullptr; return !serial && g_open_encoders.load(std::memory_order_relaxed) == 1; }

static void launch_intra_all(mi355enc_t *h, int ci) {
    int n = k_intra_diags(h->mbw, h->mbh);
    for (int d = 0; d < n; d++) k_launch_intra_diag(h->d_ctx2[ci], h->mbw, h->mbh, d, h->stream);
}
static void launch_deblock_all(mi355enc_t *h, int ci) {
    int n = k_deblock_diags(h->mbw, h->mbh);
    for (int d = 0; d < n; d++) k_launch_deblock_diag(h->d_ctx2[ci], h->mbw, h->mbh, d, h->stream);
}
static int build_graph(mi355enc_t *h, int which, int ci, hipGraphExec_t *out) {
    hipGraph_t g;
    HIPCHK(hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
    if (which == 0) launch_intra_all(h, ci); else launch_deblock_all(h, ci);
    HIPCHK(hipStreamEndCapture(h->stream, &g));
    HIPCHK(hipGraphInstantiate(out, g, nullptr, nullptr, 0));
    HIPCHK(hipGraphDestroy(g));
    return 0;
}
static int run_intra(mi355enc_t *h, int ci, const frame_ctx_t *hc, unsigned *band_done = nullptr) {
    k_launch_intra_analyse(hc, h->mbw, h->mbh, 0, h->stream); // open-loop mode analysis + decisions: one flat launch
    if (h->cfg.intra_mode == 0) { // persistent band kernel
        k_launch_intra_band(hc, h->mbh, h->d_ib_gran, err_word(h), band_done, h->stream);
        HIPCHK(hipGetLastError());
        return 0;
    }
    if (h->cfg.use_graphs) {
        if (!h->g_intra[ci]) { int r = build_graph(h, 0, ci, &h->g_intra[ci]); if (r) return r; }
        HIPCHK(hipGraphLaunch(h->g_intra[ci], h->stream));
    } else launch_intra_all(h, ci);
    return 0;
}
// whole picture on the main stream; hc: host copy of the context (by-value kernels), ci: which device copy holds the same (graph kernels)
static int run_deblock(mi355enc_t *h, int ci, const frame_ctx_t *hc, hipStream_t st, const unsigned *ip_progress, const unsigned *iband_done = nullptr, unsigned *band_done = nullptr, bool after_gated_pmb = false) {
    if (h->cfg.deblock_mode == 0) { // the persistent band kernel (its prologue derives the boundary strengths from the records)
        k_launch_deblock_bands(hc, h->mbh, 0, k_deblock_bands16(h->mbh), err_word(h), h->d_db_gran, h->d_db_par, ip_progress, iband_done, k_intra_band_rows(), band_done, h->d_progress + 1, after_gated_pmb ? h->d_row_done : nullptr, h->pmb_rows_total, st);
        h->db_started_total += 2u * (unsigned)k_deblock_bands16(h->mbh);
        HIPCHK(hipGetLastError());
        return 0;
    }
    if (h->cfg.use_graphs) {
        if (!h->g_deblock[ci]) { int r = build_graph(h, 1, ci, &h->g_deblock[ci]); if (r) return r; }
        HIPCHK(hipGraphLaunch(h->g_deblock[ci], h->stream));
    } else launch_deblock_all(h, ci);
    return 0;
}

int mi355enc_open(const mi355enc_cfg_t *cfg, mi355enc_t **out) {
    if (!cfg || !out) return MI355ENC_ERR_ARG;
    *out = nullptr;
    const double t_open = now_ms();
    if (cfg->width < 16 || cfg->height < 16 || cfg->width > 8192 || cfg->height > 8192 || (cfg->width & 1) || (cfg->height & 1) ||
        cfg->fps_num <= 0 || cfg->fps_den <= 0 || cfg->gop < 1 || cfg->me_range < 1 || cfg->me_range > 16 ||
        cfg->pipeline_depth < 0 || cfg->pipeline_depth > NSLOT - 1 || cfg->fixed_qp > 51) {
        fprintf(stderr, "mi355enc: invalid configuration\n");
        return MI355ENC_ERR_ARG;
    }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0 || cfg->device_id < 0 || cfg->device_id >= ndev) {
        fprintf(stderr, "mi355enc: no usable HIP device (count=%d, device-id=%d, %s); this encoder has no CPU path\n", ndev,
                cfg->device_id, e == hipSuccess ? "ok" : hipGetErrorString(e));
        return MI355ENC_ERR_NO_DEVICE;
    }
    HIPCHK(hipSetDevice(cfg->device_id));
    mi355enc_t *h = new (std::nothrow) mi355enc();
    if (!h) return MI355ENC_ERR_NOMEM;
    memset((void *)&h->cfg, 0, sizeof h->cfg);
    h->cfg = *cfg;
    if (h->cfg.qp_min <= 0 && h->cfg.qp_max <= 0) { h->cfg.qp_min = 10; h->cfg.qp_max = 51; }
    if (h->cfg.qp_max > 51) h->cfg.qp_max = 51;
    if (h->cfg.qp_min < 0) h->cfg.qp_min = 0;
    h->mbw = (cfg->width + 15) / 16; h->mbh = (cfg->height + 15) / 16;
    h->W = h->mbw * 16; h->H = h->mbh * 16; h->nmb = h->mbw * h->mbh;
    h->ysz = (size_t)h->W * h->H; h->csz = h->ysz / 2;
    h->head = h->tail = h->pending = 0;
    h->cur = 0; h->have_ref = 0; h->frames_since_idr = 0; h->idr_count = 0; h->last_collected_rec = 0; h->last_slot = nullptr;
    for (int i = 0; i < NSET; i++) { h->g_intra[i] = h->g_deblock[i] = nullptr; h->d_ctx2[i] = nullptr; h->d_surf[i] = nullptr; h->d_idec2[i] = nullptr; h->d_mbi_set[i] = nullptr; h->d_levels_set[i] = nullptr; }
    h->prev_slot = nullptr;
    h->d_ctx = nullptr; h->d_pre_y = h->d_pre_uv = nullptr; memset(h->d_imv, 0, sizeof h->d_imv); h->d_psrc[0] = h->d_psrc[1] = nullptr; h->psrc_cur = 0; h->fstream = nullptr; h->d_ip_progress = nullptr; h->d_ip_strips = nullptr; h->epoch = 0; h->istream = nullptr; h->ev_pmb = nullptr; h->d_db_gran = nullptr; h->d_db_done = nullptr; h->rec_epoch[0] = h->rec_epoch[1] = 0; h->db_started_total = 0; h->d_row_done = nullptr; h->pmb_rows_total = 0; h->d_db_par = nullptr; h->d_ib_gran = nullptr; h->d_iband_done = nullptr; h->ev_dbI[0] = h->ev_dbI[1] = nullptr; h->dbI_busy[0] = h->dbI_busy[1] = 0; h->d_progress = nullptr; h->d_off = nullptr; h->d_isad = nullptr; h->d_dbrec = nullptr; h->d_idec = nullptr;
    h->cstream = nullptr; h->d_mbi = nullptr; h->d_levels = nullptr; 
    memset(&h->st, 0, sizeof h->st);
    h->want_bps.store(cfg->bitrate_bps ? cfg->bitrate_bps : 2048000);
    h->fixed_qp.store(cfg->fixed_qp);
    h->fixed_drop.store(0);
    *out = h; // from here on close() cleans up partial state
    g_open_encoders.fetch_add(1, std::memory_order_relaxed);
    HIPCHK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    for (int i = 0; i < NSET; i++) HIPCHK(hipMalloc((void **)&h->d_ctx2[i], sizeof(frame_ctx_t)));
    h->d_ctx = h->d_ctx2[0];
    { // The hand-over stream gets its own priority level: HIP then backs it with a different hardware queue, so its
      // kernels run beside the persistent deblocking kernel instead of queueing behind it.
        int lo = 0, hi = 0;
        HIPCHK(hipDeviceGetStreamPriorityRange(&lo, &hi));
        HIPCHK(hipStreamCreateWithPriority(&h->cstream, hipStreamNonBlocking, hi));
        HIPCHK(hipStreamCreateWithPriority(&h->fstream, hipStreamNonBlocking, lo));
        HIPCHK(hipStreamCreateWithPriority(&h->istream, hipStreamNonBlocking, 0));
        HIPCHK(hipEventCreateWithFlags(&h->ev_pmb, hipEventDisableTiming));
    }
    for (int i = 0; i < NSET; i++) {
        HIPCHK(hipMalloc((void **)&h->d_mbi_set[i], (size_t)h->nmb * sizeof(mb_info_t)));
        HIPCHK(hipMalloc((void **)&h->d_levels_set[i], (size_t)h->nmb * MB_LEVELS * sizeof(int16_t)));
        HIPCHK(hipMemsetAsync(h->d_mbi_set[i], 0, (size_t)h->nmb * sizeof(mb_info_t), h->stream));
    }
    h->d_mbi = h->d_mbi_set[0]; h->d_levels = h->d_levels_set[0]; h->n_submitted = 0;
    h->sc_sum = 0; h->sc_cnt = 0; h->sc_prev_skip = 0; h->sc_force_at = ~0ull;
    for (int i = 0; i < 2; i++) {
        HIPCHK(hipMalloc((void **)&h->d_rec_y[i], h->ysz + SURF_PAD));
        HIPCHK(hipMalloc((void **)&h->d_rec_uv[i], h->csz + SURF_PAD));
        HIPCHK(hipMemsetAsync(h->d_rec_y[i], 0, h->ysz + SURF_PAD, h->stream));
        HIPCHK(hipMemsetAsync(h->d_rec_uv[i], 0, h->csz + SURF_PAD, h->stream));
    }
    HIPCHK(hipMalloc((void **)&h->d_isad, (size_t)h->nmb * ISAD_PER_MB * sizeof(uint16_t)));
    HIPCHK(hipMalloc((void **)&h->d_dbrec, (size_t)h->nmb * 64));
    HIPCHK(hipMalloc((void **)&h->d_idec, (size_t)h->nmb * IDEC_BYTES + 16));
    h->d_idec2[0] = h->d_idec;
    HIPCHK(hipMalloc((void **)&h->d_progress, 4 * sizeof(unsigned)));
    HIPCHK(hipMemsetAsync(h->d_progress, 0, 4 * sizeof(unsigned), h->stream)); // the error word is sticky: only cleared here
    HIPCHK(hipMalloc((void **)&h->d_db_par, k_deblock_partab_bytes(h->mbw, h->mbh)));
    HIPCHK(hipMalloc((void **)&h->d_ib_gran, (size_t)k_intra_bands(h->mbh) * h->mbw * 8 * sizeof(uint2)));
    HIPCHK(hipMemsetAsync(h->d_ib_gran, 0, (size_t)k_intra_bands(h->mbh) * h->mbw * 8 * sizeof(uint2), h->stream));
    HIPCHK(hipMalloc((void **)&h->d_iband_done, 2 * (size_t)k_intra_bands(h->mbh) * sizeof(unsigned))); // one set per reconstruction buffer: the next picture's wavefront runs beside this one's deblocking
    HIPCHK(hipMemsetAsync(h->d_iband_done, 0, 2 * (size_t)k_intra_bands(h->mbh) * sizeof(unsigned), h->stream));
    for (int i = 0; i < 2; i++) HIPCHK(hipEventCreateWithFlags(&h->ev_dbI[i], hipEventDisableTiming));
    HIPCHK(hipMalloc((void **)&h->d_row_done, (size_t)h->mbh * sizeof(unsigned)));
    HIPCHK(hipMemsetAsync(h->d_row_done, 0, (size_t)h->mbh * sizeof(unsigned), h->stream));
    HIPCHK(hipMalloc((void **)&h->d_db_done, 2 * k_deblock_done_bytes()));
    HIPCHK(hipMemsetAsync(h->d_db_done, 0, 2 * k_deblock_done_bytes(), h->stream)); // epoch 0 is never used
    HIPCHK(hipMalloc((void **)&h->d_db_gran, k_deblock_gran_bytes(h->mbw, h->mbh)));
    HIPCHK(hipMemsetAsync(h->d_db_gran, 0, k_deblock_gran_bytes(h->mbw, h->mbh), h->stream)); // epoch 0 is never used
    HIPCHK(hipMalloc((void **)&h->d_off, (size_t)h->nmb * sizeof(unsigned)));
    for (int k = 0; k < NSET; k++) {
        HIPCHK(hipMalloc((void **)&h->d_surf[k], (size_t)h->nmb * SURF_U16 * sizeof(uint16_t)));
        for (int i = 0; i < 2; i++) HIPCHK(hipMalloc((void **)&h->d_imv[k][i], (size_t)h->nmb * sizeof(imv_t)));
        if (k > 0) HIPCHK(hipMalloc((void **)&h->d_idec2[k], (size_t)h->nmb * IDEC_BYTES + 16));
    }
    for (int k = 0; k < 2; k++) {
        HIPCHK(hipMalloc((void **)&h->d_psrc[k], h->ysz + SURF_PAD));
        HIPCHK(hipMemsetAsync(h->d_psrc[k], 0, h->ysz + SURF_PAD, h->stream));
    }
    HIPCHK(hipMalloc((void **)&h->d_ip_progress, (size_t)h->mbh * sizeof(unsigned)));
    HIPCHK(hipMemsetAsync(h->d_ip_progress, 0, (size_t)h->mbh * sizeof(unsigned), h->stream)); // epoch-tagged: the epoch starts at 1
    HIPCHK(hipMalloc((void **)&h->d_ip_strips, (size_t)h->nmb * 32));
    if (cfg->keep_prefilter) {
        HIPCHK(hipMalloc((void **)&h->d_pre_y, h->ysz));
        HIPCHK(hipMalloc((void **)&h->d_pre_uv, h->csz));
    }
    for (int i = 0; i < NSLOT; i++) {
        slot_t *s = &h->slot[i];
        memset(s, 0, sizeof *s);
        HIPCHK(hipHostMalloc((void **)&s->h_ctx, sizeof(frame_ctx_t), hipHostMallocDefault));
        HIPCHK(hipHostMalloc((void **)&s->h_mbi, (size_t)h->nmb * sizeof(mb_info_t), hipHostMallocDefault));
        HIPCHK(hipHostMalloc((void **)&s->h_levels, (size_t)h->nmb * PACK_BLOCKS_MAX * 32, hipHostMallocDefault));
        HIPCHK(hipHostMalloc((void **)&s->h_hdr, (size_t)(4 + h->mbh) * sizeof(unsigned), hipHostMallocDefault)); // + the summed macroblock cost (two words)
        s->h_hdr[0] = s->h_hdr[1] = 0;
        HIPCHK(hipMalloc((void **)&s->d_src_y, h->ysz + SURF_PAD));
        HIPCHK(hipMalloc((void **)&s->d_src_uv, h->csz + SURF_PAD));
        HIPCHK(hipEventCreateWithFlags(&s->done, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&s->gpu_done, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&s->ev_front, hipEventDisableTiming));
        for (int k = 0; k < 12; k++) HIPCHK(hipEventCreate(&s->ev[k]));
    }
    h->writer = h264_writer_new(h->mbw, h->mbh, h->cfg.transform8x8);
    if (!h->writer) return MI355ENC_ERR_NOMEM;
    if (h->cfg.cavlc_threads <= 0) { // auto, like x264enc's threads=0
        const unsigned hw = std::thread::hardware_concurrency();
        int n = (int)(hw / 4);
        h->cfg.cavlc_threads = h->nmb < 1000 ? 1 : n < 1 ? 1 : n > 8 ? 8 : n;
    }
    if (h->cfg.cavlc_threads > 1 && h264_writer_set_threads(h->writer, h->cfg.cavlc_threads)) return MI355ENC_ERR_NOMEM;
    rc_init(&h->rc, (double)cfg->fps_num / cfg->fps_den, cfg->gop, h->want_bps.load(), h->cfg.qp_min, h->cfg.qp_max);
    if (h->cfg.vbv_ms > 0) rc_set_vbv(&h->rc, h->cfg.vbv_ms);
    HIPCHK(hipStreamSynchronize(h->stream));
    h->ms_open = now_ms() - t_open; h->n_skip_pictures = 0;
    return MI355ENC_OK;
}

void mi355enc_close(mi355enc_t *h) {
    if (!h) return;
    g_open_encoders.fetch_sub(1, std::memory_order_relaxed);
    (void)hipSetDevice(h->cfg.device_id);
    if (h->fstream) (void)hipStreamSynchronize(h->fstream);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->istream) (void)hipStreamSynchronize(h->istream);
    if (h->ev_pmb) (void)hipEventDestroy(h->ev_pmb);
    for (int i = 0; i < NSET; i++) { if (h->g_intra[i]) (void)hipGraphExecDestroy(h->g_intra[i]); if (h->g_deblock[i]) (void)hipGraphExecDestroy(h->g_deblock[i]); }
    for (int i = 0; i < NSLOT; i++) {
        slot_t *s = &h->slot[i];
        if (s->h_ctx) (void)hipHostFree(s->h_ctx);
        if (s->h_mbi) (void)hipHostFree(s->h_mbi);
        if (s->h_levels) (void)hipHostFree(s->h_levels);
        if (s->h_hdr) (void)hipHostFree(s->h_hdr);
        if (s->d_src_y) (void)hipFree(s->d_src_y);
        if (s->d_src_uv) (void)hipFree(s->d_src_uv);
        if (s->d_raw) (void)hipFree(s->d_raw);
        if (s->done) (void)hipEventDestroy(s->done);
        if (s->gpu_done) (void)hipEventDestroy(s->gpu_done);
        if (s->ev_front) (void)hipEventDestroy(s->ev_front);
        for (int k = 0; k < 12; k++) if (s->ev[k]) (void)hipEventDestroy(s->ev[k]);
    }
    for (int i = 0; i < 2; i++) { if (h->d_rec_y[i]) (void)hipFree(h->d_rec_y[i]); if (h->d_rec_uv[i]) (void)hipFree(h->d_rec_uv[i]); }
    if (h->d_pre_y) (void)hipFree(h->d_pre_y);
    if (h->d_pre_uv) (void)hipFree(h->d_pre_uv);
    if (h->d_isad) (void)hipFree(h->d_isad);
    if (h->d_dbrec) (void)hipFree(h->d_dbrec);
    if (h->d_idec) (void)hipFree(h->d_idec);
    if (h->d_progress) (void)hipFree(h->d_progress);
    if (h->d_db_gran) (void)hipFree(h->d_db_gran);
    if (h->d_db_done) (void)hipFree(h->d_db_done);
    if (h->d_row_done) (void)hipFree(h->d_row_done);
    if (h->d_db_par) (void)hipFree(h->d_db_par);
    if (h->d_ib_gran) (void)hipFree(h->d_ib_gran);
    if (h->d_iband_done) (void)hipFree(h->d_iband_done);
    for (int i = 0; i < 2; i++) if (h->ev_dbI[i]) (void)hipEventDestroy(h->ev_dbI[i]);
    if (h->d_off) (void)hipFree(h->d_off);
    for (int k = 0; k < NSET; k++) {
        if (h->d_surf[k]) (void)hipFree(h->d_surf[k]);
        for (int i = 0; i < 2; i++) if (h->d_imv[k][i]) (void)hipFree(h->d_imv[k][i]);
        if (k > 0 && h->d_idec2[k]) (void)hipFree(h->d_idec2[k]);
    }
    for (int k = 0; k < 2; k++) if (h->d_psrc[k]) (void)hipFree(h->d_psrc[k]);
    if (h->d_ip_progress) (void)hipFree(h->d_ip_progress);
    if (h->d_ip_strips) (void)hipFree(h->d_ip_strips);
    for (int i = 0; i < NSET; i++) if (h->d_ctx2[i]) (void)hipFree(h->d_ctx2[i]);
    for (int i = 0; i < NSET; i++) { if (h->d_mbi_set[i]) (void)hipFree(h->d_mbi_set[i]); if (h->d_levels_set[i]) (void)hipFree(h->d_levels_set[i]); }
    if (h->cstream) { (void)hipStreamSynchronize(h->cstream); (void)hipStreamDestroy(h->cstream); }
    if (h->fstream) (void)hipStreamDestroy(h->fstream);
    if (h->istream) (void)hipStreamDestroy(h->istream);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    h264_writer_free(h->writer);
    delete h;
}

int mi355enc_set_bitrate(mi355enc_t *h, uint32_t bps) {
    if (!h) return MI355ENC_ERR_ARG;
    h->want_bps.store(bps < 1000 ? 1000 : bps, std::memory_order_relaxed);
    return MI355ENC_OK;
}
uint32_t mi355enc_get_bitrate(const mi355enc_t *h) { return h ? h->want_bps.load(std::memory_order_relaxed) : 0; }
int mi355enc_set_fixed_qp(mi355enc_t *h, int qp) {
    if (!h || qp > 51) return MI355ENC_ERR_ARG;
    h->fixed_qp.store(qp < 0 ? -1 : qp, std::memory_order_relaxed);
    return MI355ENC_OK;
}
int mi355enc_set_fixed_drop(mi355enc_t *h, int drop) {
    if (!h || drop < 0 || (drop > DROP_MAX && drop != DROP_SKIP)) return MI355ENC_ERR_ARG;
    h->fixed_drop.store(drop, std::memory_order_relaxed);
    return MI355ENC_OK;
}
int mi355enc_pending(const mi355enc_t *h) { return h ? h->pending : 0; }
size_t mi355enc_max_au_bytes(const mi355enc_t *h) { return h ? h264_max_au_bytes(h->mbw, h->mbh) : 0; }
int mi355enc_mb_width(const mi355enc_t *h) { return h ? h->mbw : 0; }
int mi355enc_mb_height(const mi355enc_t *h) { return h ? h->mbh : 0; }

static int sync_compute(mi355enc_t *h) {
    HIPCHK(hipStreamSynchronize(h->fstream));
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipStreamSynchronize(h->istream));
    return 0;
}
// scene-cut recovery: the decision taken with picture k's hand-over lands on picture k + lag, the first one that cannot have been
// submitted yet (depth 0 behaves as depth 1, so that the stream is the same for both)
static int sc_lag(const mi355enc_t *h) { return h->cfg.pipeline_depth >= 2 ? h->cfg.pipeline_depth + 1 : 2; }
static hipStream_t upload_stream(const mi355enc_t *h) { return h->fstream; }

// rate control's ladder below QP 51 (oracle: k_drop_sad): the SAD under which a P macroblock carries no residual / takes the skip vector
static const uint32_t k_drop_sad[DROP_MAX + 1] = {0, 384, 512, 768, 1024, 1536, 2048, 3072, 4096, 6144, 8192, 12288, 0xFFFFFFFFu};
// ... and its counterpart for I pictures (oracle: k_idrop_ac): the sum of level magnitudes up to which a macroblock's luma / chroma residual is not sent
static const int32_t k_idrop_ac[DROP_MAX + 1] = {0, 1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 64, 0x7FFFFFFF};

static void fill_ctx(mi355enc_t *h, frame_ctx_t *c, int qp, int drop, int idr, int set = 0) {
    c->mbi = h->d_mbi; c->levels = h->d_levels; c->isad = h->d_isad; c->dbrec = h->d_dbrec; c->idec = h->d_idec2[set];
    c->stride = h->W; c->mbw = h->mbw; c->mbh = h->mbh;
    c->qp = qp; c->me_range = h->cfg.me_range; c->lambda = k_lambda[qp < 0 ? 0 : qp > 51 ? 51 : qp]; c->i4x4 = h->cfg.i4x4; c->t8 = h->cfg.transform8x8; c->all_intra = idr ? 1 : 0;
    c->surf = h->d_surf[set]; c->imv_a = h->d_imv[set][0]; c->imv_b = h->d_imv[set][1];
    c->me_ref_y = h->d_psrc[h->psrc_cur]; c->psrc_out = h->d_psrc[h->psrc_cur ^ 1];
    if (++h->epoch == 0) h->epoch = 1;
    c->epoch = h->epoch;
    c->drop_sad = (!idr && drop > 0 && drop <= DROP_MAX) ? k_drop_sad[drop] : 0;
    c->iac_drop = (idr && drop > 0 && drop <= DROP_MAX) ? k_idrop_ac[drop] : 0;
    if (c->iac_drop) c->i4x4 = 0; // on the ladder: Intra_16x16 only
    c->intra_p = (h->cfg.intra_in_p && !h->cfg.transform8x8) ? 1 : 0;
}
// P picture, front part (front stream): nothing here depends on the coding of the picture before
static int run_p_front(mi355enc_t *h, const frame_ctx_t *hc, slot_t *s, int prof) {
    hipStream_t st = h->fstream;
    if (prof) HIPCHK(hipEventRecord(s->ev[0], st));
    k_launch_me(hc, h->mbw, 0, h->mbh, st);
    if (prof) HIPCHK(hipEventRecord(s->ev[6], st));
    for (int it = 0; it < ME_ITERS; it++) k_launch_me_select(hc, h->mbw, 0, h->mbh, (it & 1) ? hc->imv_b : hc->imv_a, (it & 1) ? hc->imv_a : hc->imv_b, st);
    if (prof) HIPCHK(hipEventRecord(s->ev[1], st));
    if (!h->cfg.transform8x8 && hc->intra_p) k_launch_intra_analyse(hc, h->mbw, h->mbh, 1, st);
    if (prof) HIPCHK(hipEventRecord(s->ev[7], st));
    HIPCHK(hipGetLastError());
    return 0;
}
// ... and back part (back stream): needs the deblocked picture before it
// gate: the reference picture's band-done words (the fused stage then runs on the intra stream, beside that picture's deblocking)
// rows: the picture's deblocking launch will sit directly behind the previous one and wait on the device for this stage's rows (no event)
static int run_p_back(mi355enc_t *h, const frame_ctx_t *hc, slot_t *s, int prof, int split, const unsigned *gate, unsigned ref_epoch, int rows) {
    hipStream_t st = gate ? h->istream : h->stream;
    if (prof) HIPCHK(hipEventRecord(s->ev[8], st));
    if (h->cfg.transform8x8) { // High profile: the two-kernel form (absolute-vector refinement, 8x8 transform), no skip / intra logic
        k_launch_imv_to_mbi(hc, h->mbw, 0, h->mbh, st);
        if (h->cfg.subpel) k_launch_subpel(hc, h->mbw, 0, h->mbh, st);
        if (prof) HIPCHK(hipEventRecord(s->ev[5], st));
        k_launch_inter(hc, h->mbw, 0, h->mbh, st);
    } else {
        if (gate) k_launch_wait_started(h->d_progress + 1, h->db_started_total, err_word(h), st); // not before the reference's deblocking launch is on the chip
        k_launch_pmb(hc, h->mbw, 0, h->mbh, h->cfg.subpel, gate, ref_epoch, err_word(h), rows ? h->d_row_done : nullptr, st);
        if (prof) HIPCHK(hipEventRecord(s->ev[5], st));
        if (gate) { // the main stream carries nothing but deblocking launches, back to back: this picture's bands wait on the device for the fused
            if (rows) h->pmb_rows_total += (uint32_t)h->mbw; // stage's rows (row counts, no event between the streams), and its movers follow intra_p_kernel
            else { HIPCHK(hipEventRecord(h->ev_pmb, st)); HIPCHK(hipStreamWaitEvent(h->stream, h->ev_pmb, 0)); } // (fewer than three pictures in flight: by event)
            if (hc->intra_p) k_launch_intra_p(hc, h->mbw, h->mbh, h->d_ip_progress, h->d_ip_strips, err_word(h), st);
        } else if (split) { // intra_p_kernel leaves the chain: prep + the band deblocker follow the fused stage directly and overtake it row by row
            HIPCHK(hipEventRecord(h->ev_pmb, st));
            HIPCHK(hipStreamWaitEvent(h->istream, h->ev_pmb, 0));
            k_launch_intra_p(hc, h->mbw, h->mbh, h->d_ip_progress, h->d_ip_strips, err_word(h), h->istream);
        } else if (hc->intra_p) k_launch_intra_p(hc, h->mbw, h->mbh, h->d_ip_progress, h->d_ip_strips, err_word(h), st);
    }
    if (prof) HIPCHK(hipEventRecord(s->ev[11], st));
    HIPCHK(hipGetLastError());
    return 0;
}

// Enqueue every device step of one picture whose source is described by (src_y, src_uv, src_stride).  Anything the caller
// uploaded for this picture was enqueued on the same stream.
static int enqueue_picture(mi355enc_t *h, slot_t *s, const uint8_t *src_y, const uint8_t *src_uv, int src_stride,
                           int64_t pts, int force_idr) {
    const int idr = force_idr || !h->have_ref || h->frames_since_idr >= h->cfg.gop ||
                    (h->n_submitted == h->sc_force_at && h->frames_since_idr >= sc_lag(h)); // scene-cut recovery, see collect(): not when an IDR picture came in between
    if (idr) h->frames_since_idr = 0;
    // rate control: latch the setpoint written by the control thread, pick this picture's QP (and, below QP 51, its drop level)
    rc_set_bitrate(&h->rc, h->want_bps.load(std::memory_order_relaxed));
    int fq = h->fixed_qp.load(std::memory_order_relaxed);
    int qp, drop;
    if (fq >= 0) { qp = fq; drop = h->fixed_drop.load(std::memory_order_relaxed); }
    else rc_pick(&h->rc, idr, &qp, &drop);
    if (idr && drop == DROP_SKIP) drop = 0; // an IDR picture is never skipped; it has a ladder of its own
    const int all_skip = !idr && drop == DROP_SKIP;
    const int nxt = all_skip ? h->cur : (h->cur ^ 1); // an all-skip picture IS its reference: nothing is written
    const int set = (int)(h->n_submitted % NSET), ci = set;
    frame_ctx_t *c = s->h_ctx, *dctx = h->d_ctx2[ci];
    // stage timers: an event record costs ~5 us of queue time, so profile_events = k samples every k-th picture (IDR pictures always)
    // (a sampled picture runs its stages strictly in order; IDR pictures: every other one, or at the P pictures' cadence in an all-intra stream)
    const int prof = !all_skip && h->cfg.profile_events > 0 &&
                     ((idr && h->cfg.gop > 1) ? (h->idr_count & 1) == 0 : h->n_submitted % (uint64_t)h->cfg.profile_events == 0);
    const bool fused = !h->cfg.transform8x8;
    if (all_skip) {
        // one run of P_Skip macroblocks with the zero vector (8.4.1.1 infers it: every neighbour's vector is zero): the host
        // writes the records itself; no source sample is read, no kernel runs, the reference stays where it is
        memset(s->h_mbi, 0, (size_t)h->nmb * sizeof(mb_info_t));
        for (int i = 0; i < h->nmb; i++) { s->h_mbi[i].mb_type = 1; s->h_mbi[i].qp = (uint8_t)qp; }
        s->h_hdr[0] = 0; s->h_hdr[1] = 0;
        for (int r = 0; r < h->mbh; r++) s->h_hdr[2 + r] = 0;
        s->h_hdr[2 + h->mbh] = s->h_hdr[3 + h->mbh] = 0;
        if (h->d_pre_y) {
            HIPCHK(hipMemcpyAsync(h->d_pre_y, h->d_rec_y[nxt], h->ysz, hipMemcpyDeviceToDevice, h->stream));
            HIPCHK(hipMemcpyAsync(h->d_pre_uv, h->d_rec_uv[nxt], h->csz, hipMemcpyDeviceToDevice, h->stream));
        }
    } else {
        c->src_y = src_y; c->src_uv = src_uv; c->src_stride = src_stride;
        c->ref_y = h->d_rec_y[h->cur]; c->ref_uv = h->d_rec_uv[h->cur];
        c->rec_y = h->d_rec_y[nxt]; c->rec_uv = h->d_rec_uv[nxt];
        c->vis_h = h->cfg.height;
        fill_ctx(h, c, qp, drop, idr, set);
        c->mbi = h->d_mbi_set[set]; c->levels = h->d_levels_set[set];
        // Every kernel of the default path takes the context by value; only the kernels replayed from a hipGraph
        // (intra_mode 1, deblock_mode 1) read the device copy, so only those pictures pay for an upload.
        if ((idr && h->cfg.intra_mode != 0) || h->cfg.deblock_mode != 0) HIPCHK(hipMemcpyAsync(dctx, c, sizeof *c, hipMemcpyHostToDevice, h->stream));
        // front stream: the source is in place (upload / conversion were enqueued there); P pictures: search, selection, gated intra
        // analysis; I pictures: only the padded source copy the next picture's search will run against
        if (idr) k_launch_copy_luma(c, h->fstream);
        else { int r = run_p_front(h, c, s, prof); if (r) return r; }
        HIPCHK(hipEventRecord(s->ev_front, h->fstream));
        h->psrc_cur ^= 1;
        // P picture with intra macroblocks: intra_p_kernel (a chain along rows, 10..80 us) and the band deblocker (a chain along
        // x + y) overlap -- intra_p_kernel runs on a stream of its own and the deblocker's movers follow its per-row progress words
        // (GATED, k_deblock.hip); the chain pmb -> prep -> deblocker -> next pmb stays on one stream (a cross-stream event on the
        // chain costs 10-17 us).  Not on pictures whose stage timers are sampled (a gated launch's duration includes its waiting).
        const int split = !idr && fused && c->intra_p && h->cfg.deblock_mode == 0 && !h->d_pre_y && !prof && overlap_allowed();
        // IDR picture: the band deblocker runs on the intra stream BESIDE the intra wavefront, each of its bands waiting for the intra bands
        // of the same rows (flags + acquire); in an all-intra stream the next picture's wavefront then starts while this one is still
        // being deblocked.  What has to wait for such a deblocking: a P picture (it reads the whole reference), and whoever writes
        // the reconstruction buffer it works on (the picture after next).
        // P picture whose reference is still being deblocked: the fused stage leaves the chain too.  It runs on the intra stream, each of
        // its waves waiting for the reference's bands it reads (pmb_kernel<GATED>), so it is all but done when that deblocking ends.
        const size_t nbd = k_deblock_done_bytes() / sizeof(unsigned); // words per reconstruction buffer
        const int pgate = !idr && fused && h->cfg.deblock_mode == 0 && !h->d_pre_y && !prof && overlap_allowed() && h->rec_epoch[h->cur] != 0 && exclusive_device(h) && !no_pgate();
        HIPCHK(hipStreamWaitEvent(pgate ? h->istream : h->stream, s->ev_front, 0));
        // ... and with three pictures in flight (the next picture's front stages are done long before this launch ends) the deblocking launches go back
        // to back, each waiting on the device for its picture's rows; with fewer the host sits on the chain and a launch waiting on the chip only
        // gets in the way (1080p depth 1: 4465 -> 3980 frames/s, 2160p: 2050 -> 1615)
        const int prows = pgate && h->cfg.pipeline_depth >= 2;
        const int isplit = idr && h->cfg.intra_mode == 0 && h->cfg.deblock_mode == 0 && !h->d_pre_y && !prof && overlap_allowed();
        for (int b = 0; b < 2; b++)
            if (h->dbI_busy[b] && (!idr || b == nxt)) { HIPCHK(hipStreamWaitEvent(h->stream, h->ev_dbI[b], 0)); h->dbI_busy[b] = 0; }
        if (idr) {
            if (isplit) { HIPCHK(hipEventRecord(h->ev_pmb, h->stream)); HIPCHK(hipStreamWaitEvent(h->istream, h->ev_pmb, 0)); } // behind everything enqueued so far (a P picture's deblocker, its tables)
            if (prof) HIPCHK(hipEventRecord(s->ev[0], h->stream));
            int r = run_intra(h, ci, c, isplit ? h->d_iband_done + (size_t)nxt * k_intra_bands(h->mbh) : nullptr); if (r) return r;
            if (prof) HIPCHK(hipEventRecord(s->ev[1], h->stream));
        } else {
            int r = run_p_back(h, c, s, prof, split, pgate ? h->d_db_done + (size_t)h->cur * nbd : nullptr, h->rec_epoch[h->cur], prows); if (r) return r;
        }
        if (prof) HIPCHK(hipEventRecord(s->ev[2], h->stream));
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(s->gpu_done, (split || pgate) ? h->istream : h->stream)); // records and levels are final here; they do not depend on deblocking
        if (h->d_pre_y) {
            HIPCHK(hipMemcpyAsync(h->d_pre_y, h->d_rec_y[nxt], h->ysz, hipMemcpyDeviceToDevice, h->stream));
            HIPCHK(hipMemcpyAsync(h->d_pre_uv, h->d_rec_uv[nxt], h->csz, hipMemcpyDeviceToDevice, h->stream));
            if (prof) HIPCHK(hipEventRecord(s->ev[2], h->stream));
        }
        if (isplit) {
            int r = run_deblock(h, ci, c, h->istream, nullptr, h->d_iband_done + (size_t)nxt * k_intra_bands(h->mbh), h->d_db_done + (size_t)nxt * nbd); if (r) return r;
            HIPCHK(hipEventRecord(h->ev_dbI[nxt], h->istream));
            h->dbI_busy[nxt] = 1;
        } else { int r = run_deblock(h, ci, c, h->stream, (split || (pgate && c->intra_p)) ? h->d_ip_progress : nullptr, nullptr, h->d_db_done + (size_t)nxt * nbd, prows != 0); if (r) return r; }
        h->rec_epoch[nxt] = h->cfg.deblock_mode == 0 ? c->epoch : 0;
        if (prof) { HIPCHK(hipEventRecord(s->ev[3], h->stream)); HIPCHK(hipEventRecord(s->ev[4], h->stream)); }
        HIPCHK(hipStreamWaitEvent(h->cstream, s->gpu_done, 0));
        // Hand-over on the second stream, enqueued after the deblocking launches so that it cannot be dispatched ahead of them:
        // the device packs the non-zero blocks straight into the pinned host buffer while the band deblocker runs.
        k_launch_pack(h->d_mbi_set[set], h->d_levels_set[set], h->nmb, h->mbw, h->d_off, s->h_mbi, s->h_levels, s->h_hdr, err_word(h), h->cstream);
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(s->done, h->cstream));
    }
    h->n_submitted++;
    s->is_idr = idr; s->qp = qp; s->drop = drop; s->frame_num = h->frames_since_idr; s->idr_pic_id = h->idr_count & 0xFFFF;
    s->pts = pts; s->rec_index = nxt; s->set = set; s->prof = prof; s->fused = fused && !idr; s->index = h->n_submitted - 1; s->all_skip = all_skip;
    if (idr) h->idr_count++;
    h->frames_since_idr++;
    h->cur = nxt; h->have_ref = 1; h->prev_slot = s;
    h->head = (h->head + 1) % NSLOT; h->pending++;
    return MI355ENC_OK;
}

int mi355enc_submit(mi355enc_t *h, const uint8_t *y, int y_stride, const uint8_t *uv, int uv_stride, int64_t pts, int force_idr) {
    if (!h || !y || !uv || y_stride < h->cfg.width || uv_stride < h->cfg.width) return MI355ENC_ERR_ARG;
    if (h->pending > h->cfg.pipeline_depth) return MI355ENC_ERR_STATE;
    HIPCHK(hipSetDevice(h->cfg.device_id));
    slot_t *s = &h->slot[h->head];
    const int w = h->cfg.width, ht = h->cfg.height;
    hipStream_t up = upload_stream(h);
    HIPCHK(hipMemcpy2DAsync(s->d_src_y, h->W, y, y_stride, w, ht, hipMemcpyHostToDevice, up));
    HIPCHK(hipMemcpy2DAsync(s->d_src_uv, h->W, uv, uv_stride, w, ht / 2, hipMemcpyHostToDevice, up));
    if (w != h->W) k_launch_pad(s->d_src_y, s->d_src_uv, h->W, w, ht, h->W, h->H, up);
    return enqueue_picture(h, s, s->d_src_y, s->d_src_uv, h->W, pts, force_idr);
}

// Upload the planes of a non-NV12 picture tightly into the slot's raw staging buffer and convert into its NV12 staging surfaces.
static int upload_and_convert(mi355enc_t *h, slot_t *s, int fmt, const uint8_t *const planes[3], const int strides[3], hipStream_t up) {
    const int w = h->cfg.width, ht = h->cfg.height;
    if (fmt < MI355ENC_FMT_I420 || fmt > MI355ENC_FMT_UYVY || !planes || !strides || !planes[0]) return MI355ENC_ERR_ARG;
    if (!s->d_raw) HIPCHK(hipMalloc((void **)&s->d_raw, (size_t)(2 * h->W + 32) * h->H + 64));
    if (fmt == MI355ENC_FMT_I420) {
        if (!planes[1] || !planes[2] || strides[0] < w || strides[1] < w / 2 || strides[2] < w / 2) return MI355ENC_ERR_ARG;
        const int r0 = (w + 15) & ~15, r1 = (w / 2 + 15) & ~15;
        uint8_t *dy = s->d_raw, *du = dy + (size_t)r0 * ht, *dv = du + (size_t)r1 * (ht / 2);
        HIPCHK(hipMemcpy2DAsync(dy, r0, planes[0], strides[0], w, ht, hipMemcpyHostToDevice, up));
        HIPCHK(hipMemcpy2DAsync(du, r1, planes[1], strides[1], w / 2, ht / 2, hipMemcpyHostToDevice, up));
        HIPCHK(hipMemcpy2DAsync(dv, r1, planes[2], strides[2], w / 2, ht / 2, hipMemcpyHostToDevice, up));
        if (k_launch_csc(fmt, dy, du, dv, r0, r1, r1, s->d_src_y, s->d_src_uv, w, ht, h->W, h->H, up)) return MI355ENC_ERR_ARG;
    } else {
        if (strides[0] < 2 * w) return MI355ENC_ERR_ARG;
        const int r0 = (2 * w + 15) & ~15;
        HIPCHK(hipMemcpy2DAsync(s->d_raw, r0, planes[0], strides[0], 2 * w, ht, hipMemcpyHostToDevice, up));
        if (k_launch_csc(fmt, s->d_raw, nullptr, nullptr, r0, 0, 0, s->d_src_y, s->d_src_uv, w, ht, h->W, h->H, up)) return MI355ENC_ERR_ARG;
    }
    HIPCHK(hipGetLastError());
    return MI355ENC_OK;
}

int mi355enc_submit_fmt(mi355enc_t *h, int fmt, const uint8_t *const planes[3], const int strides[3], int64_t pts, int force_idr) {
    if (!h || !planes || !strides) return MI355ENC_ERR_ARG;
    if (fmt == MI355ENC_FMT_NV12) return mi355enc_submit(h, planes[0], strides[0], planes[1], strides[1], pts, force_idr);
    if (h->pending > h->cfg.pipeline_depth) return MI355ENC_ERR_STATE;
    HIPCHK(hipSetDevice(h->cfg.device_id));
    slot_t *s = &h->slot[h->head];
    int r = upload_and_convert(h, s, fmt, planes, strides, upload_stream(h));
    if (r) return r;
    return enqueue_picture(h, s, s->d_src_y, s->d_src_uv, h->W, pts, force_idr);
}

int mi355enc_stage_csc(mi355enc_t *h, int fmt, const uint8_t *const planes[3], const int strides[3], uint8_t *out_y, uint8_t *out_uv) {
    if (!h || !out_y || !out_uv || h->pending) return MI355ENC_ERR_ARG;
    HIPCHK(hipSetDevice(h->cfg.device_id));
    slot_t *s = &h->slot[0];
    int r = upload_and_convert(h, s, fmt, planes, strides, h->stream);
    if (r) return r;
    HIPCHK(hipMemcpyAsync(out_y, s->d_src_y, h->ysz, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(out_uv, s->d_src_uv, h->csz, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return MI355ENC_OK;
}

int mi355enc_submit_device(mi355enc_t *h, const void *d_y, int y_stride, const void *d_uv, int uv_stride, int64_t pts, int force_idr) {
    if (!h || !d_y || !d_uv || y_stride < h->cfg.width || uv_stride < h->cfg.width) return MI355ENC_ERR_ARG;
    if (h->pending > h->cfg.pipeline_depth) return MI355ENC_ERR_STATE;
    HIPCHK(hipSetDevice(h->cfg.device_id));
    slot_t *s = &h->slot[h->head];
    const int w = h->cfg.width, ht = h->cfg.height;
    hipStream_t up = upload_stream(h);
    const bool direct = w == h->W && y_stride == uv_stride && (y_stride & 15) == 0 && (((uintptr_t)d_y | (uintptr_t)d_uv) & 15) == 0;
    if (direct) return enqueue_picture(h, s, (const uint8_t *)d_y, (const uint8_t *)d_uv, y_stride, pts, force_idr);
    HIPCHK(hipMemcpy2DAsync(s->d_src_y, h->W, d_y, y_stride, w, ht, hipMemcpyDeviceToDevice, up));
    HIPCHK(hipMemcpy2DAsync(s->d_src_uv, h->W, d_uv, uv_stride, w, ht / 2, hipMemcpyDeviceToDevice, up));
    if (w != h->W) k_launch_pad(s->d_src_y, s->d_src_uv, h->W, w, ht, h->W, h->H, up);
    return enqueue_picture(h, s, s->d_src_y, s->d_src_uv, h->W, pts, force_idr);
}

int mi355enc_collect(mi355enc_t *h, uint8_t *out, size_t out_cap, size_t *out_len, int *is_keyframe, int64_t *pts, int *qp) {
    if (!h || !out || !out_len) return MI355ENC_ERR_ARG;
    if (h->pending <= 0) return MI355ENC_ERR_STATE;
    HIPCHK(hipSetDevice(h->cfg.device_id));
    slot_t *s = &h->slot[h->tail];
    double t0 = now_ms();
    if (!s->all_skip) HIPCHK(hipEventSynchronize(s->done));
    double t1 = now_ms();
    h->st.ms_wait += t1 - t0;
    if (s->h_hdr[1]) { // sticky: set by a band of an earlier picture's deblocking launch that gave up waiting
        fprintf(stderr, "mi355enc: a device-side wait timed out (error word %u: 3 pmb_kernel gate, 4 wait_started_kernel, 11 deblocker / intra bands, 12 deblocker / strips, 13 deblocker / intra_p_kernel, 14 intra band / strips, 15 intra_p_kernel / row above, 16 progress counter)\n", s->h_hdr[1]);
        return MI355ENC_ERR_HIP;
    }
    size_t n = 0;
    if (s->is_idr) {
        n = h264_write_headers(out, out_cap, h->cfg.width, h->cfg.height, h->cfg.fps_num, h->cfg.fps_den, h->cfg.transform8x8);
        if (!n) return MI355ENC_ERR_OVERFLOW;
    }
    size_t m = h264_write_slice_packed_rows(h->writer, out + n, out_cap - n, s->is_idr, s->frame_num, s->idr_pic_id, s->qp, s->h_mbi, s->h_levels, s->h_hdr + 2);
    if (!m) return MI355ENC_ERR_OVERFLOW;
    h->st.ms_entropy += now_ms() - t1;
    *out_len = n + m;
    if (is_keyframe) *is_keyframe = s->is_idr;
    if (pts) *pts = s->pts;
    if (qp) *qp = s->qp;
    rc_update(&h->rc, s->is_idr, s->qp, s->drop, n + m);
    // Scene-cut recovery (cfg.scenecut; the oracle's orc_enc_frame applies the same rule): the summed cost of the picture's
    // macroblocks came with the hand-over.  The decision lands on picture index + 2, the first one not submitted yet whatever
    // the pipeline depth, and is skipped there if picture index + 1 turned out to be an IDR: the stream does not depend on
    // the order of submit() and collect() calls.
    if (s->is_idr) { h->sc_sum = 0; h->sc_cnt = 0; }
    else if (!s->all_skip && !h->sc_prev_skip) { // (a picture that follows P_Skip-run pictures is searched against an older source: its cost says nothing about a cut)
        const uint64_t cost = (uint64_t)s->h_hdr[2 + h->mbh] | ((uint64_t)s->h_hdr[3 + h->mbh] << 32);
        const bool pending = h->sc_force_at != ~0ull && h->sc_force_at > s->index; // a decision not yet carried out stands
        if (h->cfg.scenecut && !pending && h->sc_cnt >= 2 && cost > 3 * (h->sc_sum / (uint64_t)h->sc_cnt)) h->sc_force_at = s->index + (uint64_t)sc_lag(h);
        h->sc_sum += cost; h->sc_cnt++;
    }
    h->sc_prev_skip = s->all_skip;
    if (s->prof) {
        float a = 0, b = 0, c = 0, tot = 0, sp = 0;
        {
            HIPCHK(hipEventSynchronize(s->ev[4])); // the access unit is ready before deblocking ends; the stage timers are not
            float sel = 0, an = 0, ip = 0;
            if (s->is_idr) { (void)hipEventElapsedTime(&a, s->ev[0], s->ev[1]); (void)hipEventElapsedTime(&tot, s->ev[0], s->ev[4]); }
            else { // front-stream stages and back-stream stages are timed on their own streams; the picture's total is their sum
                float fe = 0, be = 0;
                (void)hipEventElapsedTime(&a, s->ev[0], s->ev[6]);
                (void)hipEventElapsedTime(&sel, s->ev[6], s->ev[1]);
                (void)hipEventElapsedTime(&an, s->ev[1], s->ev[7]);
                (void)hipEventElapsedTime(&fe, s->ev[0], s->ev[7]);
                (void)hipEventElapsedTime(&be, s->ev[8], s->ev[4]);
                tot = fe + be;
                if (s->fused) { (void)hipEventElapsedTime(&b, s->ev[8], s->ev[11]); b += an; (void)hipEventElapsedTime(&ip, s->ev[5], s->ev[11]); } // analysis + fused stage + intra macroblocks, booked as inter
                else { (void)hipEventElapsedTime(&sp, s->ev[8], s->ev[5]); (void)hipEventElapsedTime(&b, s->ev[5], s->ev[11]); }
                h->st.ms_select += sel; h->st.ms_analyse_p += an; h->st.ms_intra_p += ip;
            }
            (void)hipEventElapsedTime(&c, s->ev[2], s->ev[3]);
        }
        if (s->is_idr) { h->st.ms_intra += a; h->st.n_intra++; }
        else { h->st.ms_me += a; h->st.n_me++; h->st.ms_inter += b; h->st.n_inter++; h->st.ms_subpel += sp; }
        h->st.ms_deblock += c; h->st.n_deblock++;
        if (s->is_idr) { h->st.ms_deblock_idr += c; h->st.n_deblock_idr++; }
        h->st.ms_total_gpu += tot; h->st.n_total_gpu++;
    }
    h->st.frames++; h->st.idr_frames += s->is_idr; h->st.bytes += n + m;
    h->st.last_qp = (uint32_t)s->qp; h->st.last_drop = (uint32_t)s->drop; h->n_skip_pictures += s->all_skip; h->st.last_bytes = (uint32_t)(n + m); h->st.target_bps = h->want_bps.load();
    h->last_slot = s; h->last_collected_rec = s->rec_index;
    h->tail = (h->tail + 1) % NSLOT; h->pending--;
    return MI355ENC_OK;
}

int mi355enc_encode(mi355enc_t *h, const uint8_t *y, int y_stride, const uint8_t *uv, int uv_stride, int64_t pts, int force_idr,
                    uint8_t *out, size_t out_cap, size_t *out_len, int *is_keyframe) {
    if (!h) return MI355ENC_ERR_ARG;
    if (h->pending) return MI355ENC_ERR_STATE;
    int r = mi355enc_submit(h, y, y_stride, uv, uv_stride, pts, force_idr);
    if (r) return r;
    return mi355enc_collect(h, out, out_cap, out_len, is_keyframe, nullptr, nullptr);
}

int mi355enc_get_stats(mi355enc_t *h, mi355enc_stats_t *st) {
    if (!h || !st) return MI355ENC_ERR_ARG;
    *st = h->st;
    st->target_bps = h->want_bps.load();
    st->cavlc_threads = (uint32_t)h->cfg.cavlc_threads;
    st->ms_open = h->ms_open; st->skip_pictures = h->n_skip_pictures;
    return MI355ENC_OK;
}
void mi355enc_reset_stats(mi355enc_t *h) { if (h) { memset(&h->st, 0, sizeof h->st); h->n_skip_pictures = 0; } }

int mi355enc_fetch(mi355enc_t *h, int what, void *dst, size_t n) {
    if (!h || !dst) return MI355ENC_ERR_ARG;
    HIPCHK(hipSetDevice(h->cfg.device_id));
    const void *src = nullptr; size_t need = 0; bool host = false;
    switch (what) {
    case MI355ENC_FETCH_RECON_Y: src = h->d_rec_y[h->last_collected_rec]; need = h->ysz; break;
    case MI355ENC_FETCH_RECON_UV: src = h->d_rec_uv[h->last_collected_rec]; need = h->csz; break;
    case MI355ENC_FETCH_PREFILTER_Y: src = h->d_pre_y; need = h->ysz; break;
    case MI355ENC_FETCH_PREFILTER_UV: src = h->d_pre_uv; need = h->csz; break;
    case MI355ENC_FETCH_MBINFO: src = h->last_slot ? h->last_slot->h_mbi : nullptr; need = (size_t)h->nmb * sizeof(mb_info_t); host = true; break;
    case MI355ENC_FETCH_LEVELS: src = h->last_slot ? h->d_levels_set[h->last_slot->set] : nullptr; need = (size_t)h->nmb * MB_LEVELS * 2; break; // dense, from HBM
    case 100: src = h->d_dbrec; need = (size_t)h->nmb * 64; break; /* development: deblocking records (cycle counters in -DD3_PROF builds) */
    case 101: src = h->d_isad; need = 1024; break;                 /* development: cycle counters of -DIB_PROF builds */
    default: return MI355ENC_ERR_ARG;
    }
    if (!src) return MI355ENC_ERR_STATE;
    if (n < need) return MI355ENC_ERR_OVERFLOW;
    if (host) { memcpy(dst, src, need); return MI355ENC_OK; }
    { int r = sync_compute(h); if (r) return r; }
    HIPCHK(hipMemcpy(dst, src, need, hipMemcpyDeviceToHost));
    return MI355ENC_OK;
}

// ---------------------------------------------------------------- single-stage entry points
static int stage_ctx(mi355enc_t *h, int qp, bool src_is_staging, int drop = 0, int idr = 0) {
    if (h->pending) return MI355ENC_ERR_STATE;
    slot_t *s = &h->slot[0];
    frame_ctx_t *c = s->h_ctx;
    c->src_y = src_is_staging ? s->d_src_y : nullptr; c->src_uv = src_is_staging ? s->d_src_uv : nullptr; c->src_stride = h->W;
    c->ref_y = h->d_rec_y[0]; c->ref_uv = h->d_rec_uv[0]; c->rec_y = h->d_rec_y[1]; c->rec_uv = h->d_rec_uv[1];
    HIPCHK(hipStreamSynchronize(h->cstream));
    { int r = sync_compute(h); if (r) return r; }
    c->vis_h = h->H;
    fill_ctx(h, c, qp, drop, idr);
    c->all_intra = 0; // the single-stage deblocking entry point takes records of either picture type
    HIPCHK(hipMemcpyAsync(h->d_ctx, c, sizeof *c, hipMemcpyHostToDevice, h->stream));
    return 0;
}
static int upload_luma_pair(mi355enc_t *h, const uint8_t *cur_y, const uint8_t *ref_y) {
    HIPCHK(hipMemcpyAsync(h->slot[0].d_src_y, cur_y, h->ysz, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_rec_y[0], ref_y, h->ysz, hipMemcpyHostToDevice, h->stream));
    return 0;
}
int mi355enc_stage_me(mi355enc_t *h, const uint8_t *cur_y, const uint8_t *ref_y, int qp, uint16_t *surf_out, void *imv_out) {
    if (!h || !cur_y || !ref_y || !imv_out || qp < 0 || qp > 51) return MI355ENC_ERR_ARG;
    HIPCHK(hipSetDevice(h->cfg.device_id));
    HIPCHK(hipMemcpyAsync(h->slot[0].d_src_y, cur_y, h->ysz, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_psrc[h->psrc_cur], ref_y, h->ysz, hipMemcpyHostToDevice, h->stream)); // what the search runs against (in the encoder: the previous source)
    int r = stage_ctx(h, qp, true); if (r) return r;
    k_launch_me(h->slot[0].h_ctx, h->mbw, 0, h->mbh, h->stream);
    HIPCHK(hipMemcpyAsync(imv_out, h->d_imv[0][0], (size_t)h->nmb * sizeof(imv_t), hipMemcpyDeviceToHost, h->stream));
    if (surf_out) HIPCHK(hipMemcpyAsync(surf_out, h->d_surf[0], (size_t)h->nmb * SURF_U16 * sizeof(uint16_t), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return MI355ENC_OK;
}
int mi355enc_stage_me_select(mi355enc_t *h, const uint16_t *surf, const void *imv_in, int qp, void *imv_out) {
    if (!h || !surf || !imv_in || !imv_out || qp < 0 || qp > 51) return MI355ENC_ERR_ARG;
    HIPCHK(hipSetDevice(h->cfg.device_id));
    int r = stage_ctx(h, qp, true); if (r) return r;
    HIPCHK(hipMemcpyAsync(h->d_surf[0], surf, (size_t)h->nmb * SURF_U16 * sizeof(uint16_t), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_imv[0][0], imv_in, (size_t)h->nmb * sizeof(imv_t), hipMemcpyHostToDevice, h->stream));
    k_launch_me_select(h->slot[0].h_ctx, h->mbw, 0, h->mbh, h->d_imv[0][0], h->d_imv[0][1], h->stream);
    HIPCHK(hipMemcpyAsync(imv_out, h->d_imv[0][1], (size_t)h->nmb * sizeof(imv_t), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return MI355ENC_OK;
}
int mi355enc_stage_subpel(mi355enc_t *h, const uint8_t *cur_y, const uint8_t *ref_y, int qp, void *mbinfo_inout) {
    if (!h || !cur_y || !ref_y || !mbinfo_inout || qp < 0 || qp > 51) return MI355ENC_ERR_ARG;
    HIPCHK(hipSetDevice(h->cfg.device_id));
    int r = upload_luma_pair(h, cur_y, ref_y); if (r) return r;
    HIPCHK(hipMemcpyAsync(h->d_mbi, mbinfo_inout, (size_t)h->nmb * sizeof(mb_info_t), hipMemcpyHostToDevice, h->stream));
    r = stage_ctx(h, qp, true); if (r) return r;
    k_launch_subpel(h->slot[0].h_ctx, h->mbw, 0, h->mbh, h->stream);
    HIPCHK(hipMemcpyAsync(mbinfo_inout, h->d_mbi, (size_t)h->nmb * sizeof(mb_info_t), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return MI355ENC_OK;
}
static int upload_planes4(mi355enc_t *h, const uint8_t *src_y, const uint8_t *src_uv, const uint8_t *ref_y, const uint8_t *ref_uv) {
    HIPCHK(hipMemcpyAsync(h->slot[0].d_src_y, src_y, h->ysz, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->slot[0].d_src_uv, src_uv, h->csz, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_rec_y[0], ref_y, h->ysz, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_rec_uv[0], ref_uv, h->csz, hipMemcpyHostToDevice, h->stream));
    return 0;
}
static int download_picture(mi355enc_t *h, void *mbinfo, uint8_t *rec_y, uint8_t *rec_uv, int16_t *levels) {
    HIPCHK(hipMemcpyAsync(mbinfo, h->d_mbi, (size_t)h->nmb * sizeof(mb_info_t), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(rec_y, h->d_rec_y[1], h->ysz, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(rec_uv, h->d_rec_uv[1], h->csz, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(levels, h->d_levels, (size_t)h->nmb * MB_LEVELS * 2, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}
int mi355enc_stage_inter(mi355enc_t *h, const uint8_t *src_y, const uint8_t *src_uv, const uint8_t *ref_y, const uint8_t *ref_uv,
                         int qp, void *mbinfo_inout, uint8_t *rec_y, uint8_t *rec_uv, int16_t *levels) {
    if (!h || !src_y || !src_uv || !ref_y || !ref_uv || !mbinfo_inout || !rec_y || !rec_uv || !levels || qp < 0 || qp > 51) return MI355ENC_ERR_ARG;
    HIPCHK(hipSetDevice(h->cfg.device_id));
    int r = upload_planes4(h, src_y, src_uv, ref_y, ref_uv); if (r) return r;
    HIPCHK(hipMemcpyAsync(h->d_mbi, mbinfo_inout, (size_t)h->nmb * sizeof(mb_info_t), hipMemcpyHostToDevice, h->stream));
    r = stage_ctx(h, qp, true); if (r) return r;
    k_launch_inter(h->slot[0].h_ctx, h->mbw, 0, h->mbh, h->stream);
    return download_picture(h, mbinfo_inout, rec_y, rec_uv, levels) ? MI355ENC_ERR_HIP : MI355ENC_OK;
}
int mi355enc_stage_pmb(mi355enc_t *h, const uint8_t *src_y, const uint8_t *src_uv, const uint8_t *ref_y, const uint8_t *ref_uv,
                       int qp, int drop, int refine, const void *imv, const uint16_t *surf, const void *idec, int run_intra_p,
                       void *mbinfo_out, uint8_t *rec_y, uint8_t *rec_uv, int16_t *levels) {
    if (!h || !src_y || !src_uv || !ref_y || !ref_uv || !imv || !surf || !mbinfo_out || !rec_y || !rec_uv || !levels || qp < 0 || qp > 51 || drop < 0 || drop > DROP_MAX)
        return MI355ENC_ERR_ARG;
    HIPCHK(hipSetDevice(h->cfg.device_id));
    int r = upload_planes4(h, src_y, src_uv, ref_y, ref_uv); if (r) return r;
    r = stage_ctx(h, qp, true, drop); if (r) return r;
    frame_ctx_t *c = h->slot[0].h_ctx;
    c->intra_p = idec ? 1 : 0;
    HIPCHK(hipMemcpyAsync((void *)k_final_imv(c), imv, (size_t)h->nmb * sizeof(imv_t), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_surf[0], surf, (size_t)h->nmb * SURF_U16 * sizeof(uint16_t), hipMemcpyHostToDevice, h->stream));
    if (idec) HIPCHK(hipMemcpyAsync(h->d_idec, idec, (size_t)h->nmb * IDEC_BYTES, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemsetAsync(h->d_rec_y[1], 0, h->ysz, h->stream)); // macroblocks decided intra stay untouched unless run_intra_p
    HIPCHK(hipMemsetAsync(h->d_rec_uv[1], 0, h->csz, h->stream));
    HIPCHK(hipMemsetAsync(h->d_levels, 0, (size_t)h->nmb * MB_LEVELS * 2, h->stream));
    k_launch_pmb(c, h->mbw, 0, h->mbh, refine ? 1 : 0, nullptr, 0, err_word(h), nullptr, h->stream);
    if (idec && run_intra_p) k_launch_intra_p(c, h->mbw, h->mbh, h->d_ip_progress, h->d_ip_strips, err_word(h), h->stream);
    HIPCHK(hipGetLastError());
    return download_picture(h, mbinfo_out, rec_y, rec_uv, levels) ? MI355ENC_ERR_HIP : MI355ENC_OK;
}
int mi355enc_stage_intra(mi355enc_t *h, const uint8_t *src_y, const uint8_t *src_uv, int qp, int drop, void *mbinfo_out, uint8_t *rec_y,
                         uint8_t *rec_uv, int16_t *levels) {
    if (!h || !src_y || !src_uv || !mbinfo_out || !rec_y || !rec_uv || !levels || qp < 0 || qp > 51 || drop < 0 || drop > DROP_MAX) return MI355ENC_ERR_ARG;
    HIPCHK(hipSetDevice(h->cfg.device_id));
    HIPCHK(hipMemcpyAsync(h->slot[0].d_src_y, src_y, h->ysz, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->slot[0].d_src_uv, src_uv, h->csz, hipMemcpyHostToDevice, h->stream));
    int r = stage_ctx(h, qp, true, drop, 1); if (r) return r;
    r = run_intra(h, 0, h->slot[0].h_ctx); if (r) return r;
    return download_picture(h, mbinfo_out, rec_y, rec_uv, levels) ? MI355ENC_ERR_HIP : MI355ENC_OK;
}
int mi355enc_stage_intra_analyse(mi355enc_t *h, const uint8_t *src_y, const uint8_t *src_uv, int qp, uint16_t *isad_out, void *idec_out) {
    if (!h || !src_y || !src_uv || !isad_out || qp < 0 || qp > 51) return MI355ENC_ERR_ARG;
    HIPCHK(hipSetDevice(h->cfg.device_id));
    HIPCHK(hipMemcpyAsync(h->slot[0].d_src_y, src_y, h->ysz, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->slot[0].d_src_uv, src_uv, h->csz, hipMemcpyHostToDevice, h->stream));
    int r = stage_ctx(h, qp, true); if (r) return r;
    k_launch_intra_analyse(h->slot[0].h_ctx, h->mbw, h->mbh, 0, h->stream);
    HIPCHK(hipMemcpyAsync(isad_out, h->d_isad, (size_t)h->nmb * ISAD_PER_MB * sizeof(uint16_t), hipMemcpyDeviceToHost, h->stream));
    if (idec_out) HIPCHK(hipMemcpyAsync(idec_out, h->d_idec, (size_t)h->nmb * IDEC_BYTES, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return MI355ENC_OK;
}
int mi355enc_stage_deblock(mi355enc_t *h, uint8_t *rec_y, uint8_t *rec_uv, const void *mbinfo) {
    if (!h || !rec_y || !rec_uv || !mbinfo) return MI355ENC_ERR_ARG;
    HIPCHK(hipSetDevice(h->cfg.device_id));
    HIPCHK(hipMemcpyAsync(h->d_rec_y[1], rec_y, h->ysz, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_rec_uv[1], rec_uv, h->csz, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_mbi, mbinfo, (size_t)h->nmb * sizeof(mb_info_t), hipMemcpyHostToDevice, h->stream));
    int r = stage_ctx(h, 26, false); if (r) return r;
    r = run_deblock(h, 0, h->slot[0].h_ctx, h->stream, nullptr); if (r) return r;
    HIPCHK(hipMemcpyAsync(rec_y, h->d_rec_y[1], h->ysz, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(rec_uv, h->d_rec_uv[1], h->csz, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return MI355ENC_OK;
}
int mi355enc_time_stage(mi355enc_t *h, int stage, int iters, double *avg_ms) {
    if (!h || !avg_ms || iters < 1 || stage < 0 || stage > 10) return MI355ENC_ERR_ARG;
    if (h->pending) return MI355ENC_ERR_STATE;
    HIPCHK(hipSetDevice(h->cfg.device_id));
    slot_t *s = &h->slot[0];
    // a valid context in both places: the last picture's host copy (slot 0) re-uploaded, or a fresh stage context
    if (!h->have_ref) { int r = stage_ctx(h, 26, true); if (r) return r; }
    else HIPCHK(hipMemcpyAsync(h->d_ctx, s->h_ctx, sizeof(frame_ctx_t), hipMemcpyHostToDevice, h->stream));
    if (stage >= 5 && !s->d_raw) { // input conversion (5 I420, 6 YUY2, 7 UYVY): any bytes will do as a source
        HIPCHK(hipMalloc((void **)&s->d_raw, (size_t)(2 * h->W + 32) * h->H + 64));
        HIPCHK(hipMemsetAsync(s->d_raw, 0x55, (size_t)(2 * h->W + 32) * h->H + 64, h->stream));
    }
    for (int warm = 0; warm < 2; warm++) {
        if (warm) HIPCHK(hipEventRecord(s->ev[0], h->stream));
        for (int i = 0; i < (warm ? iters : 1); i++) {
            if (stage == 0) k_launch_me(h->slot[0].h_ctx, h->mbw, 0, h->mbh, h->stream);
            else if (stage == 1) k_launch_inter(h->slot[0].h_ctx, h->mbw, 0, h->mbh, h->stream);
            else if (stage == 2) { if (++h->epoch == 0) h->epoch = 1; h->slot[0].h_ctx->epoch = h->epoch; int r = run_intra(h, 0, h->slot[0].h_ctx); if (r) return r; } // a fresh stamp per launch: the lines between bands are epoch-tagged
            else if (stage == 4) k_launch_subpel(h->slot[0].h_ctx, h->mbw, 0, h->mbh, h->stream);
            else if (stage == 8) k_launch_me_select(h->slot[0].h_ctx, h->mbw, 0, h->mbh, h->d_imv[0][0], h->d_imv[0][1], h->stream);
            else if (stage == 9) k_launch_pmb(h->slot[0].h_ctx, h->mbw, 0, h->mbh, 1, nullptr, 0, err_word(h), nullptr, h->stream);
            else if (stage == 10) k_launch_intra_p(h->slot[0].h_ctx, h->mbw, h->mbh, h->d_ip_progress, h->d_ip_strips, err_word(h), h->stream);
            else if (stage >= 5) {
                const int w = h->cfg.width, ht = h->cfg.height, r0 = stage == 5 ? (w + 15) & ~15 : (2 * w + 15) & ~15, r1 = (w / 2 + 15) & ~15;
                const uint8_t *p0 = s->d_raw, *p1 = p0 + (size_t)r0 * ht, *p2 = p1 + (size_t)r1 * (ht / 2);
                k_launch_csc(stage - 4, p0, p1, p2, r0, r1, r1, s->d_src_y, s->d_src_uv, w, ht, h->W, h->H, h->stream);
            }
            else { if (++h->epoch == 0) h->epoch = 1; h->slot[0].h_ctx->epoch = h->epoch; int r = run_deblock(h, 0, h->slot[0].h_ctx, h->stream, nullptr); if (r) return r; } // a fresh stamp per launch: the strips between bands are epoch-tagged
        }
        if (warm) HIPCHK(hipEventRecord(s->ev[1], h->stream));
    }
    HIPCHK(hipEventSynchronize(s->ev[1]));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, s->ev[0], s->ev[1]));
    *avg_ms = (double)ms / iters;
    return MI355ENC_OK;
}

// ---------------------------------------------------------------- host-only stages
int mi355enc_host_write_headers(int width, int height, int fps_num, int fps_den, int t8, uint8_t *out, size_t cap, size_t *out_len) {
    if (!out || !out_len || width < 16 || height < 16 || fps_num <= 0 || fps_den <= 0) return MI355ENC_ERR_ARG;
    size_t n = h264_write_headers(out, cap, width, height, fps_num, fps_den, t8);
    if (!n) return MI355ENC_ERR_OVERFLOW;
    *out_len = n;
    return MI355ENC_OK;
}
int mi355enc_host_write_slice(int mbw, int mbh, int is_idr, int frame_num, int idr_pic_id, int qp, int t8, const void *mbinfo,
                              const int16_t *levels, uint8_t *out, size_t cap, size_t *out_len) {
    if (!out || !out_len || !mbinfo || !levels || mbw < 1 || mbh < 1 || qp < 0 || qp > 51) return MI355ENC_ERR_ARG;
    h264_writer_t *w = h264_writer_new(mbw, mbh, t8);
    if (!w) return MI355ENC_ERR_NOMEM;
    size_t n = h264_write_slice(w, out, cap, is_idr, frame_num, idr_pic_id, qp, (const mb_info_t *)mbinfo, levels);
    h264_writer_free(w);
    if (!n) return MI355ENC_ERR_OVERFLOW;
    *out_len = n;
    return MI355ENC_OK;
}
// Host statement of what levels_pack_kernel + levels_scan_kernel hand over (same block order), then the slice writer on
// `threads` host threads: lets the row-parallel coder be checked against the dense single-thread one without a device.
int mi355enc_host_write_slice_packed(int mbw, int mbh, int is_idr, int frame_num, int idr_pic_id, int qp, int t8, int threads, const void *mbinfo,
                                     const int16_t *levels, uint8_t *out, size_t cap, size_t *out_len) {
    if (!out || !out_len || !mbinfo || !levels || mbw < 1 || mbh < 1 || qp < 0 || qp > 51 || threads < 1) return MI355ENC_ERR_ARG;
    const mb_info_t *mbi = (const mb_info_t *)mbinfo;
    const size_t nmb = (size_t)mbw * mbh;
    int16_t *packed = (int16_t *)malloc(nmb * PACK_BLOCKS_MAX * 32 + 32);
    uint32_t *row_off = (uint32_t *)malloc((size_t)mbh * sizeof(uint32_t));
    h264_writer_t *w = h264_writer_new(mbw, mbh, t8);
    int rc = MI355ENC_ERR_NOMEM;
    if (packed && row_off && w && h264_writer_set_threads(w, threads) == 0) {
        h264_pack_levels(mbw, mbh, mbi, levels, packed, row_off);
        size_t n = h264_write_slice_packed_rows(w, out, cap, is_idr, frame_num, idr_pic_id, qp, mbi, packed, row_off);
        rc = n ? MI355ENC_OK : MI355ENC_ERR_OVERFLOW;
        *out_len = n;
    }
    h264_writer_free(w); free(packed); free(row_off);
    return rc;
}
int mi355enc_host_cavlc_block(const int16_t *coef, int maxnum, int nC, uint8_t *out, size_t cap) {
    if (!coef || !out) return MI355ENC_ERR_ARG;
    return h264_cavlc_block_bits(coef, maxnum, nC, out, cap);
}
static_assert(sizeof(rc_state_t) <= MI355ENC_RC_BYTES, "MI355ENC_RC_BYTES too small");
void mi355enc_rc_init(void *rc, double fps, int gop, uint32_t bps, int qp_min, int qp_max) { rc_init((rc_state_t *)rc, fps, gop, bps, qp_min, qp_max); }
void mi355enc_rc_set_bitrate(void *rc, uint32_t bps) { rc_set_bitrate((rc_state_t *)rc, bps); }
void mi355enc_rc_pick(void *rc, int is_idr, int *qp, int *drop) { int q = 0, d = 0; rc_pick((rc_state_t *)rc, is_idr, &q, &d); if (qp) *qp = q; if (drop) *drop = d; }
void mi355enc_rc_update(void *rc, int is_idr, int qp, int drop, size_t bytes) { rc_update((rc_state_t *)rc, is_idr, qp, drop, bytes); }

} // extern "C"
