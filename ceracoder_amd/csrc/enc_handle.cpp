// enc_handle.cpp -- the handle of the C-ABI shim (include/mi355enc.h): open / close, the setters the control thread calls,
// statistics and the inspection of the last collected picture.  The picture pipeline is in enc_schedule.cpp.
#include "enc_internal.hpp"

// A kernel that follows another kernel's progress (the band deblocker beside intra_p_kernel) needs the two to be able to run at the
// same time.  One encoder's own streams guarantee that (intra_p_kernel is enqueued first).  Several encoders in one process share the
// process's hardware queues, where a waiting kernel of one can sit in front of the kernel another one waits for; and tools that
// serialise dispatches (rocprofv3 --pmc) run one kernel at a time.  So: only with a single open encoder in the process, and not when
// MI355ENC_SERIAL is set (tools/measure_all.sh sets it for the counter passes).  Every wait is bounded and reported anyway.
std::atomic<int> g_open_encoders{0};
bool exclusive_device(const mi355enc_t *h) { static const bool env = getenv("MI355ENC_EXCLUSIVE") != nullptr; return h->cfg.exclusive_device != 0 || env; } // cfg.exclusive_device, or the environment for tools
// The fused P stage of picture n + 1 beside the deblocking launch of picture n, its workgroups waiting on the device for the bands they read (pmb_kernel<GATED>): worth 8 % at 1080p,
// where the P period is a dependency chain with most of the chip idle -- and a LOSS where the flat kernels' chip time is the period (r04: 2160p 2 573 -> 2 757 frames/s without it;
// tests/devtools/ab_lib.py, alternating processes): tens of thousands of waiting waves take issue slots from the kernels the picture is actually waiting for.  So: up to
// MI355ENC_PGATE_MAX macroblocks per picture (default 16 384: 1080p and 1440p yes, 2160p no); MI355ENC_NO_PGATE=1 switches it off everywhere (A/B).  Latched at open().
bool pgate_on(int nmb) {
    if (getenv("MI355ENC_NO_PGATE")) return false;
    const char *m = getenv("MI355ENC_PGATE_MAX");
    return nmb <= (m ? atoi(m) : 16384);
}
// The intra macroblock rows of a P picture as workgroups of its deblocking launch instead of intra_p_kernel behind pmb_kernel: measured (tests/devtools/ab_env.py, alternating
// runs in one process) +6 % at 720p, +-0 at 1080p, -3.6 % at 2160p -- so up to 720p's 3600 macroblocks.  MI355ENC_FIP / MI355ENC_NO_FIP force it; latched once per encoder at
// open() (mi355enc::fip_rows): a running encoder's schedule does not depend on a mutable environment, and getenv() is not called beside a host application's setenv().
bool fip_on(int nmb) { return getenv("MI355ENC_NO_FIP") ? false : getenv("MI355ENC_FIP") ? true : nmb <= 3600; }
bool overlap_allowed(const mi355enc_t *h) {
    static const bool serial = getenv("MI355ENC_SERIAL") != nullptr;
    return !serial && h->safe_level == 0 && !h->cfg.single_stream && g_open_encoders.load(std::memory_order_relaxed) == 1;
}

int sync_compute(mi355enc_t *h) {
    if (h->ustream) HIPCHK(hipStreamSynchronize(h->ustream));
    HIPCHK(hipStreamSynchronize(h->fstream));
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipStreamSynchronize(h->cstream)); // (also the second home of the deblocking launches)
    HIPCHK(hipStreamSynchronize(h->istream));
    return 0;
}

// ---- pinned input memory (mi355enc_host_alloc): a process-wide list of ranges, so that submit() can tell a picture it may DMA from directly
#include <mutex>
#include <vector>
static std::mutex g_pin_mu;
static std::vector<std::pair<uintptr_t, size_t>> g_pins;
bool host_range_pinned(const void *p, size_t bytes) {
    const uintptr_t a = (uintptr_t)p;
    std::lock_guard<std::mutex> g(g_pin_mu);
    for (const auto &r : g_pins)
        if (a >= r.first && a + bytes <= r.first + r.second) return true;
    return false;
}

extern "C" {

void *mi355enc_host_alloc(size_t bytes) {
    void *p = nullptr;
    if (!bytes || hipHostMalloc(&p, bytes, hipHostMallocPortable) != hipSuccess || !p) { (void)hipGetLastError(); return nullptr; }
    std::lock_guard<std::mutex> g(g_pin_mu);
    g_pins.emplace_back((uintptr_t)p, bytes);
    return p;
}
void mi355enc_host_free(void *p) {
    if (!p) return;
    {
        std::lock_guard<std::mutex> g(g_pin_mu);
        for (size_t i = 0; i < g_pins.size(); i++)
            if (g_pins[i].first == (uintptr_t)p) { g_pins.erase(g_pins.begin() + (long)i); break; }
    }
    (void)hipHostFree(p);
}

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
    c->qp_min = 10; c->qp_max = 51; c->pipeline_depth = 0; c->profile_events = 0; c->use_graphs = 1; c->keep_prefilter = 0; c->deblock_mode = 0; c->subpel = 1; c->i4x4 = 1; c->transform8x8 = 0; c->intra_in_p = 1; c->vbv_ms = 600; c->cavlc_threads = 0; c->intra_mode = 0; c->scenecut = 1; c->exclusive_device = 0; c->aq_mode = 0; c->single_stream = 0; c->intra_slices = 0; c->partitions = 0; c->profile_overlap = 0; c->i8x8 = 0; c->slices = 0; c->slice_deblock = 1;
}

int mi355enc_open(const mi355enc_cfg_t *cfg, mi355enc_t **out) {
    if (!cfg || !out) return MI355ENC_ERR_ARG;
    *out = nullptr;
    const double t_open = now_ms();
    if (cfg->width < 16 || cfg->height < 16 || cfg->width > 8192 || cfg->height > 8192 || (cfg->width & 1) || (cfg->height & 1) ||
        cfg->fps_num <= 0 || cfg->fps_den <= 0 || cfg->gop < 1 || cfg->me_range < 1 || cfg->me_range > 16 ||
        cfg->pipeline_depth < 0 || cfg->pipeline_depth > NSLOT - 1 || cfg->fixed_qp > 51 || cfg->slices < 0 || cfg->intra_slices < 0) {
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
    {   // slices (oracle: orc_auto_intra_slices, orc_slice_rows_for).  I pictures: about 17 rows each by default, at most 8; P pictures: cfg.slices (0: the same default; 1: one slice).
        // With slice-local deblocking every slice is a whole number of the deblocker's bands.
        const bool local = h->cfg.slice_deblock != 0;
        auto rows_for = [&](int n) { if (n <= 1) return 0; int rows = (h->mbh + n - 1) / n; if (local) rows = (rows + MI355_BAND_ROWS - 1) / MI355_BAND_ROWS * MI355_BAND_ROWS; return rows >= h->mbh ? 0 : rows; };
        static_assert(MI355_BAND_ROWS == 4, "the oracle rounds slice heights to multiples of four rows (orc_slice_rows_for)");
        const int nauto = h->mbh / 17 < 1 ? 1 : h->mbh / 17 > 8 ? 8 : h->mbh / 17;
        const int ns = h->cfg.intra_slices > 0 ? h->cfg.intra_slices : nauto, np = h->cfg.slices > 0 ? h->cfg.slices : nauto;
        h->islice_rows = rows_for(ns > h->mbh ? h->mbh : ns);
        h->pslice_rows = rows_for(np > h->mbh ? h->mbh : np);
        h->slice_dbf = local ? 2 : 0;
        h->stage_slice_rows = 0; h->stage_slice_dbf = 0;
    }
    h->W = h->mbw * 16; h->H = h->mbh * 16; h->nmb = h->mbw * h->mbh;
    h->ysz = (size_t)h->W * h->H; h->csz = h->ysz / 2;
    h->head = h->tail = h->pending = 0;
    h->cur = 0; h->have_ref = 0; h->frames_since_idr = 0; h->idr_count = 0; h->last_collected_rec = 0; h->last_slot = nullptr;
    for (int i = 0; i < NSET; i++) { h->g_intra[i] = h->g_deblock[i] = nullptr; h->d_ctx2[i] = nullptr; h->d_surf[i] = nullptr; h->d_idec2[i] = nullptr; h->d_mbi_set[i] = nullptr; h->d_levels_set[i] = nullptr; h->d_qp_off[i] = nullptr; }
    h->prev_slot = nullptr;
    h->d_ctx = nullptr; h->d_pre_y = h->d_pre_uv = nullptr; memset(h->d_imv, 0, sizeof h->d_imv); h->d_psrc[0] = h->d_psrc[1] = nullptr; h->psrc_cur = 0; h->fstream = nullptr; h->ustream = nullptr; h->d_ip_progress = nullptr; h->d_ip_strips = nullptr; h->epoch = 0; h->istream = nullptr; h->ev_pmb = nullptr; h->d_db_gran = nullptr; h->d_db_done = nullptr; h->rec_epoch[0] = h->rec_epoch[1] = 0; h->db_started_total = 0; h->ip_done_total = 0; h->d_row_done = nullptr; h->pmb_rows_total = 0; h->d_db_par = nullptr; h->d_db_part = nullptr; h->d_ib_gran = nullptr; h->d_iband_done = nullptr; h->ev_dbI[0] = h->ev_dbI[1] = nullptr; h->dbI_busy[0] = h->dbI_busy[1] = 0; h->d_progress = nullptr; h->d_off = nullptr; h->d_isad = nullptr; h->d_dbrec = nullptr; h->d_idec = nullptr;
    h->cstream = nullptr; h->d_mbi = nullptr; h->d_levels = nullptr;
    memset(&h->st, 0, sizeof h->st);
    h->want_bps.store(cfg->bitrate_bps ? cfg->bitrate_bps : 2048000);
    h->fixed_qp.store(cfg->fixed_qp);
    h->fixed_drop.store(0);
    *out = h; // from here on close() cleans up partial state
    g_open_encoders.fetch_add(1, std::memory_order_relaxed);
    HIPCHK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    h->fip_rows = fip_on(h->nmb);
    h->pgate = pgate_on(h->nmb);
    for (int i = 0; i < NSET; i++) HIPCHK(hipMalloc((void **)&h->d_ctx2[i], sizeof(frame_ctx_t)));
    h->d_ctx = h->d_ctx2[0];
    if (h->cfg.single_stream) { // one hardware queue per encoder: every stage in order on the main stream
        h->cstream = h->fstream = h->istream = h->stream;
        HIPCHK(hipEventCreateWithFlags(&h->ev_pmb, hipEventDisableTiming));
    } else { // The hand-over stream gets its own priority level: HIP then backs it with a different hardware queue, so its
      // kernels run beside the persistent deblocking kernel instead of queueing behind it.
        int lo = 0, hi = 0;
        HIPCHK(hipDeviceGetStreamPriorityRange(&lo, &hi));
        HIPCHK(hipStreamCreateWithPriority(&h->cstream, hipStreamNonBlocking, hi));
        // MI355ENC_RESERVE_CUS=n[,mode] (development): the front and the intra stream -- the search, the selection, the fused stage, intra_p_kernel: thousands of
        // four-wave workgroups that fill every slot that frees up -- stay off n compute units, so that a deblocking launch (34 workgroups of twelve waves at 1080p, each
        // of which needs an all but empty compute unit) finds room when the launch before it ends instead of when the fused stage has drained.
        // mode 0: the mask's first n bits, 1: every (256 / n)-th bit.
        const char *rs = getenv("MI355ENC_RESERVE_CUS");
        const int nres = rs ? atoi(rs) : 0, rmode = (rs && strchr(rs, ',')) ? atoi(strchr(rs, ',') + 1) : 0;
        if (nres > 0 && nres < 200) {
            hipDeviceProp_t pr;
            HIPCHK(hipGetDeviceProperties(&pr, h->cfg.device_id));
            const int ncu = pr.multiProcessorCount;
            uint32_t mask[16];
            memset(mask, 0, sizeof mask);
            for (int i = 0; i < ncu && i < 512; i++) mask[i >> 5] |= 1u << (i & 31);
            for (int k = 0; k < nres; k++) { const int i = rmode == 1 ? (int)((long long)k * ncu / nres) : k; mask[i >> 5] &= ~(1u << (i & 31)); }
            HIPCHK(hipExtStreamCreateWithCUMask(&h->fstream, (uint32_t)((ncu + 31) / 32), mask));
            HIPCHK(hipExtStreamCreateWithCUMask(&h->istream, (uint32_t)((ncu + 31) / 32), mask));
        } else {
        HIPCHK(hipStreamCreateWithPriority(&h->fstream, hipStreamNonBlocking, getenv("MI355ENC_FPRIO") ? atoi(getenv("MI355ENC_FPRIO")) : lo));
        HIPCHK(hipStreamCreateWithPriority(&h->istream, hipStreamNonBlocking, 0));
        }
        if (h->cfg.pipeline_depth >= 1 && !getenv("MI355ENC_NO_UPSTREAM")) HIPCHK(hipStreamCreateWithFlags(&h->ustream, hipStreamNonBlocking));
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
    if (!getenv("MI355ENC_NO_SPLIT")) { // (A/B switch, read once per encoder)
        HIPCHK(hipMalloc((void **)&h->d_db_part, 6 * (size_t)k_deblock_bands16(h->mbh) * sizeof(unsigned))); // two counters and two {cut, epoch} granules per band
        HIPCHK(hipMemsetAsync(h->d_db_part, 0, 6 * (size_t)k_deblock_bands16(h->mbh) * sizeof(unsigned), h->stream));
    }
    HIPCHK(hipMalloc((void **)&h->d_ib_gran, (size_t)h->mbh * h->mbw * 8 * sizeof(uint2)));
    HIPCHK(hipMemsetAsync(h->d_ib_gran, 0, (size_t)h->mbh * h->mbw * 8 * sizeof(uint2), h->stream));
    HIPCHK(hipMalloc((void **)&h->d_iband_done, 2 * (size_t)h->mbh * sizeof(unsigned))); // one word per macroblock row (intra_mode 2: per band), one set per reconstruction buffer: the next picture's wavefront runs beside this one's deblocking
    HIPCHK(hipMemsetAsync(h->d_iband_done, 0, 2 * (size_t)h->mbh * sizeof(unsigned), h->stream));
    for (int i = 0; i < 2; i++) HIPCHK(hipEventCreateWithFlags(&h->ev_dbI[i], hipEventDisableTiming));
    HIPCHK(hipMalloc((void **)&h->d_row_done, (size_t)h->mbh * MI355_PROG_STRIDE * sizeof(unsigned)));
    HIPCHK(hipMemsetAsync(h->d_row_done, 0, (size_t)h->mbh * MI355_PROG_STRIDE * sizeof(unsigned), h->stream));
    HIPCHK(hipMalloc((void **)&h->d_db_done, 2 * k_deblock_done_bytes()));
    HIPCHK(hipMemsetAsync(h->d_db_done, 0, 2 * k_deblock_done_bytes(), h->stream)); // epoch 0 is never used
    HIPCHK(hipMalloc((void **)&h->d_db_gran, k_deblock_gran_bytes(h->mbw, h->mbh)));
    HIPCHK(hipMemsetAsync(h->d_db_gran, 0, k_deblock_gran_bytes(h->mbw, h->mbh), h->stream)); // epoch 0 is never used
    HIPCHK(hipMalloc((void **)&h->d_off, (size_t)h->nmb * sizeof(unsigned)));
    for (int k = 0; k < NSET; k++) {
        HIPCHK(hipMalloc((void **)&h->d_surf[k], (size_t)h->nmb * SURF_U16 * sizeof(uint16_t)));
        for (int i = 0; i < 3; i++) HIPCHK(hipMalloc((void **)&h->d_imv[k][i], (size_t)h->nmb * sizeof(imv_t)));
        if (k > 0) HIPCHK(hipMalloc((void **)&h->d_idec2[k], (size_t)h->nmb * IDEC_BYTES + 16));
        if (h->cfg.aq_mode) HIPCHK(hipMalloc((void **)&h->d_qp_off[k], (size_t)h->nmb + 16));
    }
    for (int k = 0; k < 2; k++) {
        HIPCHK(hipMalloc((void **)&h->d_psrc[k], h->ysz + SURF_PAD));
        HIPCHK(hipMemsetAsync(h->d_psrc[k], 0, h->ysz + SURF_PAD, h->stream));
    }
    HIPCHK(hipMalloc((void **)&h->d_ip_progress, (size_t)h->mbh * MI355_PROG_STRIDE * sizeof(unsigned)));
    HIPCHK(hipMemsetAsync(h->d_ip_progress, 0, (size_t)h->mbh * MI355_PROG_STRIDE * sizeof(unsigned), h->stream)); // epoch-tagged: the epoch starts at 1
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
        HIPCHK(hipEventCreateWithFlags(&s->ev_up, hipEventDisableTiming));
        for (int k = 0; k < 12; k++) HIPCHK(hipEventCreate(&s->ev[k]));
    }
    h->writer = h264_writer_new(h->mbw, h->mbh, h->cfg.transform8x8);
    if (!h->writer) return MI355ENC_ERR_NOMEM;
    h264_writer_set_slice_rows(h->writer, h->islice_rows);
    h264_writer_set_p_slices(h->writer, h->pslice_rows, h->slice_dbf);
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
    h->safe_level = 0; h->n_recoveries = 0; h->last_error_word = 0;
    // Entropy coding off the caller's thread: with pictures in flight the caller's loop was submit (~60 us of launches) + slice coding (~100 us
    // on the pool) per picture -- as long as the device's period.  A worker takes the slice coding: it waits for a picture's hand-over, codes the
    // access unit into the slot's buffer and collect() only copies it out.  Not at pipeline_depth 0 (the latency mode: nothing to overlap).
    h->wk_qh = h->wk_qt = 0; h->wk_stop = false; h->wk_on = false; h->au_cap = 0;
    if (h->cfg.pipeline_depth >= 1 && !getenv("MI355ENC_SYNC_ENTROPY")) {
        h->au_cap = h264_max_au_bytes(h->mbw, h->mbh);
        for (int i = 0; i < NSLOT; i++) { h->slot[i].au = (uint8_t *)malloc(h->au_cap); if (!h->slot[i].au) return MI355ENC_ERR_NOMEM; }
        h->wk_on = true;
        h->wk = std::thread(entropy_worker, h);
    }
    h->stg_n = h->stg_next = h->stg_done = h->stg_err = 0; h->stg_gen = 0; h->stg_stop = false; h->stg_on = false;
    if (h->cfg.pipeline_depth >= 1 && !getenv("MI355ENC_NO_STAGE_THREADS")) {
        h->stg_on = true;
        for (auto &t : h->stg_th) t = std::thread(stage_helper, h);
    }
    return MI355ENC_OK;
}

void mi355enc_close(mi355enc_t *h) {
    if (!h) return;
    if (h->wk_on) {
        { std::lock_guard<std::mutex> g(h->wk_mu); h->wk_stop = true; }
        h->wk_cv.notify_all();
        h->wk.join();
        h->wk_on = false;
    }
    if (h->stg_on) {
        { std::lock_guard<std::mutex> g(h->stg_mu); h->stg_stop = true; }
        h->stg_cv.notify_all();
        for (auto &t : h->stg_th) t.join();
        h->stg_on = false;
    }
    g_open_encoders.fetch_sub(1, std::memory_order_relaxed);
    (void)hipSetDevice(h->cfg.device_id);
    if (h->ustream) { (void)hipStreamSynchronize(h->ustream); (void)hipStreamDestroy(h->ustream); }
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
        if (s->h_src) (void)hipHostFree(s->h_src);
        free(s->au);
        if (s->d_src_y) (void)hipFree(s->d_src_y);
        if (s->d_src_uv) (void)hipFree(s->d_src_uv);
        if (s->d_raw) (void)hipFree(s->d_raw);
        if (s->done) (void)hipEventDestroy(s->done);
        if (s->gpu_done) (void)hipEventDestroy(s->gpu_done);
        if (s->ev_front) (void)hipEventDestroy(s->ev_front);
        if (s->ev_up) (void)hipEventDestroy(s->ev_up);
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
    if (h->d_db_part) (void)hipFree(h->d_db_part);
    if (h->d_ib_gran) (void)hipFree(h->d_ib_gran);
    if (h->d_iband_done) (void)hipFree(h->d_iband_done);
    for (int i = 0; i < 2; i++) if (h->ev_dbI[i]) (void)hipEventDestroy(h->ev_dbI[i]);
    if (h->d_off) (void)hipFree(h->d_off);
    for (int k = 0; k < NSET; k++) {
        if (h->d_surf[k]) (void)hipFree(h->d_surf[k]);
        for (int i = 0; i < 3; i++) if (h->d_imv[k][i]) (void)hipFree(h->d_imv[k][i]);
        if (k > 0 && h->d_idec2[k]) (void)hipFree(h->d_idec2[k]);
        if (h->d_qp_off[k]) (void)hipFree(h->d_qp_off[k]);
    }
    for (int k = 0; k < 2; k++) if (h->d_psrc[k]) (void)hipFree(h->d_psrc[k]);
    if (h->d_ip_progress) (void)hipFree(h->d_ip_progress);
    if (h->d_ip_strips) (void)hipFree(h->d_ip_strips);
    for (int i = 0; i < NSET; i++) if (h->d_ctx2[i]) (void)hipFree(h->d_ctx2[i]);
    for (int i = 0; i < NSET; i++) { if (h->d_mbi_set[i]) (void)hipFree(h->d_mbi_set[i]); if (h->d_levels_set[i]) (void)hipFree(h->d_levels_set[i]); }
    if (h->cstream && h->cstream != h->stream) { (void)hipStreamSynchronize(h->cstream); (void)hipStreamDestroy(h->cstream); }
    if (h->fstream && h->fstream != h->stream) (void)hipStreamDestroy(h->fstream);
    if (h->istream && h->istream != h->stream) (void)hipStreamDestroy(h->istream);
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

int mi355enc_get_stats(mi355enc_t *h, mi355enc_stats_t *st) {
    if (!h || !st) return MI355ENC_ERR_ARG;
    *st = h->st;
    st->target_bps = h->want_bps.load();
    st->cavlc_threads = (uint32_t)h->cfg.cavlc_threads;
    st->ms_open = h->ms_open; st->skip_pictures = h->n_skip_pictures;
    st->recoveries = h->n_recoveries; st->last_error_word = h->last_error_word; st->safe_level = (uint32_t)h->safe_level;
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
    case 102: src = h->d_db_part; need = h->d_db_part ? 6 * (size_t)k_deblock_bands16(h->mbh) * sizeof(unsigned) : 0; break; /* development: per band and plane the parts' counter, then {cut column, epoch} of the last launch */
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

} // extern "C"
