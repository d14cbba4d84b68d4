// enc_schedule.cpp -- the picture pipeline of the C-ABI shim.
//
// Per picture:
//   front stream: [H2D source | conversion] -> P: me_kernel (SAD surfaces + first selection) -> me_select_kernel x ME_ITERS
//                 -> intra analysis of the badly predicted macroblocks;  IDR: the source copy the next search runs against
//   back stream:  IDR: intra analysis (flat) -> intra wavefront (persistent bands) | P: pmb_kernel -> intra_p_kernel
//                 -> deblocking (persistent band kernel)
//   hand-over stream, from the moment records and levels are final: scan + pack into pinned host memory -> event
//   host, when the event has fired: CAVLC slice coding (h264_host.c).
// With pipeline_depth = 1 the host codes picture n while the device works on n+1; with 2 a third picture is in flight.
#include "enc_internal.hpp"

static void worker_push(mi355enc_t *h, int slot_index);

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
int run_intra(mi355enc_t *h, int ci, const frame_ctx_t *hc, unsigned *band_done) {
    k_launch_intra_analyse(hc, h->mbw, h->mbh, 0, h->stream); // open-loop mode analysis + decisions: one flat launch
    if (h->cfg.intra_mode == 0 || h->cfg.intra_mode == 2) { // one persistent launch: dataflow per macroblock row (0) / the lock-step band kernel (2, kept for A/B)
        if (h->cfg.intra_mode == 0) k_launch_intra_rows(hc, h->mbh, h->d_ib_gran, err_word(h), band_done, h->stream);
        else k_launch_intra_band(hc, h->mbh, h->d_ib_gran, err_word(h), band_done, h->stream);
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
int run_deblock(mi355enc_t *h, int ci, const frame_ctx_t *hc, hipStream_t st, const unsigned *ip_progress, const unsigned *iband_done, unsigned *band_done, bool after_gated_pmb, unsigned row_need, bool fused_ip) {
    if (h->cfg.deblock_mode == 0) { // the persistent band kernel (its prologue derives the boundary strengths from the records)
        const int wgs = k_launch_deblock_bands(hc, h->mbh, 0, k_deblock_bands16(h->mbh), err_word(h), h->d_db_gran, h->d_db_par, ip_progress, iband_done, h->cfg.intra_mode == 2 ? k_intra_band_rows() : 1, band_done, h->d_progress + 1, after_gated_pmb ? h->d_row_done : nullptr, row_need ? row_need : h->pmb_rows_total,
                                               fused_ip ? h->d_ip_strips : nullptr, fused_ip ? h->d_progress + 2 : nullptr, h->d_progress + 3, h->qpc_total, h->d_db_part, st);
        if (hc->qp_off) h->qpc_total += (uint32_t)h->mbh; // the QP_Y chain rides in the launch and counts the rows it has resolved
        if (fused_ip) h->ip_done_total += (uint32_t)h->mbh;
        h->db_started_total += (unsigned)wgs; // (two per band, or four where every band is walked as two parts)
        HIPCHK(hipGetLastError());
        return 0;
    }
    if (h->cfg.use_graphs) {
        if (!h->g_deblock[ci]) { int r = build_graph(h, 1, ci, &h->g_deblock[ci]); if (r) return r; }
        HIPCHK(hipGraphLaunch(h->g_deblock[ci], h->stream));
    } else launch_deblock_all(h, ci);
    return 0;
}
// scene-cut recovery: the decision taken with picture k's hand-over lands on picture k + lag, the first one that cannot have been
// submitted yet (depth 0 behaves as depth 1, so that the stream is the same for both)
static int sc_lag(const mi355enc_t *h) { return h->cfg.pipeline_depth >= 2 ? h->cfg.pipeline_depth + 1 : 2; }
static hipStream_t upload_stream(const mi355enc_t *h) { return h->ustream ? h->ustream : h->fstream; }
// the front stream's kernels read the source: behind the upload, if that went to a stream of its own
static int upload_done(mi355enc_t *h, slot_t *s) {
    if (h->ustream) { HIPCHK(hipEventRecord(s->ev_up, h->ustream)); HIPCHK(hipStreamWaitEvent(h->fstream, s->ev_up, 0)); }
    return 0;
}

// rate control's ladder below QP 51 (oracle: k_drop_sad): the SAD under which a P macroblock carries no residual / takes the skip vector
static const uint32_t k_drop_sad[DROP_MAX + 1] = {0, 384, 512, 768, 1024, 1536, 2048, 3072, 4096, 6144, 8192, 12288, 0xFFFFFFFFu};
// ... and its counterpart for I pictures (oracle: k_idrop_ac): the sum of level magnitudes up to which a macroblock's luma / chroma residual is not sent
#define I8_QP_MAX 37 /* Intra_8x8 is tried up to this picture quantiser (oracle: ORC_I8_QP_MAX) */
static const int32_t k_idrop_ac[DROP_MAX + 1] = {0, 1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 64, 0x7FFFFFFF};

void fill_ctx(mi355enc_t *h, frame_ctx_t *c, int qp, int drop, int idr, int set) {
    c->mbi = h->d_mbi; c->levels = h->d_levels; c->isad = h->d_isad; c->dbrec = h->d_dbrec; c->idec = h->d_idec2[set];
    c->stride = h->W; c->mbw = h->mbw; c->mbh = h->mbh;
    c->slice_rows = idr ? h->islice_rows : h->pslice_rows; c->slice_dbf = h->slice_dbf;
    c->partitions = (!idr && h->cfg.partitions && !h->cfg.transform8x8 && h->cfg.deblock_mode == 0) ? 1 : 0;
    c->qp = qp; c->me_range = h->cfg.me_range; c->lambda = k_lambda[qp < 0 ? 0 : qp > 51 ? 51 : qp]; c->i4x4 = h->cfg.i4x4; c->t8 = h->cfg.transform8x8; c->all_intra = idr ? 1 : 0;
    c->surf = h->d_surf[set]; c->imv_a = h->d_imv[set][0]; c->imv_b = h->d_imv[set][1]; c->imv_c = h->d_imv[set][2];
    c->me_ref_y = h->d_psrc[h->psrc_cur]; c->psrc_out = h->d_psrc[h->psrc_cur ^ 1];
    if (++h->epoch == 0) h->epoch = 1;
    c->epoch = h->epoch;
    c->drop_sad = (!idr && drop > 0 && drop <= DROP_MAX) ? k_drop_sad[drop] : 0;
    c->iac_drop = (idr && drop > 0 && drop <= DROP_MAX) ? k_idrop_ac[drop] : 0;
    if (c->iac_drop) c->i4x4 = 0; // on the ladder: Intra_16x16 only
    c->i8 = (idr && h->cfg.transform8x8 && h->cfg.i8x8 && h->cfg.intra_mode == 0 && !c->iac_drop && qp <= I8_QP_MAX) ? 1 : 0;
    c->qp_off = h->cfg.aq_mode ? h->d_qp_off[set] : nullptr;
    c->intra_p = h->cfg.intra_in_p ? (h->cfg.i4x4 && h->cfg.intra_in_p > 1 ? 2 : 1) : 0; // 2: Intra_4x4 as well
}
// P picture, front part (front stream): nothing here depends on the coding of the picture before
static int run_p_front(mi355enc_t *h, const frame_ctx_t *hc, slot_t *s, int prof) {
    hipStream_t st = h->fstream;
    if (prof) HIPCHK(hipEventRecord(s->ev[0], st));
    k_launch_me(hc, h->mbw, 0, h->mbh, st);
    if (prof) HIPCHK(hipEventRecord(s->ev[6], st));
    k_launch_me_select_all(hc, h->mbw, 0, h->mbh, st);
    if (prof) HIPCHK(hipEventRecord(s->ev[1], st));
    if (hc->intra_p) k_launch_intra_analyse(hc, h->mbw, h->mbh, 1, st);
    if (prof) HIPCHK(hipEventRecord(s->ev[7], st));
    HIPCHK(hipGetLastError());
    return 0;
}
// ... and back part (back stream): needs the deblocked picture before it
// gate: the reference picture's band-done words (the fused stage then runs on the intra stream, beside that picture's deblocking)
// rows: the picture's deblocking launch will sit directly behind the previous one and wait on the device for this stage's rows (no event)
static int run_p_back(mi355enc_t *h, const frame_ctx_t *hc, slot_t *s, int prof, int split, const unsigned *gate, unsigned ref_epoch, int rows, bool defer_ip = false) {
    hipStream_t st = gate ? h->istream : h->stream;
    if (prof) HIPCHK(hipEventRecord(s->ev[8], st));
    if (false) { // (the two-kernel form of the High-profile path: absolute-vector refinement, 8x8 transform, no skip / intra logic -- kept for reference, reached through the stage entry points only)
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
            if (hc->intra_p && !defer_ip) k_launch_intra_p(hc, h->mbw, h->mbh, h->d_ip_progress, h->d_ip_strips, err_word(h), st); // (defer_ip: the caller enqueues the deblocking launch that carries them)
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
int enqueue_picture(mi355enc_t *h, slot_t *s, const uint8_t *src_y, const uint8_t *src_uv, int src_stride, int64_t pts, int force_idr) {
    const int idr = force_idr || !h->have_ref || h->frames_since_idr >= h->cfg.gop ||
                    (h->n_submitted == h->sc_force_at && h->frames_since_idr >= sc_lag(h)); // scene-cut recovery, see collect(): not when an IDR picture came in between
    if (idr) h->frames_since_idr = 0;
    // rate control: latch the setpoint written by the control thread, pick this picture's QP (and, below QP 51, its drop level)
    rc_set_bitrate(&h->rc, h->want_bps.load(std::memory_order_relaxed));
    int fq = h->fixed_qp.load(std::memory_order_relaxed);
    int qp, drop;
    if (fq >= 0) { qp = fq; drop = h->fixed_drop.load(std::memory_order_relaxed); s->rc_picked = 0; } // (no pick: collect() / recover() then leave rate control alone for this picture)
    else { rc_pick(&h->rc, idr, &qp, &drop); s->rc_picked = 1; }
    if (idr && drop == DROP_SKIP) drop = 0; // an IDR picture is never skipped; it has a ladder of its own
    const int all_skip = !idr && drop == DROP_SKIP;
    const int nxt = all_skip ? h->cur : (h->cur ^ 1); // an all-skip picture IS its reference: nothing is written
    const int set = (int)(h->n_submitted % NSET), ci = set;
    frame_ctx_t *c = s->h_ctx, *dctx = h->d_ctx2[ci];
    // stage timers: an event record costs ~5 us of queue time, so profile_events = k samples every k-th picture (IDR pictures always)
    // (a sampled picture runs its stages strictly in order; IDR pictures: every other one, or at the P pictures' cadence in an all-intra stream)
    const int prof = !all_skip && h->cfg.profile_events > 0 &&
                     ((idr && h->cfg.gop > 1) ? (h->idr_count & 1) == 0 : h->n_submitted % (uint64_t)h->cfg.profile_events == 0);
    const bool fused = true; // (r03: the High-profile stream goes through the fused stage too; the two-kernel form remains behind the single-stage entry points)
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
        if ((idr && h->cfg.intra_mode == 1) || h->cfg.deblock_mode != 0) HIPCHK(hipMemcpyAsync(dctx, c, sizeof *c, hipMemcpyHostToDevice, h->stream));
        // front stream: the source is in place (upload / conversion were enqueued there); P pictures: search, selection, gated intra
        // analysis; I pictures: only the padded source copy the next picture's search will run against
        if (c->qp_off) k_launch_aq(c, h->d_qp_off[set], h->fstream); // adaptive quantisation: the offsets of this picture's macroblocks
        // (r04 tried the IDR picture's open-loop analysis on the front stream as well -- it needs nothing but the source: all-intra 3 469 -> 3 463 frames/s, the flat analysis
        // launch beside the previous picture's rows kernel slows that kernel by what it saves; not kept.)
        if (idr) k_launch_copy_luma(c, h->fstream);
        else { int r = run_p_front(h, c, s, prof); if (r) return r; }
        HIPCHK(hipEventRecord(s->ev_front, h->fstream));
        h->psrc_cur ^= 1;
        // P picture with intra macroblocks: intra_p_kernel (a chain along rows, 10..80 us) and the band deblocker (a chain along
        // x + y) overlap -- intra_p_kernel runs on a stream of its own and the deblocker's movers follow its per-row progress words
        // (GATED, k_deblock.hip); the chain pmb -> prep -> deblocker -> next pmb stays on one stream (a cross-stream event on the
        // chain costs 10-17 us).  Not on pictures whose stage timers are sampled (a gated launch's duration includes its waiting).
        // Every kernel-waits-for-kernel overlap below is opt-in (cfg.exclusive_device): next to another process's kernels on the same GPU a
        // kernel that waits on the device for a kernel that has not been placed yet can run into the bound of its wait.
        const bool may_wait = overlap_allowed(h) && exclusive_device(h) && h->cfg.deblock_mode == 0 && !h->d_pre_y && !(prof && (idr || !h->cfg.profile_overlap)) && !(h->cfg.aq_mode && h->cfg.intra_in_p == 2); // (adaptive quantisation: the QP_Y chain
                                                                                                                                                    // over the whole picture sits between a picture's records and its deblocking)
        const int split = !idr && fused && c->intra_p && may_wait;
        // IDR picture: the band deblocker runs on the intra stream BESIDE the intra wavefront, each of its bands waiting for the intra bands
        // of the same rows (flags + acquire); in an all-intra stream the next picture's wavefront then starts while this one is still
        // being deblocked.  What has to wait for such a deblocking: a P picture (it reads the whole reference), and whoever writes
        // the reconstruction buffer it works on (the picture after next).
        // P picture whose reference is still being deblocked: the fused stage leaves the chain too.  It runs on the intra stream, each of
        // its waves waiting for the reference's bands it reads (pmb_kernel<GATED>), so it is all but done when that deblocking ends.
        const size_t nbd = k_deblock_done_bytes() / sizeof(unsigned); // words per reconstruction buffer
        const int pgate = !idr && fused && may_wait && h->rec_epoch[h->cur] != 0 && h->pgate;
        HIPCHK(hipStreamWaitEvent(pgate ? h->istream : h->stream, s->ev_front, 0));
        // ... and with three pictures in flight (the next picture's front stages are done long before this launch ends) the deblocking launches go back
        // to back, each waiting on the device for its picture's rows; with fewer the host sits on the chain and a launch waiting on the chip only
        // gets in the way (1080p depth 1: 4465 -> 3980 frames/s, 2160p: 2050 -> 1615)
        const int prows = pgate && h->cfg.pipeline_depth >= 2;
        const int isplit = idr && h->cfg.intra_mode != 1 && may_wait;
        // (Round 3 also built consecutive pictures' deblocking launches on two streams, the next picture's upper bands beside this one's lower
        // ones: +2 % at 1080p, slower at 2160p, and a bounded wait that ran out in two of four runs without a cause found -- removed in round 4;
        // what shortens the chain instead is slices with slice-local deblocking)
        hipStream_t mst = h->stream;
        bool fip = false;
        for (int b = 0; b < 2; b++)
            if (h->dbI_busy[b] && (!idr || b == nxt)) { HIPCHK(hipStreamWaitEvent(mst, h->ev_dbI[b], 0)); h->dbI_busy[b] = 0; }
        if (idr) {
            if (isplit) { HIPCHK(hipEventRecord(h->ev_pmb, h->stream)); HIPCHK(hipStreamWaitEvent(h->istream, h->ev_pmb, 0)); } // behind everything enqueued so far (a P picture's deblocker, its tables)
            if (prof) HIPCHK(hipEventRecord(s->ev[0], h->stream));
            int r = run_intra(h, ci, c, isplit ? h->d_iband_done + (size_t)nxt * h->mbh : nullptr); if (r) return r;
            if (prof) HIPCHK(hipEventRecord(s->ev[1], h->stream));
        } else {
            // One launch in flight, the intra macroblock rows inside it: the rows no longer queue behind the END of pmb_kernel (stream order) -- each starts when pmb_kernel
            // has completed its row and the row above, so the upper bands find them done when the launch starts.  Behind pmb_kernel in host order: nothing here waits for a
            // kernel that is not on the chip or in front of it in its own stream.  Not on sampled pictures (the timers bracket the launches in stream order).
            fip = prows && !c->qp_off && c->intra_p && !prof && h->fip_rows;
            int r = run_p_back(h, c, s, prof, split, pgate ? h->d_db_done + (size_t)h->cur * nbd : nullptr, h->rec_epoch[h->cur], prows, fip); if (r) return r;
            if (fip) {
                r = run_deblock(h, ci, c, mst, h->d_ip_progress, nullptr, h->d_db_done + (size_t)nxt * nbd, true, h->pmb_rows_total, true); if (r) return r;
                k_launch_wait_started(h->d_progress + 2, h->ip_done_total, err_word(h), h->istream); // records and levels are final once the rows have all counted themselves
            }
        }
        if (prof) HIPCHK(hipEventRecord(s->ev[2], h->stream));
        HIPCHK(hipGetLastError());
        if (c->qp_off && h->cfg.deblock_mode != 0) k_launch_qp_chain(h->d_mbi_set[set], h->nmb, qp, c->slice_rows * h->mbw, h->stream); // 7.4.5: QP_Y of the macroblocks without mb_qp_delta, for the deblocker (everything runs on the main stream here)
        HIPCHK(hipEventRecord(s->gpu_done, (split || pgate) ? h->istream : h->stream)); // records and levels are final here; they do not depend on deblocking
        if (h->d_pre_y) {
            HIPCHK(hipMemcpyAsync(h->d_pre_y, h->d_rec_y[nxt], h->ysz, hipMemcpyDeviceToDevice, h->stream));
            HIPCHK(hipMemcpyAsync(h->d_pre_uv, h->d_rec_uv[nxt], h->csz, hipMemcpyDeviceToDevice, h->stream));
            if (prof) HIPCHK(hipEventRecord(s->ev[2], h->stream));
        }
        if (isplit) {
            int r = run_deblock(h, ci, c, h->istream, nullptr, h->d_iband_done + (size_t)nxt * h->mbh, h->d_db_done + (size_t)nxt * nbd); if (r) return r;
            HIPCHK(hipEventRecord(h->ev_dbI[nxt], h->istream));
            h->dbI_busy[nxt] = 1;
        } else if (!fip) { int r = run_deblock(h, ci, c, mst, (split || (pgate && c->intra_p)) ? h->d_ip_progress : nullptr, nullptr, h->d_db_done + (size_t)nxt * nbd, prows != 0); if (r) return r; }
        h->rec_epoch[nxt] = h->cfg.deblock_mode == 0 ? c->epoch : 0;
        if (prof) { HIPCHK(hipEventRecord(s->ev[3], h->stream)); HIPCHK(hipEventRecord(s->ev[4], h->stream)); }
        // Hand-over, enqueued after the deblocking launches so that it cannot be dispatched ahead of them: the device packs the non-zero
        // blocks straight into the pinned host buffer while the band deblocker runs.
        hipStream_t pst = h->cstream;
        HIPCHK(hipStreamWaitEvent(h->cstream, s->gpu_done, 0));
        k_launch_pack(h->d_mbi_set[set], h->d_levels_set[set], h->nmb, h->mbw, h->d_off, s->h_mbi, s->h_levels, s->h_hdr, err_word(h), pst);
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(s->done, pst));
    }
    h->n_submitted++;
    s->is_idr = idr; s->qp = qp; s->drop = drop; s->frame_num = h->frames_since_idr; s->idr_pic_id = h->idr_count & 0xFFFF;
    s->src_y = src_y; s->src_uv = src_uv; s->src_stride = src_stride; s->force_idr = force_idr;
    s->pts = pts; s->rec_index = nxt; s->set = set; s->prof = prof; s->fused = fused && !idr; s->index = h->n_submitted - 1; s->all_skip = all_skip;
    if (idr) h->idr_count++;
    h->frames_since_idr++;
    h->cur = nxt; h->have_ref = 1; h->prev_slot = s;
    const int slot_index = (int)(s - h->slot);
    h->head = (h->head + 1) % NSLOT; h->pending++;
    if (h->wk_on) worker_push(h, slot_index);
    return MI355ENC_OK;
}

// Upload the planes of a non-NV12 picture tightly into the slot's raw staging buffer and convert into its NV12 staging surfaces.
int upload_and_convert(mi355enc_t *h, slot_t *s, int fmt, const uint8_t *const planes[3], const int strides[3], hipStream_t up) {
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

// ---- the entropy-coding worker: one picture at a time, in submission order
static size_t code_access_unit(mi355enc_t *h, slot_t *s, uint8_t *out, size_t cap) {
    size_t n = 0;
    if (s->is_idr) {
        n = h264_write_headers(out, cap, h->cfg.width, h->cfg.height, h->cfg.fps_num, h->cfg.fps_den, h->cfg.transform8x8);
        if (!n) return 0;
    }
    h264_writer_set_p_slices(h->writer, s->all_skip ? 0 : h->pslice_rows, h->slice_dbf); // (an all-skip picture is one run of P_Skip macroblocks in one slice)
    const size_t m = h264_write_slice_packed_rows(h->writer, out + n, cap - n, s->is_idr, s->frame_num, s->idr_pic_id, s->qp, s->h_mbi, s->h_levels, s->h_hdr + 2);
    return m ? n + m : 0;
}
void entropy_worker(mi355enc_t *h) {
    (void)hipSetDevice(h->cfg.device_id);
    for (;;) {
        int k;
        {
            std::unique_lock<std::mutex> g(h->wk_mu);
            h->wk_cv.wait(g, [&] { return h->wk_stop || h->wk_qh != h->wk_qt; });
            if (h->wk_stop) return;
            k = h->wk_q[h->wk_qh];
        }
        slot_t *s = &h->slot[k];
        int state = 2;
        if (!s->all_skip && hipEventSynchronize(s->done) != hipSuccess) state = 3;
        const double t0 = now_ms();
        if (state == 2 && s->h_hdr[1]) state = 3; // a device-side wait ran out: collect() recovers (and queues the pictures again)
        if (state == 2) s->au_len = code_access_unit(h, s, s->au, h->au_cap);
        s->au_ms = now_ms() - t0;
        {
            std::lock_guard<std::mutex> g(h->wk_mu);
            h->wk_qh = (h->wk_qh + 1) % (NSLOT + 1);
            s->au_state = state;
        }
        h->wk_done_cv.notify_all();
    }
}
static void worker_push(mi355enc_t *h, int slot_index) {
    { std::lock_guard<std::mutex> g(h->wk_mu); h->slot[slot_index].au_state = 1; h->wk_q[h->wk_qt] = slot_index; h->wk_qt = (h->wk_qt + 1) % (NSLOT + 1); }
    h->wk_cv.notify_one();
}
// waits until the worker has nothing queued or in hand (recover() re-enqueues pictures behind its back)
static void worker_drain(mi355enc_t *h) {
    std::unique_lock<std::mutex> g(h->wk_mu);
    h->wk_done_cv.wait(g, [&] { return h->wk_qh == h->wk_qt; });
}

static const char *wait_name(unsigned code) {
    switch (code & 255u) { // (the upper bits say where: band << 8, chroma << 15, macroblock column << 16)
    case 3: return "pmb_kernel waiting for the reference's deblocking bands";
    case 4: return "wait_started_kernel";
    case 11: return "deblocker waiting for the intra bands";
    case 12: return "deblocker waiting for the strips of the band above";
    case 13: return "deblocker waiting for intra_p_kernel";
    case 14: return "intra band waiting for the lines of the band above";
    case 15: return "intra_p_kernel waiting for the row above";
    case 16: return "progress counter";
    case 17: return "deblocker waiting for pmb_kernel's rows";
    case 22: return "QP_Y chain waiting for a macroblock row";
    case 23: return "deblocker waiting for the QP_Y chain";
    default: return "injected / unknown";
    }
}
// A bounded wait on the device ran out (the kernels report it in the sticky word d_progress[0], which the hand-over copies into h_hdr[1]; once
// it is set nobody waits any more, so everything in flight completes, possibly from data that was not final).  Nothing that was produced since
// can be trusted, but nothing needs to be lost either: the sources of the pictures in flight are still where submit() put them (the slots'
// staging surfaces, or the caller's device memory, which stays valid until the matching collect()).  So: drain, clear the device-side words,
// step one level down the ladder of mi355enc::safe_level, and enqueue the pictures in flight again, the first one as an IDR picture.  The
// stream continues without a gap; the decoder re-synchronises at that IDR picture.
static int recover(mi355enc_t *h, unsigned code) {
    h->safe_level++;
    h->n_recoveries++; h->last_error_word = code;
    fprintf(stderr, "mi355enc: a device-side wait timed out (error word %u: %s); %s\n", code, wait_name(code),
            h->safe_level == 1 ? "re-encoding the pictures in flight from an IDR picture; kernels run in stream order from now on" :
            h->safe_level == 2 ? "again: one launch per wavefront step from now on (no waits on the device at all)" : "giving up");
    if (h->safe_level > 2) return MI355ENC_ERR_HIP;
    if (h->wk_on) worker_drain(h); // (the pictures behind the failing one are coded and thrown away: they are enqueued again below)
    HIPCHK(hipStreamSynchronize(h->cstream));
    { int r = sync_compute(h); if (r) return r; }
    HIPCHK(hipMemsetAsync(h->d_progress, 0, 4 * sizeof(unsigned), h->stream));
    HIPCHK(hipMemsetAsync(h->d_row_done, 0, (size_t)h->mbh * MI355_PROG_STRIDE * sizeof(unsigned), h->stream));
    if (h->d_db_part) HIPCHK(hipMemsetAsync(h->d_db_part, 0, 6 * (size_t)k_deblock_bands16(h->mbh) * sizeof(unsigned), h->stream));
    HIPCHK(hipMemsetAsync(h->d_db_done, 0, 2 * k_deblock_done_bytes(), h->stream));
    HIPCHK(hipMemsetAsync(h->d_iband_done, 0, 2 * (size_t)h->mbh * sizeof(unsigned), h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->pmb_rows_total = 0; h->db_started_total = 0; h->ip_done_total = 0; h->qpc_total = 0; h->rec_epoch[0] = h->rec_epoch[1] = 0; h->dbI_busy[0] = h->dbI_busy[1] = 0;
    if (h->safe_level == 2) { h->cfg.deblock_mode = 1; h->cfg.intra_mode = 1; }
    const int n = h->pending;
    struct { const uint8_t *y, *uv; int stride, force_idr; int64_t pts; } again[NSLOT];
    for (int i = 0; i < n; i++) {
        slot_t *p = &h->slot[(h->tail + i) % NSLOT];
        again[i] = {p->src_y, p->src_uv, p->src_stride, p->force_idr, p->pts};
        if (p->is_idr) h->idr_count--;
        p->h_hdr[1] = 0;
        if (p->rc_picked) rc_cancel(&h->rc); // their rate-control bookings (rc_cancel takes back the newest one outstanding: only the count matters)
    }
    h->n_submitted -= (uint64_t)n; h->head = h->tail; h->pending = 0; h->have_ref = 0;
    for (int i = 0; i < n; i++) {
        int r = enqueue_picture(h, &h->slot[h->head], again[i].y, again[i].uv, again[i].stride, again[i].pts, i == 0 ? 1 : again[i].force_idr);
        if (r) return r;
    }
    return MI355ENC_OK;
}

// one piece of a pageable source picture: into the pinned staging buffer, then on its way to the device
static int stage_piece(mi355enc_t *h, const mi355enc::stage_job &j, hipStream_t up) {
    if (j.src_stride == (int)j.dst_stride) memcpy(j.dst, j.src, j.dst_stride * (size_t)(j.rows - 1) + (size_t)j.width);
    else for (int r = 0; r < j.rows; r++) memcpy(j.dst + (size_t)r * j.dst_stride, j.src + (size_t)r * j.src_stride, (size_t)j.width);
    return hipMemcpyAsync(j.dev, j.dst, j.dst_stride * (size_t)(j.rows - 1) + (size_t)j.width, hipMemcpyHostToDevice, up) == hipSuccess ? 0 : 1;
}
void stage_helper(mi355enc_t *h) {
    (void)hipSetDevice(h->cfg.device_id);
    unsigned long long seen = 0;
    for (;;) {
        {
            std::unique_lock<std::mutex> g(h->stg_mu);
            h->stg_cv.wait(g, [&] { return h->stg_stop || (h->stg_gen != seen && h->stg_next < h->stg_n); });
            if (h->stg_stop) return;
        }
        const hipStream_t up = upload_stream(h);
        for (;;) {
            int i;
            { std::lock_guard<std::mutex> g(h->stg_mu); seen = h->stg_gen; i = h->stg_next < h->stg_n ? h->stg_next++ : -1; }
            if (i < 0) break;
            const int e = stage_piece(h, h->stg_job[i], up);
            bool last;
            { std::lock_guard<std::mutex> g(h->stg_mu); if (e) h->stg_err = e; last = ++h->stg_done >= h->stg_n; }
            if (last) h->stg_done_cv.notify_all();
        }
    }
}
extern "C" {

// Host input.  A picture in memory from mi355enc_host_alloc() is DMA'd from where it lies (the call returns at once; the memory is the
// caller's again after the matching collect()).  Anything else is pageable as far as HIP knows: a stream-ordered copy from pageable memory
// blocks the calling thread while the runtime stages it chunk by chunk through its own pinned buffers -- so the picture is copied once, by
// this thread, into the slot's pinned staging buffer and leaves from there in one asynchronous transfer per plane, on the front stream,
// beside the kernels of the pictures before it.
int mi355enc_submit(mi355enc_t *h, const uint8_t *y, int y_stride, const uint8_t *uv, int uv_stride, int64_t pts, int force_idr) {
    if (!h || !y || !uv || y_stride < h->cfg.width || uv_stride < h->cfg.width) return MI355ENC_ERR_ARG;
    if (h->pending > h->cfg.pipeline_depth) return MI355ENC_ERR_STATE;
    HIPCHK(hipSetDevice(h->cfg.device_id));
    slot_t *s = &h->slot[h->head];
    const int w = h->cfg.width, ht = h->cfg.height;
    hipStream_t up = upload_stream(h);
    const bool pinned = host_range_pinned(y, (size_t)y_stride * (ht - 1) + w) && host_range_pinned(uv, (size_t)uv_stride * (ht / 2 - 1) + w);
    if (pinned || h->cfg.pipeline_depth == 0) {
        // pinned: transferred in place.  pipeline_depth 0 (the latency mode: collect() follows at once, there is nothing to run beside): the
        // runtime's own pageable path, which stages and transfers in chunks on its side of the call (measured 0.06 ms less per 1080p picture
        // than staging here and transferring afterwards)
        HIPCHK(hipMemcpy2DAsync(s->d_src_y, h->W, y, y_stride, w, ht, hipMemcpyHostToDevice, up));
        HIPCHK(hipMemcpy2DAsync(s->d_src_uv, h->W, uv, uv_stride, w, ht / 2, hipMemcpyHostToDevice, up));
        h->st.pinned_inputs += pinned ? 1 : 0;
    } else {
        if (!s->h_src) HIPCHK(hipHostMalloc((void **)&s->h_src, h->ysz + h->csz, hipHostMallocDefault));
        // rows at the coded stride, so that a range of rows is one contiguous transfer; in pieces (luma thirds or sixths, the chroma plane in one or two), each
        // sent as soon as it is staged: the transfer of one piece runs beside the staging of the next, and with the helper threads three pieces are
        // staged side by side (a single thread copies 3.1 MB in 0.15-0.2 ms, as long as the device needs for the whole picture)
        uint8_t *hy = s->h_src, *huv = s->h_src + (size_t)h->W * ht;
        mi355enc::stage_job jobs[8];
        int nj = 0;
        const int ny = h->stg_on ? 6 : 3, nc = h->stg_on ? 2 : 1;
        for (int k = 0; k < ny; k++) {
            const int r0 = (ht * k / ny) & ~1, r1 = k == ny - 1 ? ht : (ht * (k + 1) / ny) & ~1;
            if (r1 > r0) jobs[nj++] = {y + (size_t)r0 * y_stride, hy + (size_t)r0 * h->W, s->d_src_y + (size_t)r0 * h->W, y_stride, r1 - r0, w, (size_t)h->W};
        }
        for (int k = 0; k < nc; k++) {
            const int r0 = (ht / 2) * k / nc, r1 = (ht / 2) * (k + 1) / nc;
            if (r1 > r0) jobs[nj++] = {uv + (size_t)r0 * uv_stride, huv + (size_t)r0 * h->W, s->d_src_uv + (size_t)r0 * h->W, uv_stride, r1 - r0, w, (size_t)h->W};
        }
        if (h->stg_on) {
            { std::lock_guard<std::mutex> g(h->stg_mu); for (int i = 0; i < nj; i++) h->stg_job[i] = jobs[i]; h->stg_n = nj; h->stg_next = 0; h->stg_done = 0; h->stg_err = 0; h->stg_gen++; }
            h->stg_cv.notify_all();
            for (;;) { // the caller takes pieces too
                int i;
                { std::lock_guard<std::mutex> g(h->stg_mu); i = h->stg_next < h->stg_n ? h->stg_next++ : -1; }
                if (i < 0) break;
                const int e = stage_piece(h, h->stg_job[i], up);
                { std::lock_guard<std::mutex> g(h->stg_mu); if (e) h->stg_err = e; h->stg_done++; }
            }
            { std::unique_lock<std::mutex> g(h->stg_mu); h->stg_done_cv.wait(g, [&] { return h->stg_done >= h->stg_n; }); if (h->stg_err) return MI355ENC_ERR_HIP; }
        } else
            for (int i = 0; i < nj; i++) if (stage_piece(h, jobs[i], up)) return MI355ENC_ERR_HIP;
    }
    if (w != h->W) k_launch_pad(s->d_src_y, s->d_src_uv, h->W, w, ht, h->W, h->H, up);
    { int r = upload_done(h, s); if (r) return r; }
    return enqueue_picture(h, s, s->d_src_y, s->d_src_uv, h->W, pts, force_idr);
}

int mi355enc_submit_fmt(mi355enc_t *h, int fmt, const uint8_t *const planes[3], const int strides[3], int64_t pts, int force_idr) {
    if (!h || !planes || !strides) return MI355ENC_ERR_ARG;
    if (fmt == MI355ENC_FMT_NV12) return mi355enc_submit(h, planes[0], strides[0], planes[1], strides[1], pts, force_idr);
    if (h->pending > h->cfg.pipeline_depth) return MI355ENC_ERR_STATE;
    HIPCHK(hipSetDevice(h->cfg.device_id));
    slot_t *s = &h->slot[h->head];
    int r = upload_and_convert(h, s, fmt, planes, strides, upload_stream(h));
    if (r) return r;
    r = upload_done(h, s); if (r) return r;
    return enqueue_picture(h, s, s->d_src_y, s->d_src_uv, h->W, pts, force_idr);
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
    { int r = upload_done(h, s); if (r) return r; }
    return enqueue_picture(h, s, s->d_src_y, s->d_src_uv, h->W, pts, force_idr);
}

int mi355enc_collect(mi355enc_t *h, uint8_t *out, size_t out_cap, size_t *out_len, int *is_keyframe, int64_t *pts, int *qp) {
    if (!h || !out || !out_len) return MI355ENC_ERR_ARG;
    if (h->pending <= 0) return MI355ENC_ERR_STATE;
    HIPCHK(hipSetDevice(h->cfg.device_id));
    slot_t *s = &h->slot[h->tail];
    double t0 = now_ms();
    size_t n = 0, m = 0;
    if (h->wk_on) { // the worker codes the access unit; here it is only copied out
        for (int attempt = 0;; attempt++) {
            { std::unique_lock<std::mutex> g(h->wk_mu); h->wk_done_cv.wait(g, [&] { return s->au_state >= 2; }); }
            if (s->au_state == 2) break;
            if (attempt >= 2 || !s->h_hdr[1]) return MI355ENC_ERR_HIP;
            int r = recover(h, s->h_hdr[1]);
            if (r) return r;
            s = &h->slot[h->tail];
        }
        h->st.ms_wait += now_ms() - t0 - s->au_ms > 0 ? now_ms() - t0 - s->au_ms : 0;
        h->st.ms_entropy += s->au_ms;
        if (!s->au_len || s->au_len > out_cap) return MI355ENC_ERR_OVERFLOW;
        memcpy(out, s->au, s->au_len);
        m = s->au_len;
        s->au_state = 0;
    } else {
        if (!s->all_skip) HIPCHK(hipEventSynchronize(s->done));
        double t1 = now_ms();
        h->st.ms_wait += t1 - t0;
        for (int attempt = 0; s->h_hdr[1]; attempt++) { // set by a kernel of this or an earlier picture that gave up waiting on the device
            if (attempt >= 2) return MI355ENC_ERR_HIP;
            int r = recover(h, s->h_hdr[1]);
            if (r) return r;
            s = &h->slot[h->tail];
            if (!s->all_skip) HIPCHK(hipEventSynchronize(s->done));
        }
        m = code_access_unit(h, s, out, out_cap);
        if (!m) return MI355ENC_ERR_OVERFLOW;
        h->st.ms_entropy += now_ms() - t1;
    }
    *out_len = n + m;
    if (is_keyframe) *is_keyframe = s->is_idr;
    if (pts) *pts = s->pts;
    if (qp) *qp = s->qp;
    if (s->rc_picked) rc_update(&h->rc, s->is_idr, s->qp, s->drop, n + m); // picks and updates stay paired: a picture coded at a fixed QP booked nothing
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

} // extern "C"
