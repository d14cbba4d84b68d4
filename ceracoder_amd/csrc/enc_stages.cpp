// enc_stages.cpp -- single-stage entry points of the C ABI (the same kernels the encoder launches, one stage at a time: parity
// tests and probes) and the host-only stages (parameter sets, slice writer, rate-control model: no device needed).
#include "enc_internal.hpp"

// A single-stage call made with pictures in flight would overwrite what they use: refused BEFORE anything is uploaded (the caller collects first).
#define STAGE_IDLE(h) do { if ((h)->pending) return MI355ENC_ERR_STATE; } while (0)

extern "C" {

int mi355enc_stage_csc(mi355enc_t *h, int fmt, const uint8_t *const planes[3], const int strides[3], uint8_t *out_y, uint8_t *out_uv) {
    if (!h || !out_y || !out_uv || h->pending) return MI355ENC_ERR_ARG;
    HIPCHK(hipSetDevice(h->cfg.device_id));
    STAGE_IDLE(h);
    slot_t *s = &h->slot[0];
    int r = upload_and_convert(h, s, fmt, planes, strides, h->stream);
    if (r) return r;
    HIPCHK(hipMemcpyAsync(out_y, s->d_src_y, h->ysz, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(out_uv, s->d_src_uv, h->csz, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return MI355ENC_OK;
}

// ---------------------------------------------------------------- single-stage entry points
// A single-stage call overwrites the reconstruction buffers, the previous-source planes and the record sets behind the encoder's back: the
// next picture submitted through the pipeline must not predict from any of it, so it is coded as an IDR picture and nothing of the
// previous picture's on-device progress words is trusted.
static void stage_touched(mi355enc_t *h) { h->have_ref = 0; h->rec_epoch[0] = h->rec_epoch[1] = 0; h->dbI_busy[0] = h->dbI_busy[1] = 0; }
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
    c->slice_rows = h->stage_slice_rows; c->slice_dbf = h->stage_slice_dbf;
    c->all_intra = 0; // the single-stage deblocking entry point takes records of either picture type
    c->qp_off = nullptr; // (single stages: one QP per picture)
    HIPCHK(hipMemcpyAsync(h->d_ctx, c, sizeof *c, hipMemcpyHostToDevice, h->stream));
    stage_touched(h);
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
    STAGE_IDLE(h);
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
    STAGE_IDLE(h);
    int r = stage_ctx(h, qp, true); if (r) return r;
    HIPCHK(hipMemcpyAsync(h->d_surf[0], surf, (size_t)h->nmb * SURF_U16 * sizeof(uint16_t), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_imv[0][0], imv_in, (size_t)h->nmb * sizeof(imv_t), hipMemcpyHostToDevice, h->stream));
    k_launch_me_select(h->slot[0].h_ctx, h->mbw, 0, h->mbh, h->d_imv[0][0], h->d_imv[0][1], nullptr, 0, h->stream);
    HIPCHK(hipMemcpyAsync(imv_out, h->d_imv[0][1], (size_t)h->nmb * sizeof(imv_t), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return MI355ENC_OK;
}
// ... and the form the encoder's later iterations use (me_select_sparse_kernel): imv_prev is the field imv_in was selected from -- a macroblock whose predictors are
// the same in both copies its entry of imv_in.  Must equal mi355enc_stage_me_select(surf, imv_in) whenever imv_in really is the selection over imv_prev.
int mi355enc_stage_me_select_next(mi355enc_t *h, const uint16_t *surf, const void *imv_in, const void *imv_prev, int qp, void *imv_out) {
    if (!h || !surf || !imv_in || !imv_prev || !imv_out || qp < 0 || qp > 51) return MI355ENC_ERR_ARG;
    HIPCHK(hipSetDevice(h->cfg.device_id));
    STAGE_IDLE(h);
    int r = stage_ctx(h, qp, true); if (r) return r;
    HIPCHK(hipMemcpyAsync(h->d_surf[0], surf, (size_t)h->nmb * SURF_U16 * sizeof(uint16_t), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_imv[0][0], imv_in, (size_t)h->nmb * sizeof(imv_t), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_imv[0][2], imv_prev, (size_t)h->nmb * sizeof(imv_t), hipMemcpyHostToDevice, h->stream));
    k_launch_me_select(h->slot[0].h_ctx, h->mbw, 0, h->mbh, h->d_imv[0][0], h->d_imv[0][1], h->d_imv[0][2], 2, h->stream);
    HIPCHK(hipMemcpyAsync(imv_out, h->d_imv[0][1], (size_t)h->nmb * sizeof(imv_t), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return MI355ENC_OK;
}
int mi355enc_stage_subpel(mi355enc_t *h, const uint8_t *cur_y, const uint8_t *ref_y, int qp, void *mbinfo_inout) {
    if (!h || !cur_y || !ref_y || !mbinfo_inout || qp < 0 || qp > 51) return MI355ENC_ERR_ARG;
    HIPCHK(hipSetDevice(h->cfg.device_id));
    STAGE_IDLE(h);
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
    STAGE_IDLE(h);
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
    STAGE_IDLE(h);
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
    STAGE_IDLE(h);
    HIPCHK(hipMemcpyAsync(h->slot[0].d_src_y, src_y, h->ysz, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->slot[0].d_src_uv, src_uv, h->csz, hipMemcpyHostToDevice, h->stream));
    int r = stage_ctx(h, qp, true, drop, 1); if (r) return r;
    r = run_intra(h, 0, h->slot[0].h_ctx); if (r) return r;
    return download_picture(h, mbinfo_out, rec_y, rec_uv, levels) ? MI355ENC_ERR_HIP : MI355ENC_OK;
}
int mi355enc_stage_intra_analyse(mi355enc_t *h, const uint8_t *src_y, const uint8_t *src_uv, int qp, uint16_t *isad_out, void *idec_out) {
    if (!h || !src_y || !src_uv || !isad_out || qp < 0 || qp > 51) return MI355ENC_ERR_ARG;
    HIPCHK(hipSetDevice(h->cfg.device_id));
    STAGE_IDLE(h);
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
    STAGE_IDLE(h);
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
int mi355enc_debug_trip_wait(mi355enc_t *h, unsigned code) {
    if (!h || !code) return MI355ENC_ERR_ARG;
    HIPCHK(hipSetDevice(h->cfg.device_id));
    HIPCHK(hipStreamSynchronize(h->cstream));
    { int r = sync_compute(h); if (r) return r; }
    HIPCHK(hipMemcpy(h->d_progress, &code, sizeof code, hipMemcpyHostToDevice));
    return MI355ENC_OK;
}
int mi355enc_time_stage(mi355enc_t *h, int stage, int iters, double *avg_ms) {
    if (!h || !avg_ms || iters < 1 || stage < 0 || stage > 10) return MI355ENC_ERR_ARG;
    if (h->pending) return MI355ENC_ERR_STATE;
    HIPCHK(hipSetDevice(h->cfg.device_id));
    slot_t *s = &h->slot[0];
    // a valid context in both places: the LAST COLLECTED picture's host copy (the slots rotate: it is slot 0's only one time in three), brought
    // to slot 0 and re-uploaded -- or a fresh stage context.  The stages run on that picture's surfaces in place (stage 3 filters its
    // reconstruction again), so the next picture through the pipeline starts a new GOP.
    if (!h->have_ref || !h->last_slot) { int r = stage_ctx(h, 26, true); if (r) return r; }
    else {
        { int r = sync_compute(h); if (r) return r; }
        HIPCHK(hipStreamSynchronize(h->cstream));
        if (h->last_slot != s) *s->h_ctx = *h->last_slot->h_ctx;
        HIPCHK(hipMemcpyAsync(h->d_ctx, s->h_ctx, sizeof(frame_ctx_t), hipMemcpyHostToDevice, h->stream));
        stage_touched(h);
    }
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
            else if (stage == 8) k_launch_me_select(h->slot[0].h_ctx, h->mbw, 0, h->mbh, h->d_imv[0][0], h->d_imv[0][1], nullptr, 0, h->stream);
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
static int g_host_slice_rows = 0;
static int g_host_pslice_rows = 0, g_host_dbf_idc = 0;
void mi355enc_host_set_slice_rows(int rows) { g_host_slice_rows = rows > 0 ? rows : 0; }
void mi355enc_host_set_p_slices(int rows, int dbf_idc) { g_host_pslice_rows = rows > 0 ? rows : 0; g_host_dbf_idc = dbf_idc == 2 ? 2 : 0; }
int mi355enc_stage_set_slice_rows(mi355enc_t *h, int rows) { if (!h || rows < 0) return MI355ENC_ERR_ARG; h->stage_slice_rows = rows; return MI355ENC_OK; }
int mi355enc_slice_rows(const mi355enc_t *h) { return h ? h->islice_rows : 0; }
int mi355enc_p_slice_rows(const mi355enc_t *h) { return h ? h->pslice_rows : 0; }
int mi355enc_stage_set_slice_deblock(mi355enc_t *h, int idc) { if (!h || (idc != 0 && idc != 2)) return MI355ENC_ERR_ARG; h->stage_slice_dbf = idc; return MI355ENC_OK; }
int mi355enc_host_write_slice(int mbw, int mbh, int is_idr, int frame_num, int idr_pic_id, int qp, int t8, const void *mbinfo,
                              const int16_t *levels, uint8_t *out, size_t cap, size_t *out_len) {
    if (!out || !out_len || !mbinfo || !levels || mbw < 1 || mbh < 1 || qp < 0 || qp > 51) return MI355ENC_ERR_ARG;
    h264_writer_t *w = h264_writer_new(mbw, mbh, t8);
    if (!w) return MI355ENC_ERR_NOMEM;
    h264_writer_set_slice_rows(w, g_host_slice_rows);
    h264_writer_set_p_slices(w, g_host_pslice_rows, g_host_dbf_idc);
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
        h264_writer_set_slice_rows(w, g_host_slice_rows);
        h264_writer_set_p_slices(w, g_host_pslice_rows, g_host_dbf_idc);
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
