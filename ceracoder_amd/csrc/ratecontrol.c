/*
 * ratecontrol.c -- one QP per picture from the bits/s setpoint that ceracoder's balancer
 * writes every 20 ms (/root/reference/src/ceracoder.c:237-264 ->
 * /root/reference/src/gst/encoder_control.c:45-57).  Setpoints are multiples of 100 kbit/s
 * inside [300 kbit/s, 30 Mbit/s] (/root/reference/src/core/bitrate_control.h:30-32,
 * bitrate_control.c:206).  Non-normative host logic (floating point allowed).
 *
 * Model: bits(qp) ~= C / qstep(qp), qstep = 2^((qp-4)/6), with C tracked separately for
 * IDR and P pictures; a virtual buffer pulls the per-picture budget back to the setpoint
 * within about half a second, so a step on `bps` is honoured well inside one GOP.
 */
#include <math.h>

#include "h264_host.h"

static double qstep(int qp) { return pow(2.0, (qp - 4) / 6.0); }

void rc_init(rc_state_t *rc, double fps, int gop, uint32_t bps, int qp_min, int qp_max) {
    rc->fps = fps > 0 ? fps : 30.0;
    rc->gop = gop > 0 ? gop : 1;
    rc->qp_min = qp_min; rc->qp_max = qp_max;
    rc->target_bps = bps;
    rc->cplx_i = rc->cplx_p = 0;
    rc->fullness = 0;
    rc->last_qp_i = rc->last_qp_p = 30;
    rc->frames_in_gop = 0;
    rc->have_i = rc->have_p = 0;
}
void rc_set_bitrate(rc_state_t *rc, uint32_t bps) {
    if (bps < 1000) bps = 1000;
    if ((double)bps != rc->target_bps) {
        /* keep the debt proportional to the new rate so an emergency drop is not delayed */
        rc->fullness *= (double)bps / rc->target_bps;
        rc->target_bps = bps;
    }
}
int rc_pick_qp(rc_state_t *rc, int is_idr) {
    const double per_frame = rc->target_bps / rc->fps;
    /* share of an IDR relative to a P picture, from the tracked complexities */
    double ratio = (rc->have_i && rc->have_p && rc->cplx_p > 0) ? rc->cplx_i / rc->cplx_p : 4.0;
    if (ratio < 1.5) ratio = 1.5;
    if (ratio > 12.0) ratio = 12.0;
    const double gop_bits = per_frame * rc->gop;
    const double p_bits = gop_bits / (rc->gop - 1 + ratio);
    double budget = is_idr ? p_bits * ratio : p_bits;
    /* buffer feedback: work the surplus/deficit off over ~fps/2 pictures */
    budget -= rc->fullness / (0.5 * rc->fps);
    if (budget < per_frame * 0.1) budget = per_frame * 0.1;
    double cplx = is_idr ? rc->cplx_i : rc->cplx_p;
    int have = is_idr ? rc->have_i : rc->have_p;
    int qp;
    if (!have) {
        if (is_idr) { /* first picture: bits-per-pixel heuristic is not available here, start mid-range */
            qp = rc->have_p ? rc->last_qp_p - 2 : 32;
        } else qp = rc->last_qp_i + 2;
    } else {
        double q = cplx / budget; /* wanted qstep */
        qp = (int)lround(4.0 + 6.0 * log2(q > 1e-6 ? q : 1e-6));
        int last = is_idr ? rc->last_qp_i : rc->last_qp_p;
        if (qp > last + 6) qp = last + 6;
        if (qp < last - 4) qp = last - 4;
    }
    if (qp < rc->qp_min) qp = rc->qp_min;
    if (qp > rc->qp_max) qp = rc->qp_max;
    return qp;
}
void rc_pick(rc_state_t *rc, int is_idr, int *qp, int *drop) {
    *qp = rc_pick_qp(rc, is_idr);
    *drop = 0;
}
void rc_update(rc_state_t *rc, int is_idr, int qp, size_t bytes) {
    const double bits = 8.0 * (double)bytes, c = bits * qstep(qp);
    if (is_idr) {
        rc->cplx_i = rc->have_i ? 0.5 * rc->cplx_i + 0.5 * c : c;
        rc->have_i = 1; rc->last_qp_i = qp; rc->frames_in_gop = 0;
    } else {
        rc->cplx_p = rc->have_p ? 0.7 * rc->cplx_p + 0.3 * c : c;
        rc->have_p = 1; rc->last_qp_p = qp;
    }
    rc->frames_in_gop++;
    rc->fullness += bits - rc->target_bps / rc->fps;
    /* bound the memory of the buffer to one second of stream either way */
    if (rc->fullness > rc->target_bps) rc->fullness = rc->target_bps;
    if (rc->fullness < -rc->target_bps) rc->fullness = -rc->target_bps;
}
