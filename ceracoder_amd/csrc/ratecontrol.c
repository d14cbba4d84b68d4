/*
 * ratecontrol.c -- one QP per picture, and below QP 51 one level of the drop ladder, from the bits/s setpoint that
 * ceracoder's balancer writes every 20 ms (/root/reference/src/ceracoder.c:237-264 ->
 * /root/reference/src/gst/encoder_control.c:45-57).  Setpoints are multiples of 100 kbit/s inside [300 kbit/s, 30 Mbit/s]
 * (/root/reference/src/core/bitrate_control.h:30-32, bitrate_control.c:206), and the balancer cuts hardest exactly when the
 * link is congested (bitrate_control.c:176-199): the whole range has to be honoured, not only the part a QP can reach.
 * Non-normative host logic (floating point allowed).
 *
 * Quantiser model: bits(vqp) ~= C / qstep(vqp), qstep = 2^((vqp-4)/6), C tracked separately for IDR and P pictures.  vqp is a
 * VIRTUAL quantiser: up to qp_max (51) it is the QP; above, every RC_DROP_DQ steps are one level of the ladder the macroblock
 * stages implement (mi355enc_dev.h DROP_MAX: P macroblocks whose prediction error is below a threshold carry no residual /
 * take the P_Skip vector; I pictures stop sending the residual of macroblocks that have little), and beyond the ladder's last
 * level a P picture is coded as one run of P_Skip macroblocks (a few bytes).  The tracker does not need to know what a level
 * is worth: it sees the bits that came out.  The quantiser may rise by 8 per picture but fall only by 2: on content whose
 * size is a cliff in QP (a still scene with sensor noise: nothing, then everything) a model-sized dive overshoots by orders
 * of magnitude.
 *
 * Allocation: GOP level, as in MPEG-2 TM5.  At every IDR picture the GOP is granted `gop` pictures' worth of the setpoint
 * (plus a bounded carry of what the previous GOP left or overspent); the IDR picture takes the share its complexity asks for --
 * capped by the VBV (x264enc's default vbv-buf-capacity, 600 ms of stream: an IDR may take half of it), floored by what an IDR
 * costs on the last ladder level -- and every P picture gets what is left divided by the pictures left.  A step on the
 * setpoint re-prices the pictures still to come.  So the bits of a GOP add up to the setpoint by construction, as long as
 * the last pictures of the GOP can still absorb the error -- which the all-skip picture guarantees in the downward direction.
 * `vbv` is the leaky bucket at the setpoint's rate; while it is nearly full the P pictures are all-skip.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "h264_host.h"

#define RC_DROP_DQ 2  /* virtual-QP steps per ladder level, P pictures */
#define RC_DROP_DQ_I 0.5 /* ... and I pictures: their ladder (residual of macroblocks that have little is not sent) is worth about a factor of two in all */
#define RC_VQP_MAX(rc) ((rc)->qp_max + RC_DROP_DQ * DROP_MAX)
#define RC_SKIP_BITS 160.0 /* an all-skip picture: slice header + one skip run */

static double qstep(double vqp) { return pow(2.0, (vqp - 4) / 6.0); }

void rc_init(rc_state_t *rc, double fps, int gop, uint32_t bps, int qp_min, int qp_max) {
    memset(rc, 0, sizeof *rc);
    rc->fps = fps > 0 ? fps : 30.0;
    rc->gop = gop > 0 ? gop : 1;
    rc->qp_min = qp_min; rc->qp_max = qp_max;
    rc->target_bps = bps;
    rc->vbv_ms = 600;
    rc->last_vqp_i = rc->last_vqp_p = 30;
}
void rc_set_vbv(rc_state_t *rc, int vbv_ms) { rc->vbv_ms = vbv_ms < 100 ? 100 : vbv_ms; }
void rc_set_bitrate(rc_state_t *rc, uint32_t bps) {
    if (bps < 1000) bps = 1000;
    if ((double)bps != rc->target_bps) {
        /* the pictures of this GOP still to come are re-priced; what is owed or saved so far scales with the rate, so that an
         * emergency drop is not delayed by debts run up at the old rate */
        const double k = (double)bps / rc->target_bps;
        rc->gop_bits = rc->gop_bits * k;
        rc->prev_rest *= k; rc->carry_used *= k;
        rc->target_bps = bps;
        /* pictures in flight were planned -- and will have been sized -- at the old rate: their account is settled in the new
         * rate's terms (rc_update) */
        for (uint32_t i = rc->n_upd; (int32_t)(rc->n_pick - i) > 0; i++) { rc->plan[i & 3] *= k; rc->plan_k[i & 3] *= k; } /* (wrap-safe: never more than the picks outstanding) */
        /* the bucket holds bits that are already on their way: a rate change does not change them.  After a cut they would
         * keep the stream frozen for seconds (the buffer shrinks with the rate); half the new buffer is what is kept */
        if (rc->vbv > 0.5 * bps * rc->vbv_ms / 1000.0) rc->vbv = 0.5 * bps * rc->vbv_ms / 1000.0;
        /* the quantiser follows the model at once when the rate falls (the per-picture limit on its rise would otherwise let a
         * few pictures through at the old size -- and they are the ones sent into the congestion the cut reacts to); when the
         * rate rises it is moved at most two octaves, the rest it walks (see the note on cliffs above) */
        int d = (int)lround(-6.0 * log2(k));
        if (d < -12) d = -12;
        /* A rise that starts at the coarse end of the scale -- on the ladder below the last quantiser, or within a few steps of it: there bits * qstep(virtual QP) says little
         * about what a real quantiser will cost (a ladder level is worth far less than the RC_DROP_DQ steps it is booked as; at QP 46+ a picture is headers and vectors), so the
         * tracker predicts QP 51 for a target that QP 30 would meet, "two octaves, then walk" is pulled back up by it, and the GOP that starts with the step ends short (measured:
         * 300 kbit/s -> 1 Mbit/s at 1080p60 0.87, 2160p60 0.89, the still scene 0.85; 1.5 -> 20 Mbit/s with three pictures in flight 0.77 in tools/rc_sim_oracle.py).  The tracker's
         * value from the last time the stream lived on real quantisers (QP <= 45: cplx_q) says where to go: the quantiser it predicts for a picture's share at the new rate, two
         * steps on the safe side, is taken at once when it lies below the bounded move, the tracker continues from that value, and the sizes of the pictures still in flight --
         * picked in the old regime -- no longer move it (rc_update: stale). */
        if (k > 1.0 && rc->have_q && rc->last_vqp_p > rc->qp_max - 6) {
            const double per = (double)bps / rc->fps;
            int vq = (int)lround(4.0 + 6.0 * log2(rc->cplx_q / (per > 1 ? per : 1))) + 2;
            if (vq < rc->qp_min) vq = rc->qp_min;
            if (vq < rc->last_vqp_p + d) { d = vq - rc->last_vqp_p; rc->cplx_p = rc->cplx_q; rc->cliff_age = 0; rc->regime++; rc->known_vqp_p = 0; rc->catchup = 3; } /* (catchup: the first pictures coded down here also pay for what the ladder and its skip runs left uncoded -- transients, not a cliff) */ /* (the model continues from the same value: the ladder's samples would pull the next pick back up) */
        }
        rc->last_vqp_p += d; rc->last_vqp_i += d > -12 ? d : -12;
        if (rc->last_vqp_p > RC_VQP_MAX(rc)) rc->last_vqp_p = RC_VQP_MAX(rc);
        if (rc->last_vqp_i > RC_VQP_MAX(rc)) rc->last_vqp_i = RC_VQP_MAX(rc);
        if (rc->last_vqp_p < rc->qp_min) rc->last_vqp_p = rc->qp_min;
        if (rc->last_vqp_i < rc->qp_min) rc->last_vqp_i = rc->qp_min;
    }
}
/* what a GOP inherits from the one before: a surplus up to a twentieth of a GOP; an overspent GOP is repaid over the following ones, a tenth
 * of a GOP at a time */
static double gop_carry(double rest, double G) { return rest > 0.05 * G ? 0.05 * G : rest < -0.10 * G ? -0.10 * G : rest; }
void rc_pick(rc_state_t *rc, int is_idr, int *qp, int *drop) {
    const double per = rc->target_bps / rc->fps, G = per * rc->gop, vbv_bits = rc->target_bps * rc->vbv_ms / 1000.0;
    *drop = 0;
    if (is_idr || rc->gop_left <= 0) { /* a new GOP: its grant, plus a bounded carry of the previous one's remainder */
        rc->prev_rest = rc->started ? rc->gop_bits : 0;
        rc->carry_used = gop_carry(rc->prev_rest, G);
        rc->gop_bits = G + rc->carry_used;
        rc->gop_serial++;
        rc->gop_left = rc->gop;
        rc->started = 1;
    }
    double target;
    if (is_idr && rc->gop > 1) {
        double ratio = (rc->have_i && rc->have_p && rc->cplx_p > 0) ? rc->cplx_i / rc->cplx_p : 4.0;
        if (ratio < 1.5) ratio = 1.5;
        if (ratio > 100.0) ratio = 100.0;
        target = rc->gop_bits * ratio / (rc->gop - 1 + ratio);
        if (target > 0.5 * vbv_bits) target = 0.5 * vbv_bits;                              /* VBV: an IDR may take half the buffer */
        const double floor_i = rc->have_i ? rc->cplx_i / qstep(rc->qp_max + RC_DROP_DQ_I * DROP_MAX) : 0; /* what an IDR costs on the last ladder level */
        if (target < floor_i) target = floor_i;
        if (target > 0.9 * rc->gop_bits) target = 0.9 * rc->gop_bits;
        if (target < per) target = per;
    } else target = rc->gop_bits / (rc->gop_left > 0 ? rc->gop_left : 1);
    const int vmax = RC_VQP_MAX(rc);
    const int exhausted = !is_idr && target < 0.25 * per; /* the GOP's grant is (nearly) used up: what such a picture is given says something about the books, nothing about the content */
    const double cplx = is_idr ? rc->cplx_i : rc->cplx_p;
    const int have = is_idr ? rc->have_i : rc->have_p;
    int skip = 0, vqp, idrop = 0;
    if (!is_idr && (rc->vbv > 0.9 * vbv_bits || target < 2 * RC_SKIP_BITS)) skip = 1; /* the bucket is full, or the GOP has nothing left */
    if (!have) {
        if (is_idr) vqp = rc->have_p ? rc->last_vqp_p - 2 : 32; /* first picture: nothing is known yet, start mid-range */
        else vqp = rc->last_vqp_i + 2;
    } else {
        const double q = cplx / (target > 1 ? target : 1); /* wanted qstep */
        vqp = (int)lround(4.0 + 6.0 * log2(q > 1e-6 ? q : 1e-6));
        if (is_idr) { /* IDR pictures are a second apart: the model decides, floored relative to where the P pictures are */
            const int lp = rc->last_vqp_p > rc->qp_max ? rc->qp_max : rc->last_vqp_p;
            if (rc->have_p && vqp < lp - 6) vqp = lp - 6;
            if (vqp > rc->qp_max) { /* onto the I ladder: half a quantiser step per level */
                const double v = 4.0 + 6.0 * log2(q > 1e-6 ? q : 1e-6);
                int d = (int)ceil((v - rc->qp_max) / RC_DROP_DQ_I);
                idrop = d < 1 ? 1 : d > DROP_MAX ? DROP_MAX : d;
                vqp = rc->qp_max;
            }
        } else {
            const int last = rc->last_vqp_p;
            /* Below the ladder: when even its last level is predicted to cost well over a picture's share, pictures are coded at
             * a regular cadence -- one in every (cost / share), the others as P_Skip runs -- rather than in bursts between long
             * freezes; what a coded picture really costs (more, the more pictures were skipped before it) comes back through the
             * complexity tracker. */
            const double at_vmax = cplx / qstep(vmax);
            if (at_vmax > 1.4 * target) {
                if (rc->since_real + 1 < 0.9 * at_vmax / (target > 1 ? target : 1)) skip = 1;
                else vqp = vmax;
            }
            if (vqp > last + 8) vqp = last + 8;
            /* a quantiser at which a picture cost several times its target is a cliff edge (a still scene with sensor noise codes
             * nothing, then everything): it is not visited again before the target could pay most of what it cost */
            if (rc->cliff_age > 0 && vqp <= rc->cliff_vqp && vqp <= rc->qp_max && target < 0.7 * rc->cliff_bits) vqp = rc->cliff_vqp + 1;
            /* downwards two steps a picture while the pictures are far below their target, one step once they are within a
             * factor of two of it: near a cliff a step of two is the difference between a tenth and ten times the target */
            const int down = (rc->last_bits_p > 0.5 * rc->last_target_p && last <= rc->qp_max) ? 1 : 2; /* (a ladder level is two steps) */
            if (vqp < last - down) vqp = last - down;
            /* Pictures in flight (pipeline depth 1 / 2: the sizes of the last one / two picks are not known yet).  A step down is an experiment, and
             * the model's C / qstep can be wrong by a factor of three to five PER STEP where the content has a cliff (the 4K clip: 23 KB at QP 29,
             * 73 at 28, 234 at 27 against a 42 KB share; walking on for two more pictures before the first size came back cost several pictures'
             * worth of bits, repaid with runs of P_Skip pictures: 88 of 480 in the model run of tests/test_abi_cpu.py).  So the walk may lead the
             * last quantiser whose size is KNOWN only by as many steps as could not cost more than twice this picture's share even if every
             * one of them multiplied the size by 3.3: far below the target that is two or three steps (the walk keeps its pace), at half the
             * target one, at the target none.  On the ladder below QP 51 a level is worth far less than that (RC_DROP_DQ), and the walk is left alone. */
            if (rc->n_pick != rc->n_upd && rc->known_vqp_p > 0 && rc->known_vqp_p <= rc->qp_max && rc->last_bits_p > 0) {
                const double room = 2.0 * target / rc->last_bits_p;
                int n = room > 1.0 ? (int)floor(log(room) / log(3.3)) : 0;
                if (n < 1 && target > 1.1 * rc->last_bits_p) { /* one step is the experiment itself: one picture takes it, and its size is waited for */
                    int out = 0;
                    for (uint32_t i = rc->n_upd; (int32_t)(rc->n_pick - i) > 0; i++) out |= rc->plan_vqp[i & 3] < rc->known_vqp_p;
                    if (!out) n = 1;
                }
                if (vqp < rc->known_vqp_p - n) vqp = rc->known_vqp_p - n;
            }
        }
    }
    static int trace = -1; /* dev aid: MI355ENC_RC_TRACE=1 prints every decision */
    if (trace < 0) trace = getenv("MI355ENC_RC_TRACE") != NULL;
    if (trace) fprintf(stderr, "rc idr=%d tgt=%.0f cplx=%.3g have=%d vqp=%d skip=%d since=%d vbv=%.0f known=%d lastb=%.0f lastv=%d gopb=%.0f left=%d np=%u nu=%u cliff=%d/%.0f/%d catchup=%d\n", is_idr, target, cplx, have, have ? vqp : -1, skip, rc->since_real, rc->vbv, rc->known_vqp_p, rc->last_bits_p, rc->last_vqp_p, rc->gop_bits, rc->gop_left, rc->n_pick, rc->n_upd, rc->cliff_vqp, rc->cliff_bits, rc->cliff_age, rc->catchup);
    const int gap_before = rc->since_real > 0x7FFF ? 0x7FFF : rc->since_real;
    if (!is_idr) rc->since_real = skip ? rc->since_real + 1 : 0;
    if (skip) {
        *qp = rc->qp_max; *drop = DROP_SKIP;
        target = RC_SKIP_BITS;
    } else {
        if (vqp < rc->qp_min) vqp = rc->qp_min;
        if (vqp > vmax) vqp = vmax;
        if (is_idr) { if (vqp > rc->qp_max) vqp = rc->qp_max; *qp = vqp; *drop = idrop; }
        else if (vqp <= rc->qp_max) *qp = vqp;
        else { *qp = rc->qp_max; *drop = (vqp - rc->qp_max + RC_DROP_DQ - 1) / RC_DROP_DQ; vqp = rc->qp_max + RC_DROP_DQ * *drop; }
        if (is_idr) rc->last_vqp_i = vqp;
        else if (!exhausted) rc->last_vqp_p = vqp; /* remembered when chosen: the picture's size arrives a picture later, after the next choice.  (Not the quantiser of a picture at the end
                                                     * of an exhausted GOP: it is driven to the ladder's last level by the books, and the next GOP would start its walk from there -- a dozen pictures
                                                     * at a level a picture: 1 Mbit/s at 1080p60, second GOP 0.93) */
    }
    rc->plan_vqp[rc->n_pick & 3] = (short)((is_idr || skip) ? 0x7FFF : vqp);
    rc->plan_gop[rc->n_pick & 3] = rc->gop_serial;
    rc->plan_gap[rc->n_pick & 3] = (short)(is_idr ? 0 : gap_before);
    rc->plan_k[rc->n_pick & 3] = 1.0;
    rc->plan_reg[rc->n_pick & 3] = rc->regime;
    rc->plan[rc->n_pick++ & 3] = target; /* booked now, corrected when the picture's size is known */
    rc->gop_bits -= target;
    rc->gop_left--;
}
/* takes the newest pick back (the picture will be picked again: a recovery re-enqueues the pictures in flight) */
void rc_cancel(rc_state_t *rc) {
    if ((int32_t)(rc->n_pick - rc->n_upd) <= 0) return;
    rc->gop_bits += rc->plan[--rc->n_pick & 3];
    rc->gop_left++;
}
int rc_pick_qp(rc_state_t *rc, int is_idr) {
    int qp, drop;
    rc_pick(rc, is_idr, &qp, &drop);
    return qp;
}
void rc_update(rc_state_t *rc, int is_idr, int qp, int drop, size_t bytes) {
    if ((int32_t)(rc->n_pick - rc->n_upd) <= 0) return; /* every update settles a pick (h264_host.h): a picture coded at a fixed QP has none */
    const double bits = 8.0 * (double)bytes;
    const int gap = rc->plan_gap[rc->n_upd & 3];
    const double kk = rc->plan_k[rc->n_upd & 3]; /* != 1: picked before the setpoint moved */
    const int stale = rc->plan_reg[rc->n_upd & 3] != rc->regime; /* picked on the ladder, before a rise that left it: says nothing about the quantisers now in use */
    const double planned = rc->plan[rc->n_upd++ & 3];
    if (drop != DROP_SKIP && !(stale && !is_idr)) {
        const double vqp = is_idr ? qp + RC_DROP_DQ_I * drop : qp + RC_DROP_DQ * drop;
        const double c = bits * qstep(vqp);
        if (is_idr) {
            rc->cplx_i = rc->have_i ? 0.5 * rc->cplx_i + 0.5 * c : c;
            rc->have_i = 1;
        } else {
            /* believe bad news faster than good news -- except the good news that the tracker's samples were taken after longer runs of
             * all-skip pictures than this one (leaving the cadence regime: those samples say nothing about pictures coded back to back);
             * and a picture coded on a lower ladder level than the one before it (or the one after that picture) also pays for what the higher level left
             * uncoded in the reference (53 -> 51 on the 1080p clip: 64 and 88 kbit, then 13 at QP 50): that says nothing about the level itself */
            if (rc->have_p && drop < rc->upd_drop_p) rc->catchup = 2;
            rc->upd_drop_p = drop;
            const int transient = rc->catchup > 0 && c > rc->cplx_p;
            if (rc->catchup > 0) rc->catchup--;
            const double a = transient ? 0.0 : (rc->have_p && c > rc->cplx_p) ? 0.6 : (rc->have_p && gap + 1 < 0.5 * (rc->cplx_gap + 1)) ? 0.8 : 0.35;
            rc->cplx_gap = rc->have_p ? (1 - a) * rc->cplx_gap + a * gap : gap;
            rc->cplx_p = rc->have_p ? (1 - a) * rc->cplx_p + a * c : c;
            rc->have_p = 1; rc->last_bits_p = bits; rc->last_target_p = planned; rc->known_vqp_p = (int)vqp;
            if (drop == 0 && gap == 0 && !transient && vqp <= rc->qp_max - 6) { rc->cplx_q = rc->have_q ? 0.5 * rc->cplx_q + 0.5 * c : c; rc->have_q = 1; } /* (a picture coded on a real quantiser, well inside the scale, right behind another coded one) */
            if (transient) { /* not a cliff either */ }
            else if (bits > 3.0 * planned && planned > 4 * RC_SKIP_BITS && vqp <= rc->qp_max && vqp < rc->last_vqp_p + 2 && gap == 0) { /* (gap: a picture behind a run of all-skip pictures carries their changes too) */ rc->cliff_vqp = (int)vqp; rc->cliff_bits = bits; rc->cliff_age = (int)rc->fps; }
            else if (rc->cliff_age > 0) {
                rc->cliff_age--;
                /* A remembered cliff keeps the quantiser one step above it while the target could not pay what the edge cost.  What is taken for an edge can be a transient (the
                 * first picture on a finer quantiser after the ladder also pays for the reference the ladder left: 59 kbit at QP 49 between pictures of 10-15 kbit at QP 50 and,
                 * later, at QP 49 as well -- the GOP that starts with 1 -> 1.5 Mbit/s then sat at QP 50 for 38 pictures at half its target: 0.79): eight pictures in a row right
                 * above the edge at under 60 % of their target and the edge is tried again (a real one is remembered again at the cost of one picture). */
                rc->cliff_doubt = ((int)vqp == rc->cliff_vqp + 1 && bits < 0.6 * planned) ? rc->cliff_doubt + 1 : 0;
                if (rc->cliff_doubt >= 8) { rc->cliff_age = 0; rc->cliff_doubt = 0; }
            }
        }
    }
    if (rc->plan_gop[(rc->n_upd - 1) & 3] == rc->gop_serial) rc->gop_bits += planned - bits * kk;
    else { /* a picture of the previous GOP (in flight when this one was granted): through the carry's cap, as if it had been known then */
        const double G = rc->target_bps / rc->fps * rc->gop;
        rc->prev_rest += planned - bits * kk;
        const double c = gop_carry(rc->prev_rest, G);
        rc->gop_bits += c - rc->carry_used;
        rc->carry_used = c;
    }
    rc->vbv += bits - rc->target_bps / rc->fps;
    /* the same forgiveness as in rc_set_bitrate for a picture that was already on its way when the setpoint was cut */
    if (kk < 1.0 && rc->vbv > 0.5 * rc->target_bps * rc->vbv_ms / 1000.0) rc->vbv = 0.5 * rc->target_bps * rc->vbv_ms / 1000.0;
    if (rc->vbv < 0) rc->vbv = 0; /* a CBR channel cannot send what has not been produced: the bucket does not go negative */
}
