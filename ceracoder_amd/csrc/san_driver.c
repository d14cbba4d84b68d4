/*
 * san_driver.c -- drives the host-only C of libmi355enc (h264_host.c, ratecontrol.c, tsmux.c) under the sanitizers
 * (the reference has none; `make sanitize` builds this file twice, -fsanitize=address,undefined and
 * -fsanitize=thread).  No HIP, no GStreamer: the same translation units the product links, compiled with gcc.
 *
 *   san_driver code CASE OUT THREADS   read a case file {int32 mbw, mbh, is_idr, frame_num, idr_pic_id, qp, t8, width,
 *                                      height, fps; mb records; dense levels}, write: parameter sets (IDR) + the slice through
 *                                      the dense writer, then the same access unit through the packed hand-over format on
 *                                      THREADS row-parallel threads (must be identical: exit 4 otherwise), then the access
 *                                      unit through the TS muxer.  OUT receives the access unit followed by the TS packets.
 *   san_driver race CASE ITERS         the threading the element has: a control thread storing bitrate setpoints into an
 *                                      atomic every few microseconds (g_object_set "bps" from the GLib main thread,
 *                                      /root/reference/src/ceracoder.c:266-295) while the streaming thread latches it into
 *                                      the rate control and codes the picture on 8 row-parallel threads, ITERS times.
 */
#include <pthread.h>
#include <stdatomic.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/mi355ts.h"
#include "h264_host.h"

typedef struct { int32_t mbw, mbh, is_idr, frame_num, idr_pic_id, qp, t8, width, height, fps; } case_hdr_t;
typedef struct { case_hdr_t h; mb_info_t *mbi; int16_t *levels; size_t nmb; } case_t;

static int load_case(const char *path, case_t *c) {
    FILE *f = fopen(path, "rb");
    if (!f) { perror(path); return -1; }
    if (fread(&c->h, sizeof c->h, 1, f) != 1) { fclose(f); return -1; }
    c->nmb = (size_t)c->h.mbw * c->h.mbh;
    c->mbi = (mb_info_t *)malloc(c->nmb * sizeof(mb_info_t));
    c->levels = (int16_t *)malloc(c->nmb * MB_LEVELS * sizeof(int16_t));
    int ok = c->mbi && c->levels && fread(c->mbi, sizeof(mb_info_t), c->nmb, f) == c->nmb &&
             fread(c->levels, MB_LEVELS * sizeof(int16_t), c->nmb, f) == c->nmb;
    fclose(f);
    return ok ? 0 : -1;
}

static int run_code(const char *in, const char *outp, int threads) {
    case_t c;
    if (load_case(in, &c)) return 2;
    const size_t cap = h264_max_au_bytes(c.h.mbw, c.h.mbh);
    uint8_t *a = (uint8_t *)malloc(cap), *b = (uint8_t *)malloc(cap);
    int16_t *packed = (int16_t *)malloc(c.nmb * PACK_BLOCKS_MAX * 32 + 32);
    uint32_t *row_off = (uint32_t *)malloc((size_t)c.h.mbh * sizeof(uint32_t));
    h264_writer_t *w = h264_writer_new(c.h.mbw, c.h.mbh, c.h.t8), *wp = h264_writer_new(c.h.mbw, c.h.mbh, c.h.t8);
    if (!a || !b || !packed || !row_off || !w || !wp || h264_writer_set_threads(wp, threads)) return 2;
    size_t n = 0;
    if (c.h.is_idr) n = h264_write_headers(a, cap, c.h.width, c.h.height, c.h.fps, 1, c.h.t8);
    size_t m = h264_write_slice(w, a + n, cap - n, c.h.is_idr, c.h.frame_num, c.h.idr_pic_id, c.h.qp, c.mbi, c.levels);
    if (!m) return 3;
    h264_pack_levels(c.h.mbw, c.h.mbh, c.mbi, c.levels, packed, row_off);
    size_t m2 = 0;
    for (int rep = 0; rep < 3; rep++) { /* the pool is reused picture after picture */
        m2 = h264_write_slice_packed_rows(wp, b, cap, c.h.is_idr, c.h.frame_num, c.h.idr_pic_id, c.h.qp, c.mbi, packed, row_off);
        if (m2 != m || memcmp(a + n, b, m)) { fprintf(stderr, "packed/row-parallel slice differs from the dense one (%zu vs %zu bytes)\n", m2, m); return 4; }
    }
    /* rate control: a few GOPs of feedback with this picture's size, setpoint steps in between */
    rc_state_t rc;
    rc_init(&rc, c.h.fps, 60, 6000000, 10, 51);
    for (int i = 0; i < 400; i++) {
        if (i % 97 == 0) rc_set_bitrate(&rc, (uint32_t)(300000 + (i * 7919) % 29700000));
        const int idr = i % 60 == 0;
        int qp, drop;
        rc_pick(&rc, idr, &qp, &drop);
        if (qp < 0 || qp > 51 || drop < 0 || (drop > DROP_MAX && drop != DROP_SKIP)) return 5;
        rc_update(&rc, idr, qp, drop, drop == DROP_SKIP ? 12 : (n + m) * (idr ? 6 : 1) * (size_t)(52 - qp) / 26 / (size_t)(1 + drop));
    }
    /* transport stream */
    mi355ts_t *ts = mi355ts_open();
    const size_t tcap = mi355ts_bound(n + m);
    uint8_t *t = (uint8_t *)malloc(tcap);
    size_t tn = 0;
    if (!ts || !t || mi355ts_mux(ts, a, n + m, 1000000000ll, c.h.is_idr, t, tcap, &tn) || tn % MI355TS_PACKET) return 6;
    FILE *f = fopen(outp, "wb");
    if (!f) return 2;
    fwrite(a, 1, n + m, f); fwrite(t, 1, tn, f);
    fclose(f);
    printf("{\"au_bytes\":%zu,\"ts_bytes\":%zu,\"threads\":%d}\n", n + m, tn, threads);
    mi355ts_close(ts); free(t);
    h264_writer_free(w); h264_writer_free(wp); free(a); free(b); free(packed); free(row_off); free(c.mbi); free(c.levels);
    return 0;
}

static _Atomic uint32_t want_bps = 6000000;
static _Atomic int stop_flag;
static void *setter_thread(void *arg) {
    (void)arg;
    uint32_t v = 300000;
    while (!atomic_load_explicit(&stop_flag, memory_order_relaxed)) {
        v = v >= 30000000 ? 300000 : v + 100000; /* the balancer's grid: multiples of 100 kbit/s in [300k, 30M] */
        atomic_store_explicit(&want_bps, v, memory_order_relaxed);
        for (volatile int spin = 0; spin < 200; spin++) { }
    }
    return NULL;
}
static int run_race(const char *in, int iters) {
    case_t c;
    if (load_case(in, &c)) return 2;
    const size_t cap = h264_max_au_bytes(c.h.mbw, c.h.mbh);
    uint8_t *a = (uint8_t *)malloc(cap);
    int16_t *packed = (int16_t *)malloc(c.nmb * PACK_BLOCKS_MAX * 32 + 32);
    uint32_t *row_off = (uint32_t *)malloc((size_t)c.h.mbh * sizeof(uint32_t));
    h264_writer_t *wp = h264_writer_new(c.h.mbw, c.h.mbh, c.h.t8);
    if (!a || !packed || !row_off || !wp || h264_writer_set_threads(wp, 8)) return 2;
    h264_pack_levels(c.h.mbw, c.h.mbh, c.mbi, c.levels, packed, row_off);
    rc_state_t rc;
    rc_init(&rc, c.h.fps, 60, atomic_load(&want_bps), 10, 51);
    pthread_t th;
    if (pthread_create(&th, NULL, setter_thread, NULL)) return 2;
    size_t first = 0;
    int bad = 0;
    for (int i = 0; i < iters; i++) {
        rc_set_bitrate(&rc, atomic_load_explicit(&want_bps, memory_order_relaxed)); /* latched once per picture, as enqueue_picture() does */
        int qp, drop;
        rc_pick(&rc, c.h.is_idr, &qp, &drop);
        const size_t m = h264_write_slice_packed_rows(wp, a, cap, c.h.is_idr, c.h.frame_num, c.h.idr_pic_id, c.h.qp, c.mbi, packed, row_off);
        if (!i) first = m;
        if (!m || m != first) bad = 1;
        rc_update(&rc, c.h.is_idr, qp, drop, m);
    }
    atomic_store(&stop_flag, 1);
    pthread_join(th, NULL);
    printf("{\"iters\":%d,\"slice_bytes\":%zu,\"stable\":%s}\n", iters, first, bad ? "false" : "true");
    h264_writer_free(wp); free(a); free(packed); free(row_off); free(c.mbi); free(c.levels);
    return bad ? 4 : 0;
}

int main(int argc, char **argv) {
    if (argc == 5 && !strcmp(argv[1], "code")) return run_code(argv[2], argv[3], atoi(argv[4]));
    if (argc == 4 && !strcmp(argv[1], "race")) return run_race(argv[2], atoi(argv[3]));
    fprintf(stderr, "usage: %s code CASE OUT THREADS | race CASE ITERS\n", argv[0]);
    return 2;
}
