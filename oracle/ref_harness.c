/*
 * oracle/ref_harness.c -- TEST INFRASTRUCTURE: an SRT-free replay of ceracoder's main()
 * around the encoder element, built against the REFERENCE's own objects
 * (oracle/_ref/libceracoder_ref.so = /root/reference/src/core, src/gst/encoder_control.c,
 * src/io/pipeline_loader.c compiled unmodified).  It follows /root/reference/src/ceracoder.c:
 *   :466-470  pipeline_file_load + pipeline_create      (gst_parse_launch of a text file)
 *   :514-518  encoder_control_init + first set_bitrate while the pipeline is in state NULL
 *   :546-552  appsink callbacks {NULL, NULL, new_buf_cb}
 *   :592      g_timeout_add(20 ms) -> the bitrate writer (here: a scripted setpoint list
 *             instead of SRT statistics; libsrt is not in this image)
 *   :425-438  bus watch: ERROR / EOS stop the loop
 * Output file: records {u32 length, u64 pts_ns, bytes} per appsink sample.
 * Latency: t0 = buffer enters the encoder's sink pad (pad probe), t1 = the
 * sample reaches new_buf_cb, t2 = the last 1316-byte datagram of that sample has been sent.  The
 * sender regroups samples into 7 x 188-byte payloads like /root/reference/src/ceracoder.c:297-339
 * does for srt_send, but over UDP to a loopback socket (libsrt is absent: the t2 - t1 segment is
 * "UDP loopback, same packetisation", not SRT).
 *
 * Timestamp contract: an `identity name=ptsfixup signal-handoffs=TRUE` in the pipeline gets a handoff
 * that restates cb_ptsfixup (/root/reference/src/ceracoder.c:371-423: DTS forced to 0, PTS moved onto a grid of one
 * rolling-average frame period, pictures that arrive too early flagged DROPPABLE with their raw PTS); an
 * `identity name=jitter signal-handoffs=TRUE` in front of it gets a handoff that damages the timestamps the way a capture
 * device does (jitter of a few ms, repeated PTS, a picture stamped two periods early, garbage DTS).  The summary counts
 * what left the pipeline: samples whose DTS differs from their PTS and samples whose PTS ran backwards.
 *
 * usage: ref_harness PIPELINE_FILE OUT_FILE [SCRIPT_FILE]     (script lines: "<ms> <bps>")
 */
#include <arpa/inet.h>
#include <gst/app/gstappsink.h>
#include <gst/gst.h>
#include <netinet/in.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/socket.h>
#include <unistd.h>

#include "encoder_control.h"
#include "pipeline_loader.h"

#define BITRATE_UPDATE_INT 20 /* /root/reference/src/core/bitrate_control.h:35 */

static GMainLoop *loop;
static FILE *out;
static EncoderControl enc;
static guint64 n_samples, n_bytes;
static int exit_code;
static struct { long ms; int bps; } script[4096];
static int script_n, script_i;
static gint64 t_start;

/* ---- latency bookkeeping */
#define LAT_RING 256
#define LAT_MAX 65536
#define PKT_SIZE (188 * 7)
static struct { guint64 pts; gint64 t0; } lat_ring[LAT_RING];
static unsigned lat_head;
static GMutex lat_lock;
static float lat_enc[LAT_MAX], lat_send[LAT_MAX];
static unsigned lat_n;
static int udp_tx = -1, udp_rx = -1;
static struct sockaddr_in udp_dst;
static unsigned char pkt[PKT_SIZE];
static int pkt_len;
static guint64 n_datagrams;

/* ---- timestamp damage + the reference's repair (both optional, by element name) */
static guint64 n_dts_ne_pts, n_pts_backwards, n_pts_repeated, n_droppable_in, last_out_pts = GST_CLOCK_TIME_NONE;
static void jitter_cb(GstElement *identity, GstBuffer *buffer, gpointer user) {
    (void)identity; (void)user;
    static guint64 i, prev;
    guint64 pts = GST_BUFFER_PTS(buffer);
    gint64 d = ((gint64)((i * 7919u) % 5u) - 2) * 3 * GST_MSECOND; /* -6 .. +6 ms */
    if (i && i % 17 == 0) pts = prev;                              /* the same stamp twice */
    else if (i && i % 29 == 0) pts = pts > 40 * GST_MSECOND ? pts - 40 * GST_MSECOND : 0; /* far too early */
    else if ((gint64)pts + d > 0) pts = (guint64)((gint64)pts + d);
    GST_BUFFER_PTS(buffer) = pts;
    GST_BUFFER_DTS(buffer) = (i * 1000003u) % 7919u; /* garbage: the encoder must not look at it */
    prev = pts; i++;
}
static int sink_framerate(GstElement *e, int *num, int *den) {
    GstPad *pad = gst_element_get_static_pad(e, "sink");
    GstCaps *caps = pad ? gst_pad_get_current_caps(pad) : NULL;
    int ok = 0;
    if (caps && gst_caps_get_size(caps) > 0) ok = gst_structure_get_fraction(gst_caps_get_structure(caps, 0), "framerate", num, den);
    if (caps) gst_caps_unref(caps);
    if (pad) gst_object_unref(pad);
    return ok && *num > 0 && *den > 0;
}
static void ptsfixup_cb(GstElement *identity, GstBuffer *buffer, gpointer user) { /* restates ceracoder.c:371-423 */
    (void)user;
    static gint64 out_pts, period, prev_in;
    const gint64 in = (gint64)GST_BUFFER_PTS(buffer);
    GST_BUFFER_DTS(buffer) = 0;                                    /* :377 */
    if (out_pts == 0) {                                            /* first picture: nominal period from the caps */
        int n = 0, d = 0;
        if (sink_framerate(identity, &n, &d)) { out_pts = in; period = (gint64)GST_SECOND * d / n; }
    } else {
        period = (period * 997 + 500) / 1000 + ((in - prev_in) * 3 + 500) / 1000;  /* rolling average, weight 3/1000 */
        const gint64 diff = in - out_pts, incr = (diff / 2 + period) / period * period;
        if (incr > 0) { out_pts += incr; GST_BUFFER_PTS(buffer) = (guint64)out_pts; }
        else { GST_BUFFER_FLAG_SET(buffer, GST_BUFFER_FLAG_DROPPABLE); n_droppable_in++; } /* :414-419 */
    }
    prev_in = in;
}

static GstPadProbeReturn enc_sink_probe(GstPad *pad, GstPadProbeInfo *info, gpointer user) {
    (void)pad; (void)user;
    GstBuffer *b = GST_PAD_PROBE_INFO_BUFFER(info);
    if (b) {
        g_mutex_lock(&lat_lock);
        lat_ring[lat_head % LAT_RING].pts = GST_BUFFER_PTS(b);
        lat_ring[lat_head % LAT_RING].t0 = g_get_monotonic_time();
        lat_head++;
        g_mutex_unlock(&lat_lock);
    }
    return GST_PAD_PROBE_OK;
}
static void udp_open(void) {
    udp_rx = socket(AF_INET, SOCK_DGRAM, 0);
    udp_tx = socket(AF_INET, SOCK_DGRAM, 0);
    if (udp_rx < 0 || udp_tx < 0) { udp_tx = -1; return; }
    struct sockaddr_in a;
    memset(&a, 0, sizeof a);
    a.sin_family = AF_INET; a.sin_addr.s_addr = htonl(INADDR_LOOPBACK); a.sin_port = 0;
    socklen_t al = sizeof a;
    if (bind(udp_rx, (struct sockaddr *)&a, sizeof a) || getsockname(udp_rx, (struct sockaddr *)&a, &al)) { udp_tx = -1; return; }
    udp_dst = a; /* datagrams pile up in (and overflow) the never-read receive buffer; the sender does not care */
}
static void send_regrouped(const unsigned char *data, size_t n) {
    while (n) {
        size_t k = (size_t)(PKT_SIZE - pkt_len) < n ? (size_t)(PKT_SIZE - pkt_len) : n;
        memcpy(pkt + pkt_len, data, k);
        pkt_len += (int)k; data += k; n -= k;
        if (pkt_len == PKT_SIZE) {
            if (udp_tx >= 0) (void)sendto(udp_tx, pkt, PKT_SIZE, 0, (struct sockaddr *)&udp_dst, sizeof udp_dst);
            n_datagrams++;
            pkt_len = 0;
        }
    }
}
static int cmp_float(const void *a, const void *b) { float x = *(const float *)a, y = *(const float *)b; return (x > y) - (x < y); }
static void print_pct(const char *name, float *v, unsigned n) {
    if (!n) { printf(",\"%s\":null", name); return; }
    qsort(v, n, sizeof *v, cmp_float);
    printf(",\"%s\":{\"p50\":%.3f,\"p95\":%.3f,\"max\":%.3f,\"n\":%u}", name, v[n / 2], v[(unsigned)(n * 0.95)], v[n - 1], n);
}

static GstFlowReturn new_buf_cb(GstAppSink *sink, gpointer user) {
    (void)user;
    GstSample *sample = gst_app_sink_pull_sample(sink);
    if (!sample) return GST_FLOW_OK;
    GstBuffer *buf = gst_sample_get_buffer(sample);
    GstMapInfo map;
    const gint64 t1 = g_get_monotonic_time();
    if (gst_buffer_map(buf, &map, GST_MAP_READ)) {
        guint32 len = (guint32)map.size;
        guint64 pts = GST_BUFFER_PTS(buf);
        if (GST_BUFFER_DTS(buf) != pts) n_dts_ne_pts++;
        if (last_out_pts != GST_CLOCK_TIME_NONE && pts < last_out_pts) n_pts_backwards++;
        if (last_out_pts != GST_CLOCK_TIME_NONE && pts == last_out_pts) n_pts_repeated++;
        last_out_pts = pts;
        send_regrouped(map.data, map.size);
        const gint64 t2 = g_get_monotonic_time();
        gint64 t0 = -1;
        g_mutex_lock(&lat_lock);
        for (unsigned i = 0; i < LAT_RING && i < lat_head; i++)
            if (lat_ring[(lat_head - 1 - i) % LAT_RING].pts == pts) { t0 = lat_ring[(lat_head - 1 - i) % LAT_RING].t0; break; }
        g_mutex_unlock(&lat_lock);
        if (t0 >= 0 && lat_n < LAT_MAX) { lat_enc[lat_n] = (float)((t1 - t0) / 1e3); lat_send[lat_n] = (float)((t2 - t1) / 1e3); lat_n++; }
        fwrite(&len, 4, 1, out); fwrite(&pts, 8, 1, out); fwrite(map.data, 1, map.size, out);
        n_samples++; n_bytes += map.size;
        gst_buffer_unmap(buf, &map);
    }
    gst_sample_unref(sample);
    return GST_FLOW_OK;
}
static gboolean bus_cb(GstBus *bus, GstMessage *msg, gpointer user) {
    (void)bus; (void)user;
    if (GST_MESSAGE_TYPE(msg) == GST_MESSAGE_ERROR) {
        GError *e = NULL; gchar *dbg = NULL;
        gst_message_parse_error(msg, &e, &dbg);
        fprintf(stderr, "gstreamer error: %s (%s)\n", e->message, dbg ? dbg : "");
        g_error_free(e); g_free(dbg);
        exit_code = 3;
        g_main_loop_quit(loop);
    } else if (GST_MESSAGE_TYPE(msg) == GST_MESSAGE_EOS) g_main_loop_quit(loop);
    return TRUE;
}
static gboolean housekeeping(gpointer user) { /* the 20 ms control tick */
    (void)user;
    long now_ms = (long)((g_get_monotonic_time() - t_start) / 1000);
    while (script_i < script_n && script[script_i].ms <= now_ms) {
        if (encoder_control_available(&enc)) encoder_control_set_bitrate(&enc, script[script_i].bps);
        script_i++;
    }
    return TRUE;
}

int main(int argc, char **argv) {
    if (argc < 3) { fprintf(stderr, "usage: %s PIPELINE_FILE OUT_FILE [SCRIPT_FILE]\n", argv[0]); return 2; }
    if (argc > 3) {
        FILE *f = fopen(argv[3], "r");
        if (!f) { perror(argv[3]); return 2; }
        while (script_n < 4096 && fscanf(f, "%ld %d", &script[script_n].ms, &script[script_n].bps) == 2) script_n++;
        fclose(f);
    }
    gst_init(&argc, &argv);
    PipelineFile pf;
    if (pipeline_file_load(&pf, argv[1]) != 0) return 2;
    GstPipeline *pipeline = pipeline_create(&pf);
    if (!pipeline) return 2;
    out = fopen(argv[2], "wb");
    if (!out) { perror(argv[2]); return 2; }
    loop = g_main_loop_new(NULL, FALSE);
    GstBus *bus = gst_pipeline_get_bus(pipeline);
    gst_bus_add_watch(bus, bus_cb, NULL);
    int have_enc = encoder_control_init(&enc, pipeline) == 0;
    int first = script_n ? script[0].bps : 6000000;
    if (have_enc && g_object_class_find_property(G_OBJECT_GET_CLASS(enc.element), "bitrate")) { /* what the pipeline text itself set, before the first control write */
        guint b0 = 0, k0 = 0;
        g_object_get(G_OBJECT(enc.element), "bps", &b0, "bitrate", &k0, NULL);
        fprintf(stderr, "{\"bps_from_pipeline_text\":%u,\"bitrate_kbps_from_pipeline_text\":%u}\n", b0, k0);
    }
    if (have_enc) encoder_control_set_bitrate(&enc, first); /* state NULL, as ceracoder.c:515-518 */
    guint bps_prop = 0, kbps_prop = 0;
    if (have_enc) g_object_get(G_OBJECT(enc.element), "bps", &bps_prop, NULL);
    if (have_enc && g_object_class_find_property(G_OBJECT_GET_CLASS(enc.element), "bitrate")) g_object_get(G_OBJECT(enc.element), "bitrate", &kbps_prop, NULL);
    fprintf(stderr, "{\"encoder_found\":%d,\"bitrate_div\":%d,\"bps_after_null_state_write\":%u,\"bitrate_kbps\":%u}\n", have_enc, enc.bitrate_div, bps_prop, kbps_prop);
    if (have_enc) {
        GstPad *sp = gst_element_get_static_pad(enc.element, "sink");
        if (sp) { gst_pad_add_probe(sp, GST_PAD_PROBE_TYPE_BUFFER, enc_sink_probe, NULL, NULL); gst_object_unref(sp); }
    }
    {
        GstElement *j = gst_bin_get_by_name(GST_BIN(pipeline), "jitter"), *f = gst_bin_get_by_name(GST_BIN(pipeline), "ptsfixup");
        if (j) { g_signal_connect(j, "handoff", G_CALLBACK(jitter_cb), NULL); gst_object_unref(j); }
        if (f) { g_signal_connect(f, "handoff", G_CALLBACK(ptsfixup_cb), NULL); gst_object_unref(f); } /* ceracoder.c:536-543 */
    }
    udp_open();
    GstElement *sink = gst_bin_get_by_name(GST_BIN(pipeline), "appsink");
    if (!sink) { fprintf(stderr, "no element named appsink\n"); return 2; }
    GstAppSinkCallbacks cbs = {NULL, NULL, new_buf_cb, {0}};
    gst_app_sink_set_callbacks(GST_APP_SINK(sink), &cbs, NULL, NULL);
    t_start = g_get_monotonic_time();
    g_timeout_add(BITRATE_UPDATE_INT, housekeeping, NULL);
    gst_element_set_state(GST_ELEMENT(pipeline), GST_STATE_PLAYING);
    g_main_loop_run(loop);
    double secs = (g_get_monotonic_time() - t_start) / 1e6;
    gst_element_set_state(GST_ELEMENT(pipeline), GST_STATE_NULL);
    fclose(out);
    printf("{\"samples\":%" G_GUINT64_FORMAT ",\"bytes\":%" G_GUINT64_FORMAT ",\"seconds\":%.3f,\"setpoints_applied\":%d,\"datagrams_1316\":%" G_GUINT64_FORMAT,
           n_samples, n_bytes, secs, script_i, n_datagrams);
    printf(",\"dts_ne_pts\":%" G_GUINT64_FORMAT ",\"pts_backwards\":%" G_GUINT64_FORMAT ",\"pts_repeated\":%" G_GUINT64_FORMAT ",\"droppable_in\":%" G_GUINT64_FORMAT,
           n_dts_ne_pts, n_pts_backwards, n_pts_repeated, n_droppable_in);
    unsigned skip = lat_n > 90 ? 60 : 0; /* discard the first GOP (warm-up) when the run is long enough */
    print_pct("ms_encoder_sink_to_appsink", lat_enc + skip, lat_n - skip);
    print_pct("ms_appsink_to_last_udp_send", lat_send + skip, lat_n - skip);
    printf("}\n");
    if (udp_tx >= 0) close(udp_tx);
    if (udp_rx >= 0) close(udp_rx);
    pipeline_file_unload(&pf);
    return exit_code;
}
