/*
 * mi355_gst_probe.c -- product-side latency probe for the metric "appsink -> SRT p50 frame latency"
 * (BASELINE.json).  Runs a gst-launch style description that must contain an element named
 * `venc_bps` (the encoder) and one named `appsink`, and stamps every buffer three times:
 *   t0  buffer enters the encoder's sink pad            (pad probe)
 *   t1  the muxed sample reaches the appsink callback   (what /root/reference/src/ceracoder.c:297 new_buf_cb receives)
 *   t2  the last 1316-byte datagram of that sample has been handed to the socket
 * The sender regroups samples into 7 x 188-byte payloads exactly like new_buf_cb does for srt_send
 * (/root/reference/src/ceracoder.c:48-51,313-332) but over UDP to a loopback socket: libsrt is not in this image, so
 * t2 - t1 is "UDP loopback, same packetisation", not SRT.  Prints one JSON line.
 *
 * Throughput (SURVEY.md 8d "Timing method"): with a non-live source the same run gives frames/s as wall-clock between buffers arriving at the
 * sink behind the encoder, the first GOP (60 buffers) discarded: "fps_after_first_gop".  With --no-encoder the description needs no
 * `venc_bps` (the source's own ceiling: `videotestsrc ! appsink`).
 *
 * With --appsrc N W H [pinned] the description starts with `appsrc name=src`: the probe feeds N NV12 pictures itself (eight
 * pre-rendered pictures of a panning texture, wrapped without a copy, as fast as the pipeline takes them), so that the figure is the
 * element's and not the test source's (videotestsrc paints 1080p at 300-400 pictures/s).  `pinned`: the pictures lie in
 * mi355enc_host_alloc memory, as a capture source that adopted the element's buffer pool would deliver them.  `--clip FILE`: the pictures come from FILE (raw NV12
 * pictures of W x H one after the other, at most 32) instead of being rendered here -- bench.py passes the pictures of its own clip, so that this leg and the C-ABI legs
 * code the same content.
 *
 * With --props the description is parsed, the coding-tool properties of the encoder element (named venc_bps or venc_kbps) are printed as they will be used --
 * an explicit property, else what speed-preset selects -- and nothing runs (no device needed).
 *
 * usage: mi355_gst_probe "PIPELINE DESCRIPTION" [--no-encoder | --appsrc N W H [pinned] | --props]
 */
#include <arpa/inet.h>
#include <gst/app/gstappsink.h>
#include <gst/app/gstappsrc.h>
#include <dlfcn.h>
#include <libgen.h>
#include <gst/gst.h>
#include <netinet/in.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/socket.h>
#include <unistd.h>

#define RING 512
#define MAXN 65536
#define PKT (188 * 7)

static GMainLoop *loop;
static struct { guint64 pts; gint64 t0; } ring[RING];
static unsigned head;
static GMutex lock;
static float l_enc[MAXN], l_send[MAXN];
static gint64 t_arr[MAXN]; /* arrival of every sample at the sink */
static unsigned n_lat;
static guint64 n_samples, n_bytes, n_dgrams;
static gint64 us_in_callback; /* time the sink's streaming thread spends in on_sample (pull, map, 1316-byte regrouping, sendto) */
static volatile gint n_out; /* samples that reached the sink (read by the feeder) */
static int tx = -1, rx = -1, exit_code;
static struct sockaddr_in dst;
static unsigned char pkt[PKT];
static int pkt_len;

static GstPadProbeReturn on_enc_sink(GstPad *pad, GstPadProbeInfo *info, gpointer u) {
    (void)pad; (void)u;
    GstBuffer *b = GST_PAD_PROBE_INFO_BUFFER(info);
    if (b) {
        g_mutex_lock(&lock);
        ring[head % RING].pts = GST_BUFFER_PTS(b); ring[head % RING].t0 = g_get_monotonic_time(); head++;
        g_mutex_unlock(&lock);
    }
    return GST_PAD_PROBE_OK;
}
static void send_regrouped(const unsigned char *d, size_t n) {
    while (n) {
        size_t k = (size_t)(PKT - pkt_len) < n ? (size_t)(PKT - pkt_len) : n;
        memcpy(pkt + pkt_len, d, k);
        pkt_len += (int)k; d += k; n -= k;
        if (pkt_len == PKT) {
            if (tx >= 0) (void)sendto(tx, pkt, PKT, 0, (struct sockaddr *)&dst, sizeof dst);
            n_dgrams++; pkt_len = 0;
        }
    }
}
static GstFlowReturn on_sample(GstAppSink *sink, gpointer u) {
    (void)u;
    const gint64 tc = g_get_monotonic_time();
    GstSample *s = gst_app_sink_pull_sample(sink);
    if (!s) return GST_FLOW_OK;
    const gint64 t1 = g_get_monotonic_time();
    GstBuffer *b = gst_sample_get_buffer(s);
    GstMapInfo m;
    if (gst_buffer_map(b, &m, GST_MAP_READ)) {
        send_regrouped(m.data, m.size);
        const gint64 t2 = g_get_monotonic_time();
        const guint64 pts = GST_BUFFER_PTS(b);
        gint64 t0 = -1;
        g_mutex_lock(&lock);
        for (unsigned i = 0; i < RING && i < head; i++)
            if (ring[(head - 1 - i) % RING].pts == pts) { t0 = ring[(head - 1 - i) % RING].t0; break; }
        g_mutex_unlock(&lock);
        if (t0 >= 0 && n_lat < MAXN) { l_enc[n_lat] = (float)((t1 - t0) / 1e3); l_send[n_lat] = (float)((t2 - t1) / 1e3); n_lat++; }
        if (n_samples < MAXN) t_arr[n_samples] = t1;
        n_samples++; n_bytes += m.size;
        g_atomic_int_inc(&n_out);
        gst_buffer_unmap(b, &m);
    }
    gst_sample_unref(s);
    us_in_callback += g_get_monotonic_time() - tc;
    return GST_FLOW_OK;
}
static gboolean on_bus(GstBus *bus, GstMessage *msg, gpointer u) {
    (void)bus; (void)u;
    if (GST_MESSAGE_TYPE(msg) == GST_MESSAGE_ERROR) {
        GError *e = NULL; gchar *dbg = NULL;
        gst_message_parse_error(msg, &e, &dbg);
        fprintf(stderr, "gstreamer error: %s (%s)\n", e->message, dbg ? dbg : "");
        g_error_free(e); g_free(dbg);
        exit_code = 3; g_main_loop_quit(loop);
    } else if (GST_MESSAGE_TYPE(msg) == GST_MESSAGE_EOS) g_main_loop_quit(loop);
    return TRUE;
}
/* ---- --appsrc: eight pictures of a texture panning by (+3, -2) per picture, pushed forwards and backwards */
static struct { GstElement *src; int n, w, h, fps, npic; guint8 *mem; gsize fsz; const char *clip; } feed;
static gpointer feeder(gpointer u) {
    (void)u;
    for (int i = 0; i < feed.n; i++) {
        const int np = feed.npic, k = np > 1 ? i % (2 * np - 2) : 0;  /* forwards and backwards through the pictures (0 1 .. 7 6 .. 1 0 1 ..), as bench.py walks its clip: */
        guint8 *d = feed.mem + (gsize)(k < np ? k : 2 * np - 2 - k) * feed.fsz; /* wrapping 7 -> 0 would be a jump of (24, 16) samples every eighth picture -- a scene cut for the encoder, an IDR picture shortly after */
        GstBuffer *b = gst_buffer_new_wrapped_full(GST_MEMORY_FLAG_READONLY, d, feed.fsz, 0, feed.fsz, NULL, NULL);
        GST_BUFFER_PTS(b) = gst_util_uint64_scale(GST_SECOND, (guint64)i, (guint64)feed.fps);
        GST_BUFFER_DURATION(b) = gst_util_uint64_scale(GST_SECOND, 1, (guint64)feed.fps);
        if (gst_app_src_push_buffer(GST_APP_SRC(feed.src), b) != GST_FLOW_OK) break;
        /* own flow control (appsrc's block=true stalls with this GStreamer 1.14 when fed from a thread of ours): at most 8 pictures between
         * the source and the sink, which keeps the encoder's pipeline (three in flight) and the queue in front of it full */
        while (i + 1 - (int)g_atomic_int_get(&n_out) > 8 && !exit_code) g_usleep(20);
    }
    gst_app_src_end_of_stream(GST_APP_SRC(feed.src));
    return NULL;
}
static void render_pictures(void) {
    const int w = feed.w, h = feed.h;
    for (int f = 0; f < 8; f++) {
        guint8 *y = feed.mem + (gsize)f * feed.fsz, *uv = y + (gsize)w * h;
        for (int r = 0; r < h; r++)
            for (int c = 0; c < w; c++) {
                const unsigned tx = (unsigned)(c + 3 * f), ty = (unsigned)(r + 2 * (8 - f));
                unsigned v = 128u + (((tx * 7u) ^ (ty * 13u)) & 31u) + (((tx >> 5) + (ty >> 5)) & 1u) * 40u + ((tx * tx + ty * 31u) >> 7 & 15u);
                y[(gsize)r * w + c] = (guint8)(v > 235u ? 235u : v);
            }
        for (int r = 0; r < h / 2; r++)
            for (int c = 0; c < w; c++) uv[(gsize)r * w + c] = (guint8)(128 + ((((c >> 1) + 3 * f / 2) >> 4) + (r >> 4)) % 9 - 4);
    }
}
static int cmpf(const void *a, const void *b) { float x = *(const float *)a, y = *(const float *)b; return (x > y) - (x < y); }
static void pct(const char *name, float *v, unsigned n) {
    if (!n) { printf(",\"%s\":null", name); return; }
    qsort(v, n, sizeof *v, cmpf);
    printf(",\"%s\":{\"p50\":%.3f,\"p95\":%.3f,\"max\":%.3f,\"n\":%u}", name, v[n / 2], v[(unsigned)(n * 0.95)], v[n - 1], n);
}
int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s \"PIPELINE\"\n", argv[0]); return 2; }
    gst_init(&argc, &argv);
    GError *err = NULL;
    GstElement *pipe = gst_parse_launch(argv[1], &err);
    if (!pipe) { fprintf(stderr, "parse error: %s\n", err ? err->message : "?"); return 2; }
    GstElement *enc = gst_bin_get_by_name(GST_BIN(pipe), "venc_bps"), *sink = gst_bin_get_by_name(GST_BIN(pipe), "appsink");
    int no_enc = 0, use_appsrc = 0, pinned = 0, ai = 0;
    for (int i = 2; i < argc; i++)
        if (!strcmp(argv[i], "--props")) {
            if (!enc) enc = gst_bin_get_by_name(GST_BIN(pipe), "venc_kbps");
            if (!enc) { fprintf(stderr, "--props needs an element named venc_bps or venc_kbps\n"); return 2; }
            gint preset = 0, aq = 0, iip = 0, slices = 0, islices = 0;
            gboolean dct = FALSE, i8 = FALSE, sdb = FALSE;
            g_object_get(enc, "speed-preset", &preset, "dct8x8", &dct, "i8x8", &i8, "aq-mode", &aq, "intra-in-p", &iip, "slices", &slices, "slice-deblock", &sdb,
                         "intra-slices", &islices, NULL);
            printf("{\"speed_preset\":%d,\"dct8x8\":%d,\"i8x8\":%d,\"aq_mode\":%d,\"intra_in_p\":%d,\"slices\":%d,\"slice_deblock\":%d,\"intra_slices\":%d,\"has_partitions_property\":%d}\n", preset, dct, i8, aq,
                   iip, slices, sdb, islices, g_object_class_find_property(G_OBJECT_GET_CLASS(enc), "partitions") != NULL);
            return 0;
        }
    for (int i = 2; i < argc; i++) {
        if (!strcmp(argv[i], "--no-encoder")) no_enc = 1;
        else if (!strcmp(argv[i], "--appsrc") && i + 3 < argc) { use_appsrc = 1; ai = i; i += 3; }
        else if (!strcmp(argv[i], "pinned")) pinned = 1;
        else if (!strcmp(argv[i], "--clip") && i + 1 < argc) feed.clip = argv[++i];
    }
    if (use_appsrc) {
        feed.src = gst_bin_get_by_name(GST_BIN(pipe), "src");
        feed.n = atoi(argv[ai + 1]); feed.w = atoi(argv[ai + 2]) & ~3; feed.h = atoi(argv[ai + 3]) & ~1; feed.fps = 60;
        if (!feed.src || feed.n < 1 || feed.w < 16 || feed.h < 16) { fprintf(stderr, "--appsrc needs `appsrc name=src` in the description and N W H\n"); return 2; }
        feed.fsz = (gsize)feed.w * feed.h * 3 / 2;
        feed.npic = 8;
        FILE *cf = feed.clip ? fopen(feed.clip, "rb") : NULL;
        if (feed.clip && !cf) { fprintf(stderr, "cannot open %s\n", feed.clip); return 2; }
        if (cf) { fseek(cf, 0, SEEK_END); const long sz = ftell(cf); fseek(cf, 0, SEEK_SET); feed.npic = (int)(sz / (long)feed.fsz); if (feed.npic > 32) feed.npic = 32; if (feed.npic < 1) { fprintf(stderr, "%s holds no %dx%d NV12 picture\n", feed.clip, feed.w, feed.h); return 2; } }
        if (pinned) { /* libmi355enc.so sits beside this program; loaded here rather than linked (this image's GStreamer brings an older libstdc++ than HIP's) */
            char exe[4096];
            ssize_t n = readlink("/proc/self/exe", exe, sizeof exe - 32);
            if (n > 0) {
                exe[n] = 0;
                char lib[4200];
                snprintf(lib, sizeof lib, "%s/libmi355enc.so", dirname(exe));
                void *dl = dlopen(lib, RTLD_NOW | RTLD_GLOBAL);
                void *(*alloc)(size_t) = dl ? (void *(*)(size_t))dlsym(dl, "mi355enc_host_alloc") : NULL;
                if (alloc) feed.mem = (guint8 *)alloc((gsize)feed.npic * feed.fsz);
                else fprintf(stderr, "cannot load %s: %s\n", lib, dlerror());
            }
        }
        if (pinned && !feed.mem) fprintf(stderr, "mi355enc_host_alloc unavailable: pageable pictures instead\n");
        if (!feed.mem) feed.mem = (guint8 *)g_malloc((gsize)feed.npic * feed.fsz);
        if (cf) { if (fread(feed.mem, feed.fsz, (size_t)feed.npic, cf) != (size_t)feed.npic) { fprintf(stderr, "short read from %s\n", feed.clip); return 2; } fclose(cf); }
        else render_pictures();
        g_object_set(feed.src, "format", GST_FORMAT_TIME, NULL);
    }
    if ((!enc && !no_enc) || !sink) { fprintf(stderr, "the description needs elements named venc_bps and appsink\n"); return 2; }
    if (enc) {
        GstPad *sp = gst_element_get_static_pad(enc, "sink");
        gst_pad_add_probe(sp, GST_PAD_PROBE_TYPE_BUFFER, on_enc_sink, NULL, NULL);
        gst_object_unref(sp);
    }
    rx = socket(AF_INET, SOCK_DGRAM, 0); tx = socket(AF_INET, SOCK_DGRAM, 0);
    struct sockaddr_in a; memset(&a, 0, sizeof a);
    a.sin_family = AF_INET; a.sin_addr.s_addr = htonl(INADDR_LOOPBACK);
    socklen_t al = sizeof a;
    if (rx < 0 || tx < 0 || bind(rx, (struct sockaddr *)&a, sizeof a) || getsockname(rx, (struct sockaddr *)&a, &al)) tx = -1;
    dst = a; /* the receive buffer is never read: datagrams are dropped there, the sender does not care */
    GstAppSinkCallbacks cb = {NULL, NULL, on_sample, {0}};
    gst_app_sink_set_callbacks(GST_APP_SINK(sink), &cb, NULL, NULL);
    loop = g_main_loop_new(NULL, FALSE);
    GstBus *bus = gst_element_get_bus(pipe);
    gst_bus_add_watch(bus, on_bus, NULL);
    const gint64 t_start = g_get_monotonic_time();
    gst_element_set_state(pipe, GST_STATE_PLAYING);
    GThread *ft = use_appsrc ? g_thread_new("feeder", feeder, NULL) : NULL;
    g_main_loop_run(loop);
    if (ft && exit_code == 0) g_thread_join(ft);
    const double secs = (g_get_monotonic_time() - t_start) / 1e6;
    gst_element_set_state(pipe, GST_STATE_NULL);
    printf("{\"samples\":%" G_GUINT64_FORMAT ",\"bytes\":%" G_GUINT64_FORMAT ",\"seconds\":%.3f,\"datagrams_1316\":%" G_GUINT64_FORMAT, n_samples, n_bytes, secs, n_dgrams);
    {
        const guint64 n = n_samples < MAXN ? n_samples : MAXN;
        if (n > 120 && t_arr[n - 1] > t_arr[60]) printf(",\"fps_after_first_gop\":%.1f,\"buffers_timed\":%" G_GUINT64_FORMAT, (double)(n - 1 - 60) * 1e6 / (double)(t_arr[n - 1] - t_arr[60]), n - 1 - 60);
        else printf(",\"fps_after_first_gop\":null");
    }
    printf(",\"us_sink_callback_per_sample\":%.1f", n_samples ? (double)us_in_callback / (double)n_samples : 0.0);
    const unsigned skip = n_lat > 90 ? 60 : 0; /* discard the first GOP (warm-up) */
    pct("ms_encoder_sink_to_appsink", l_enc + skip, n_lat - skip);
    pct("ms_appsink_to_last_udp_send", l_send + skip, n_lat - skip);
    printf("}\n");
    if (tx >= 0) close(tx);
    if (rx >= 0) close(rx);
    return exit_code;
}
