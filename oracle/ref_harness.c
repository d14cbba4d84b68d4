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
 *
 * usage: ref_harness PIPELINE_FILE OUT_FILE [SCRIPT_FILE]     (script lines: "<ms> <bps>")
 */
#include <gst/app/gstappsink.h>
#include <gst/gst.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

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

static GstFlowReturn new_buf_cb(GstAppSink *sink, gpointer user) {
    (void)user;
    GstSample *sample = gst_app_sink_pull_sample(sink);
    if (!sample) return GST_FLOW_OK;
    GstBuffer *buf = gst_sample_get_buffer(sample);
    GstMapInfo map;
    if (gst_buffer_map(buf, &map, GST_MAP_READ)) {
        guint32 len = (guint32)map.size;
        guint64 pts = GST_BUFFER_PTS(buf);
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
    if (have_enc) encoder_control_set_bitrate(&enc, first); /* state NULL, as ceracoder.c:515-518 */
    guint bps_prop = 0;
    if (have_enc) g_object_get(G_OBJECT(enc.element), "bps", &bps_prop, NULL);
    fprintf(stderr, "{\"encoder_found\":%d,\"bitrate_div\":%d,\"bps_after_null_state_write\":%u}\n", have_enc, enc.bitrate_div, bps_prop);
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
    printf("{\"samples\":%" G_GUINT64_FORMAT ",\"bytes\":%" G_GUINT64_FORMAT ",\"seconds\":%.3f,\"setpoints_applied\":%d}\n", n_samples, n_bytes, secs, script_i);
    pipeline_file_unload(&pf);
    return exit_code;
}
