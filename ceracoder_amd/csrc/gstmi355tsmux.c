/*
 * gstmi355tsmux.c -- GStreamer element `mi355tsmux`: video-only stand-in for the
 * `h264parse config-interval=-1 ! queue ! mpegtsmux name=mux` tail of ceracoder's pipeline files
 * (/root/reference/pipeline/generic/x264_superfast_camlink:6,10), for systems without
 * gst-plugins-bad.  One sink pad (Annex-B access units with SPS/PPS
 * in band, which is what mi355h264enc emits), one src pad (188-byte transport packets, the
 * unit /root/reference/src/ceracoder.c:297-339 regroups into 1316-byte SRT payloads).
 * The muxing itself is the plain-C library behind include/mi355ts.h.
 */
#include <gst/gst.h>
#include <string.h>

#include "../../include/mi355ts.h"

typedef struct {
    GstElement parent;
    GstPad *sinkpad, *srcpad;
    mi355ts_t *mux;
    gboolean sent_caps;
} GstMi355TsMux;
typedef struct { GstElementClass parent_class; } GstMi355TsMuxClass;

GType gst_mi355tsmux_get_type(void);
#define GST_TYPE_MI355TSMUX (gst_mi355tsmux_get_type())
#define GST_MI355TSMUX(o) (G_TYPE_CHECK_INSTANCE_CAST((o), GST_TYPE_MI355TSMUX, GstMi355TsMux))
G_DEFINE_TYPE(GstMi355TsMux, gst_mi355tsmux, GST_TYPE_ELEMENT)

static GstStaticPadTemplate ts_sink_tmpl = GST_STATIC_PAD_TEMPLATE("sink", GST_PAD_SINK, GST_PAD_ALWAYS,
    GST_STATIC_CAPS("video/x-h264, stream-format=(string)byte-stream, alignment=(string)au"));
static GstStaticPadTemplate ts_src_tmpl = GST_STATIC_PAD_TEMPLATE("src", GST_PAD_SRC, GST_PAD_ALWAYS,
    GST_STATIC_CAPS("video/mpegts, systemstream=(boolean)true, packetsize=(int)188"));

static GstFlowReturn ts_chain(GstPad *pad, GstObject *parent, GstBuffer *buf) {
    GstMi355TsMux *s = GST_MI355TSMUX(parent);
    (void)pad;
    GstMapInfo in;
    if (!gst_buffer_map(buf, &in, GST_MAP_READ)) { gst_buffer_unref(buf); return GST_FLOW_ERROR; }
    const gsize cap = mi355ts_bound(in.size);
    GstBuffer *out = gst_buffer_new_allocate(NULL, cap, NULL);
    GstMapInfo om;
    if (!out || !gst_buffer_map(out, &om, GST_MAP_WRITE)) {
        gst_buffer_unmap(buf, &in); gst_buffer_unref(buf);
        if (out) gst_buffer_unref(out);
        return GST_FLOW_ERROR;
    }
    const GstClockTime pts = GST_BUFFER_PTS_IS_VALID(buf) ? GST_BUFFER_PTS(buf) : 0;
    const int key = !GST_BUFFER_FLAG_IS_SET(buf, GST_BUFFER_FLAG_DELTA_UNIT);
    size_t n = 0;
    const int rc = mi355ts_mux(s->mux, in.data, in.size, (int64_t)pts, key, om.data, om.size, &n);
    gst_buffer_unmap(out, &om);
    gst_buffer_unmap(buf, &in);
    if (rc) {
        GST_ELEMENT_ERROR(s, STREAM, MUX, ("mi355tsmux: cannot mux access unit (rc=%d)", rc), (NULL));
        gst_buffer_unref(buf); gst_buffer_unref(out);
        return GST_FLOW_ERROR;
    }
    gst_buffer_set_size(out, (gssize)n);
    GST_BUFFER_PTS(out) = GST_BUFFER_PTS(buf); GST_BUFFER_DTS(out) = GST_BUFFER_DTS(buf); GST_BUFFER_DURATION(out) = GST_BUFFER_DURATION(buf);
    if (!key) GST_BUFFER_FLAG_SET(out, GST_BUFFER_FLAG_DELTA_UNIT);
    gst_buffer_unref(buf);
    return gst_pad_push(s->srcpad, out);
}

static gboolean ts_sink_event(GstPad *pad, GstObject *parent, GstEvent *ev) {
    GstMi355TsMux *s = GST_MI355TSMUX(parent);
    if (GST_EVENT_TYPE(ev) == GST_EVENT_CAPS) { /* our output format does not depend on the input caps */
        GstCaps *c = gst_static_pad_template_get_caps(&ts_src_tmpl);
        gboolean ok = gst_pad_set_caps(s->srcpad, c);
        gst_caps_unref(c);
        gst_event_unref(ev);
        return ok;
    }
    return gst_pad_event_default(pad, parent, ev);
}

static GstStateChangeReturn ts_change_state(GstElement *e, GstStateChange t) {
    GstMi355TsMux *s = GST_MI355TSMUX(e);
    if (t == GST_STATE_CHANGE_READY_TO_PAUSED) {
        if (s->mux) mi355ts_close(s->mux);
        s->mux = mi355ts_open(); /* fresh continuity counters and PSI schedule per run */
        if (!s->mux) return GST_STATE_CHANGE_FAILURE;
    }
    GstStateChangeReturn r = GST_ELEMENT_CLASS(gst_mi355tsmux_parent_class)->change_state(e, t);
    if (t == GST_STATE_CHANGE_PAUSED_TO_READY && s->mux) { mi355ts_close(s->mux); s->mux = NULL; }
    return r;
}

static void ts_finalize(GObject *o) {
    GstMi355TsMux *s = GST_MI355TSMUX(o);
    if (s->mux) mi355ts_close(s->mux);
    G_OBJECT_CLASS(gst_mi355tsmux_parent_class)->finalize(o);
}

static void gst_mi355tsmux_class_init(GstMi355TsMuxClass *k) {
    GstElementClass *e = GST_ELEMENT_CLASS(k);
    G_OBJECT_CLASS(k)->finalize = ts_finalize;
    e->change_state = ts_change_state;
    gst_element_class_add_static_pad_template(e, &ts_sink_tmpl);
    gst_element_class_add_static_pad_template(e, &ts_src_tmpl);
    gst_element_class_set_static_metadata(e, "Minimal MPEG-TS muxer (H.264 video only)", "Codec/Muxer",
        "Wraps H.264 access units into 188-byte MPEG-2 transport packets (PAT/PMT/PCR/PES) without gst-plugins-bad", "ceracoder-amd");
}
static void gst_mi355tsmux_init(GstMi355TsMux *s) {
    s->mux = NULL; s->sent_caps = FALSE;
    s->sinkpad = gst_pad_new_from_static_template(&ts_sink_tmpl, "sink");
    gst_pad_set_chain_function(s->sinkpad, ts_chain);
    gst_pad_set_event_function(s->sinkpad, ts_sink_event);
    gst_element_add_pad(GST_ELEMENT(s), s->sinkpad);
    s->srcpad = gst_pad_new_from_static_template(&ts_src_tmpl, "src");
    gst_pad_use_fixed_caps(s->srcpad);
    gst_element_add_pad(GST_ELEMENT(s), s->srcpad);
}
