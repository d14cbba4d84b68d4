/*
 * gstmi355h264enc.c -- GStreamer element `mi355h264enc`: the drop-in for the `x264enc`
 * token of ceracoder's pipeline files
 * (/root/reference/pipeline/generic/x264_superfast_camlink:5:
 *      `x264enc speed-preset=2 key-int-max=60 name=venc_kbps`).
 *
 * Contract taken from the reference:
 *  - found by NAME, not type: `venc_bps` (value in bit/s) is tried first, then `venc_kbps`
 *    (value / 1000)                       /root/reference/src/gst/encoder_control.c:29-32
 *  - the property the binary writes is "bps"        .../encoder_control.c:53
 *    (README/docs call it "bitrate": /root/reference/README.md:256) -> both exist here and
 *    alias one target; "bitrate" is kbit/s with x264enc's default 2048
 *  - every x264 pipeline file of the reference says `name=venc_kbps`
 *    (pipeline/generic/x264_*:5-6, generic-builder.ts:54), for which encoder_control.c:53
 *    writes bitrate / 1000 into "bps": an element NAMED venc_kbps therefore reads and writes
 *    "bps" in kbit/s, so that such a file works with only the factory token changed
 *  - first write happens in state NULL, before PLAYING /root/reference/src/ceracoder.c:515-518
 *  - later writes come from the GLib main thread every <= 20 ms while the streaming thread
 *    encodes                                         /root/reference/src/ceracoder.c:266-295
 *  - upstream zeroes DTS and smooths PTS             /root/reference/src/ceracoder.c:371-423
 *    -> output DTS = PTS, no reordering
 *  - errors go to the bus; the app exits on ERROR    /root/reference/src/ceracoder.c:425-438
 * API level: GStreamer 1.14 (GstVideoEncoder).  Plain C; the device is reached only
 * through the C ABI in include/mi355enc.h.  There is no CPU path: without a HIP device the
 * element posts an ERROR when it leaves READY.
 */
#include <gst/gst.h>
#include <gst/video/gstvideoencoder.h>
#include <gst/video/video.h>
#include <string.h>

#include "../../include/mi355enc.h"

#define PACKAGE "ceracoder-amd"
#define VERSION "0.1.0"

GST_DEBUG_CATEGORY_STATIC(mi355_debug);
#define GST_CAT_DEFAULT mi355_debug

typedef struct {
    GstVideoEncoder parent;
    /* properties (guarded by the object lock; bitrate is additionally forwarded atomically) */
    guint rate_raw;     /* the target as it was written: through "bps" in the unit the element's NAME implies, through "bitrate" in kbit/s */
    gboolean rate_is_bps; /* the last write came through "bps" */
    guint key_int_max;
    gint device_id, me_range, qp, pipeline_depth, speed_preset;
    gint open_depth;  /* pipeline-depth the encoder was opened with: a write to the property in mid-stream takes effect at the next (re)negotiation */
    gboolean stats;
    gint threads;
    gboolean scenecut, exclusive_gpu;
    guint vbv_ms;
    gboolean pinned_input;
    /* coding tools: -1 = not set on the element, i.e. what speed-preset selects (preset_tools); an explicit write wins */
    gint dct8x8, i8x8, aq_mode, intra_in_p, slices, slice_deblock;
    gint intra_slices;
    gboolean single_stream;
    /* streaming state */
    mi355enc_t *enc;
    GstVideoCodecState *input_state;
    gsize max_au;
    guint8 *au_buf;     /* access units are coded here (worst-case size, allocated once per format) and copied into right-sized buffers */
    GstClockTime last_pts;
    /* where the streaming thread's time goes, per stage (microseconds, summed; printed with stats=true): what separates the element's rate from the C ABI's */
    gint64 us_map, us_submit, us_collect, us_output, us_push, us_frames;
} GstMi355H264Enc;
typedef struct { GstVideoEncoderClass parent_class; } GstMi355H264EncClass;

#define GST_TYPE_MI355H264ENC (gst_mi355h264enc_get_type())
#define GST_MI355H264ENC(o) (G_TYPE_CHECK_INSTANCE_CAST((o), GST_TYPE_MI355H264ENC, GstMi355H264Enc))
G_DEFINE_TYPE(GstMi355H264Enc, gst_mi355h264enc, GST_TYPE_VIDEO_ENCODER)

enum { PROP_0, PROP_BPS, PROP_BITRATE, PROP_KEY_INT_MAX, PROP_DEVICE_ID, PROP_ME_RANGE, PROP_QP, PROP_PIPELINE_DEPTH,
       PROP_SPEED_PRESET, PROP_STATS, PROP_DCT8X8, PROP_THREADS, PROP_SCENECUT, PROP_VBV, PROP_INTRA_IN_P, PROP_EXCLUSIVE, PROP_PINNED_INPUT, PROP_AQ_MODE, PROP_SINGLE_STREAM, PROP_INTRA_SLICES, PROP_I8X8, PROP_SLICES, PROP_SLICE_DEBLOCK };

static GstStaticPadTemplate sink_tmpl = GST_STATIC_PAD_TEMPLATE("sink", GST_PAD_SINK, GST_PAD_ALWAYS,
    GST_STATIC_CAPS("video/x-raw, format=(string){ NV12, I420, YUY2, UYVY }, width=(int)[16,8192], height=(int)[16,8192], framerate=(fraction)[0/1,MAX]"));
static GstStaticPadTemplate src_tmpl = GST_STATIC_PAD_TEMPLATE("src", GST_PAD_SRC, GST_PAD_ALWAYS,
    GST_STATIC_CAPS("video/x-h264, stream-format=(string)byte-stream, alignment=(string)au, profile=(string){ constrained-baseline, high }, "
                    "width=(int)[16,8192], height=(int)[16,8192], framerate=(fraction)[0/1,MAX]"));

/* x264enc's speed-preset enum, accepted so that an x264enc line converts by changing only the factory name */
static GType speed_preset_type(void) {
    static GType t = 0;
    static const GEnumValue v[] = {{0, "No preset", "None"}, {1, "ultrafast", "ultrafast"}, {2, "superfast", "superfast"},
                                   {3, "veryfast", "veryfast"}, {4, "faster", "faster"}, {5, "fast", "fast"}, {6, "medium", "medium"},
                                   {7, "slow", "slow"}, {8, "slower", "slower"}, {9, "veryslow", "veryslow"}, {10, "placebo", "placebo"},
                                   {0, NULL, NULL}};
    if (!t) t = g_enum_register_static("GstMi355H264EncPreset", v);
    return t;
}

/* What x264enc's speed-preset selects, restated in the tools this encoder has.  The reference's pipeline files pass speed-preset=2 (superfast) and =3
 * (veryfast) (/root/reference/pipeline/generic/x264_superfast_camlink:5, x264_veryfast_camlink:5; generator bindings/typescript/src/pipeline/generic-builder.ts:43-55),
 * so the one-token swap has to land on a toolset and not on "ignored".  x264's own table: ultrafast = no 8x8 transform, no adaptive quantisation, no partitions;
 * superfast and slower keep the 8x8 transform with Intra_8x8 and aq-mode 1 (partitions i8x8,i4x4); veryfast and slower analyse inter partitions and
 * Intra_4x4 in P pictures.  Here: 0 (none) / 1: Constrained Baseline, one QP per picture -- the library's defaults; 2 (superfast) and above: dct8x8 + i8x8 +
 * aq-mode 1 (High profile).  What the slower presets would add -- Intra_4x4 in P pictures (intra-in-p=2), inter partitions (4x4 transform only) -- measured
 * rate-distortion neutral here and, with adaptive quantisation, Intra_4x4 in P puts a picture's stages in stream order: they stay explicit
 * switches.  Slices follow the library's default (mi355enc_default_cfg) unless `slices` / `slice-deblock` are set. */
typedef struct { gint dct8x8, i8x8, aq_mode, partitions, intra_in_p, slices, slice_deblock; } toolset_t;
static void preset_tools(gint preset, toolset_t *t) {
    t->dct8x8 = t->i8x8 = t->aq_mode = preset >= 2; t->partitions = 0; /* (inter partitions: no property -- the search the fused stage can afford is rate-distortion neutral; cfg.partitions remains in the C ABI) */ t->intra_in_p = 1;
    t->slices = -1; t->slice_deblock = -1; /* the library's defaults (mi355enc_default_cfg) */
}
/* the element's effective tools: explicit properties over the preset's (object lock held) */
static void effective_tools(GstMi355H264Enc *s, toolset_t *t) {
    preset_tools(s->speed_preset, t);
    if (s->dct8x8 >= 0) t->dct8x8 = s->dct8x8;
    if (s->i8x8 >= 0) t->i8x8 = s->i8x8;
    if (s->aq_mode >= 0) t->aq_mode = s->aq_mode;
    if (s->intra_in_p >= 0) t->intra_in_p = s->intra_in_p;
    if (s->slices >= 0) t->slices = s->slices;
    if (s->slice_deblock >= 0) t->slice_deblock = s->slice_deblock;
    if (!t->dct8x8) t->i8x8 = 0; /* Intra_8x8 needs the 8x8 transform (High profile) */
}

/* The unit of "bps" follows the reference's naming rule (encoder_control.c:29-32,53): an element named venc_kbps is sent
 * bitrate / 1000.  Called with the object lock held (the name is guarded by it). */
static guint bps_unit(GstMi355H264Enc *s) {
    const gchar *n = GST_OBJECT_NAME(s);
    return (n && strcmp(n, "venc_kbps") == 0) ? 1000u : 1u;
}
static guint clamp_bps(guint64 v) { return v < 1000 ? 1000u : v > 1000000000u ? 1000000000u : (guint)v; }
/* The target in bit/s, resolved where it is consumed (object lock held): `mi355h264enc bps=6000 name=venc_kbps` sets the property before
 * the element has its name, and g_object_set followed by gst_object_set_name does the same -- a unit applied at write time would be wrong. */
static guint target_bps(GstMi355H264Enc *s) { return clamp_bps((guint64)s->rate_raw * (s->rate_is_bps ? bps_unit(s) : 1000u)); }

static void set_property(GObject *obj, guint id, const GValue *val, GParamSpec *ps) {
    GstMi355H264Enc *s = GST_MI355H264ENC(obj);
    GST_OBJECT_LOCK(s);
    switch (id) {
    case PROP_BPS: s->rate_raw = g_value_get_uint(val); s->rate_is_bps = TRUE; if (s->enc) mi355enc_set_bitrate(s->enc, target_bps(s)); break;
    case PROP_BITRATE: s->rate_raw = g_value_get_uint(val); s->rate_is_bps = FALSE; if (s->enc) mi355enc_set_bitrate(s->enc, target_bps(s)); break;
    case PROP_KEY_INT_MAX: s->key_int_max = g_value_get_uint(val); break;
    case PROP_DEVICE_ID: s->device_id = g_value_get_int(val); break;
    case PROP_ME_RANGE: s->me_range = g_value_get_int(val); break;
    case PROP_QP: s->qp = g_value_get_int(val); if (s->enc) mi355enc_set_fixed_qp(s->enc, s->qp); break;
    case PROP_PIPELINE_DEPTH: s->pipeline_depth = g_value_get_int(val); break;
    case PROP_SPEED_PRESET: s->speed_preset = g_value_get_enum(val); break;
    case PROP_STATS: s->stats = g_value_get_boolean(val); break;
    case PROP_DCT8X8: s->dct8x8 = g_value_get_boolean(val) ? 1 : 0; break;
    case PROP_THREADS: s->threads = g_value_get_int(val); break;
    case PROP_SCENECUT: s->scenecut = g_value_get_boolean(val); break;
    case PROP_EXCLUSIVE: s->exclusive_gpu = g_value_get_boolean(val); break;
    case PROP_VBV: s->vbv_ms = g_value_get_uint(val); break;
    case PROP_INTRA_IN_P: s->intra_in_p = g_value_get_int(val); break;
    case PROP_PINNED_INPUT: s->pinned_input = g_value_get_boolean(val); break;
    case PROP_AQ_MODE: s->aq_mode = g_value_get_int(val); break;
    case PROP_INTRA_SLICES: s->intra_slices = g_value_get_int(val); break;
    case PROP_SLICES: s->slices = g_value_get_int(val); break;
    case PROP_SLICE_DEBLOCK: s->slice_deblock = g_value_get_boolean(val) ? 1 : 0; break;
    case PROP_I8X8: s->i8x8 = g_value_get_boolean(val) ? 1 : 0; break;
    case PROP_SINGLE_STREAM: s->single_stream = g_value_get_boolean(val); break;
    default: G_OBJECT_WARN_INVALID_PROPERTY_ID(obj, id, ps); break;
    }
    GST_OBJECT_UNLOCK(s);
}
static void get_property(GObject *obj, guint id, GValue *val, GParamSpec *ps) {
    GstMi355H264Enc *s = GST_MI355H264ENC(obj);
    toolset_t t;
    GST_OBJECT_LOCK(s);
    effective_tools(s, &t); /* a tool property reads as what the encoder will use: the explicit value, or the preset's */
    switch (id) {
    case PROP_BPS: g_value_set_uint(val, s->rate_is_bps ? s->rate_raw : target_bps(s) / bps_unit(s)); break;
    case PROP_BITRATE: g_value_set_uint(val, target_bps(s) / 1000u); break;
    case PROP_KEY_INT_MAX: g_value_set_uint(val, s->key_int_max); break;
    case PROP_DEVICE_ID: g_value_set_int(val, s->device_id); break;
    case PROP_ME_RANGE: g_value_set_int(val, s->me_range); break;
    case PROP_QP: g_value_set_int(val, s->qp); break;
    case PROP_PIPELINE_DEPTH: g_value_set_int(val, s->pipeline_depth); break;
    case PROP_SPEED_PRESET: g_value_set_enum(val, s->speed_preset); break;
    case PROP_STATS: g_value_set_boolean(val, s->stats); break;
    case PROP_DCT8X8: g_value_set_boolean(val, t.dct8x8 != 0); break;
    case PROP_THREADS: g_value_set_int(val, s->threads); break;
    case PROP_SCENECUT: g_value_set_boolean(val, s->scenecut); break;
    case PROP_EXCLUSIVE: g_value_set_boolean(val, s->exclusive_gpu); break;
    case PROP_VBV: g_value_set_uint(val, s->vbv_ms); break;
    case PROP_INTRA_IN_P: g_value_set_int(val, t.intra_in_p); break;
    case PROP_PINNED_INPUT: g_value_set_boolean(val, s->pinned_input); break;
    case PROP_AQ_MODE: g_value_set_int(val, t.aq_mode); break;
    case PROP_INTRA_SLICES: g_value_set_int(val, s->intra_slices); break;
    case PROP_SLICES: g_value_set_int(val, t.slices < 0 ? 0 : t.slices); break;
    case PROP_SLICE_DEBLOCK: g_value_set_boolean(val, t.slice_deblock != 0); break; /* (unset: the library's default, on) */
    case PROP_I8X8: g_value_set_boolean(val, t.i8x8 != 0); break;
    case PROP_SINGLE_STREAM: g_value_set_boolean(val, s->single_stream); break;
    default: G_OBJECT_WARN_INVALID_PROPERTY_ID(obj, id, ps); break;
    }
    GST_OBJECT_UNLOCK(s);
}

static void close_encoder(GstMi355H264Enc *s) {
    mi355enc_t *e;
    GST_OBJECT_LOCK(s);
    e = s->enc; s->enc = NULL;
    GST_OBJECT_UNLOCK(s);
    if (e) {
        if (s->stats) {
            mi355enc_stats_t st;
            if (mi355enc_get_stats(e, &st) == 0)
                g_printerr("{\"element\":\"mi355h264enc\",\"frames\":%" G_GUINT64_FORMAT ",\"idr\":%" G_GUINT64_FORMAT ",\"bytes\":%" G_GUINT64_FORMAT
                           ",\"ms_entropy\":%.3f,\"ms_wait\":%.3f,\"last_qp\":%u,\"open_ms\":%.1f,\"target_bps\":%u"
                           ",\"streaming_thread_us_per_frame\":{\"map_input\":%.1f,\"submit\":%.1f,\"collect\":%.1f,\"output_buffer\":%.1f,\"push_downstream\":%.1f}}\n", st.frames, st.idr_frames, st.bytes,
                           st.ms_entropy, st.ms_wait, st.last_qp, st.ms_open, st.target_bps, (double)s->us_map / (double)(s->us_frames ? s->us_frames : 1),
                           (double)s->us_submit / (double)(s->us_frames ? s->us_frames : 1), (double)s->us_collect / (double)(s->us_frames ? s->us_frames : 1),
                           (double)s->us_output / (double)(s->us_frames ? s->us_frames : 1), (double)s->us_push / (double)(s->us_frames ? s->us_frames : 1));
        }
        mi355enc_close(e);
    }
}
static gboolean enc_stop(GstVideoEncoder *ve) {
    GstMi355H264Enc *s = GST_MI355H264ENC(ve);
    close_encoder(s);
    if (s->input_state) { gst_video_codec_state_unref(s->input_state); s->input_state = NULL; }
    g_free(s->au_buf); s->au_buf = NULL;
    return TRUE;
}
static GstFlowReturn drain(GstMi355H264Enc *s, gboolean push);
static gboolean enc_start(GstVideoEncoder *ve) {
    (void)ve;
    return TRUE; /* geometry is unknown until set_format; the device is opened there, still before data flows */
}

static gboolean enc_set_format(GstVideoEncoder *ve, GstVideoCodecState *state) {
    GstMi355H264Enc *s = GST_MI355H264ENC(ve);
    GstVideoInfo *vi = &state->info;
    mi355enc_cfg_t cfg;
    mi355enc_t *e = NULL;
    int fn = GST_VIDEO_INFO_FPS_N(vi), fd = GST_VIDEO_INFO_FPS_D(vi);
    if (fn <= 0 || fd <= 0) { fn = 30; fd = 1; } /* variable framerate: rate control assumes 30 */
    if (s->enc) drain(s, TRUE); /* renegotiation in mid-stream: the picture still on the device belongs to the old format (x264enc flushes here too) */
    close_encoder(s);
    mi355enc_default_cfg(&cfg, GST_VIDEO_INFO_WIDTH(vi), GST_VIDEO_INFO_HEIGHT(vi), fn, fd);
    GST_OBJECT_LOCK(s);
    cfg.gop = s->key_int_max ? (int)s->key_int_max : 250;
    cfg.me_range = s->me_range; cfg.bitrate_bps = target_bps(s); cfg.device_id = s->device_id; cfg.fixed_qp = s->qp;
    cfg.pipeline_depth = s->pipeline_depth; cfg.cavlc_threads = s->threads > 0 ? s->threads : 0; cfg.scenecut = s->scenecut ? 1 : 0; cfg.exclusive_device = s->exclusive_gpu ? 1 : 0; cfg.vbv_ms = (int)s->vbv_ms;
    cfg.single_stream = s->single_stream ? 1 : 0; cfg.intra_slices = s->intra_slices;
    {
        toolset_t t;
        effective_tools(s, &t);
        cfg.transform8x8 = t.dct8x8; cfg.i8x8 = t.i8x8; cfg.aq_mode = t.aq_mode; cfg.partitions = t.partitions; cfg.intra_in_p = t.intra_in_p;
        if (t.slices >= 0) cfg.slices = t.slices;               /* (-1: mi355enc_default_cfg's) */
        if (t.slice_deblock >= 0) cfg.slice_deblock = t.slice_deblock;
    }
    GST_OBJECT_UNLOCK(s);
    int r = mi355enc_open(&cfg, &e);
    if (r != MI355ENC_OK) {
        if (e) mi355enc_close(e);
        GST_ELEMENT_ERROR(s, LIBRARY, INIT, ("mi355h264enc: cannot open the MI355X encoder: %s", mi355enc_strerror(r)),
                          ("mi355enc_open(%dx%d, device-id=%d) returned %d", cfg.width, cfg.height, cfg.device_id, r));
        return FALSE;
    }
    GST_OBJECT_LOCK(s);
    s->enc = e;
    s->open_depth = cfg.pipeline_depth;
    mi355enc_set_bitrate(e, target_bps(s)); /* a write that raced with open() must not be lost */
    GST_OBJECT_UNLOCK(s);
    s->max_au = mi355enc_max_au_bytes(e);
    g_free(s->au_buf);
    s->au_buf = g_malloc(s->max_au);
    s->last_pts = GST_CLOCK_TIME_NONE;
    {
        mi355enc_stats_t st;
        if (mi355enc_get_stats(e, &st) == 0)
            GST_INFO_OBJECT(s, "opened %dx%d @%d/%d on device %d in %.1f ms, target %u bit/s, gop %d", cfg.width, cfg.height, fn, fd, cfg.device_id, st.ms_open, cfg.bitrate_bps, cfg.gop);
    }
    if (s->input_state) gst_video_codec_state_unref(s->input_state);
    s->input_state = gst_video_codec_state_ref(state);
    GstCaps *caps = gst_caps_new_simple("video/x-h264", "stream-format", G_TYPE_STRING, "byte-stream", "alignment", G_TYPE_STRING, "au",
                                        "profile", G_TYPE_STRING, cfg.transform8x8 ? "high" : "constrained-baseline", NULL);
    GstVideoCodecState *out = gst_video_encoder_set_output_state(ve, caps, state);
    gst_video_codec_state_unref(out);
    if (cfg.pipeline_depth > 0) {
        GstClockTime d = gst_util_uint64_scale(GST_SECOND, (guint64)fd * cfg.pipeline_depth, fn);
        gst_video_encoder_set_latency(ve, d, d);
    }
    return gst_video_encoder_negotiate(ve);
}

/* entropy-code the oldest submitted picture and push it in a buffer of exactly its size (the worst case, max_au, is
 * 12 MB at 1080p: buffers of that capacity would sit in the downstream queues of the reference's pipelines, which hold up
 * to 1000 of them -- pipeline/generic/x264_superfast_camlink:7) */
static GstFlowReturn collect_into(GstMi355H264Enc *s, GstVideoCodecFrame *frame) {
    GstVideoEncoder *ve = GST_VIDEO_ENCODER(s);
    size_t len = 0;
    int key = 0, qp = 0;
    const gint64 t0 = g_get_monotonic_time();
    int r = mi355enc_collect(s->enc, s->au_buf, s->max_au, &len, &key, NULL, &qp);
    const gint64 t1 = g_get_monotonic_time();
    s->us_collect += t1 - t0;
    if (r != MI355ENC_OK) {
        GST_ELEMENT_ERROR(s, STREAM, ENCODE, ("mi355h264enc: encode failed: %s", mi355enc_strerror(r)), ("mi355enc_collect returned %d", r));
        gst_video_encoder_finish_frame(ve, frame);
        return GST_FLOW_ERROR;
    }
    GstFlowReturn fr = gst_video_encoder_allocate_output_frame(ve, frame, len);
    if (fr != GST_FLOW_OK) { gst_video_encoder_finish_frame(ve, frame); return fr; }
    gst_buffer_fill(frame->output_buffer, 0, s->au_buf, len);
    if (key) GST_VIDEO_CODEC_FRAME_SET_SYNC_POINT(frame);
    else GST_VIDEO_CODEC_FRAME_UNSET_SYNC_POINT(frame);
    /* No reordering: DTS = PTS.  Upstream (`identity name=ptsfixup`, ceracoder.c:371-423) rewrites PTS onto a fixed grid and
     * zeroes DTS; whatever arrives, the PTS sequence leaving here never runs backwards (mpegtsmux and SRT receivers reject
     * that), so a picture stamped at or before its predecessor takes the predecessor's stamp. */
    if (GST_CLOCK_TIME_IS_VALID(frame->pts) && GST_CLOCK_TIME_IS_VALID(s->last_pts) && frame->pts < s->last_pts) frame->pts = s->last_pts;
    if (GST_CLOCK_TIME_IS_VALID(frame->pts)) s->last_pts = frame->pts;
    frame->dts = frame->pts;
    GST_LOG_OBJECT(s, "access unit %" G_GSIZE_FORMAT " bytes, %s, qp %d, pts %" GST_TIME_FORMAT, len, key ? "IDR" : "P", qp, GST_TIME_ARGS(frame->pts));
    const gint64 t2 = g_get_monotonic_time();
    s->us_output += t2 - t1;
    fr = gst_video_encoder_finish_frame(ve, frame); /* the base class's bookkeeping and the push into whatever follows (a queue in the reference's pipelines) */
    s->us_push += g_get_monotonic_time() - t2; s->us_frames++;
    return fr;
}

static GstFlowReturn enc_handle_frame(GstVideoEncoder *ve, GstVideoCodecFrame *frame) {
    GstMi355H264Enc *s = GST_MI355H264ENC(ve);
    GstVideoFrame vf;
    if (!s->enc || !s->input_state) { gst_video_encoder_finish_frame(ve, frame); return GST_FLOW_NOT_NEGOTIATED; }
    if (GST_BUFFER_FLAG_IS_SET(frame->input_buffer, GST_BUFFER_FLAG_DROPPABLE)) {
        /* what `identity name=ptsfixup` does to a picture that arrived too early to get a slot on its PTS grid (ceracoder.c:414-419:
         * "dropping an input buffer"): it keeps its raw, non-monotone PTS.  Not coded, no output buffer. */
        GST_DEBUG_OBJECT(s, "input picture flagged DROPPABLE (pts %" GST_TIME_FORMAT "): not coded", GST_TIME_ARGS(frame->pts));
        return gst_video_encoder_finish_frame(ve, frame);
    }
    const gint64 t0 = g_get_monotonic_time();
    if (!gst_video_frame_map(&vf, &s->input_state->info, frame->input_buffer, GST_MAP_READ)) {
        gst_video_encoder_finish_frame(ve, frame);
        return GST_FLOW_ERROR;
    }
    /* NV12 goes in as is; I420 (x264enc's native format) and packed 4:2:2 are converted on the device, so the
     * `videoconvert` in front of the encoder (pipeline/generic/x264_superfast_camlink:4) degenerates to a pass-through */
    int fmt = MI355ENC_FMT_NV12;
    switch (GST_VIDEO_INFO_FORMAT(&s->input_state->info)) {
    case GST_VIDEO_FORMAT_I420: fmt = MI355ENC_FMT_I420; break;
    case GST_VIDEO_FORMAT_YUY2: fmt = MI355ENC_FMT_YUY2; break;
    case GST_VIDEO_FORMAT_UYVY: fmt = MI355ENC_FMT_UYVY; break;
    default: break;
    }
    const uint8_t *planes[3] = {NULL, NULL, NULL};
    int strides[3] = {0, 0, 0};
    for (guint i = 0; i < GST_VIDEO_FRAME_N_PLANES(&vf) && i < 3; i++) {
        planes[i] = GST_VIDEO_FRAME_PLANE_DATA(&vf, i);
        strides[i] = GST_VIDEO_FRAME_PLANE_STRIDE(&vf, i);
    }
    const gint64 t1 = g_get_monotonic_time();
    int r = mi355enc_submit_fmt(s->enc, fmt, planes, strides, (int64_t)frame->pts, GST_VIDEO_CODEC_FRAME_IS_FORCE_KEYFRAME(frame) ? 1 : 0);
    s->us_map += t1 - t0; s->us_submit += g_get_monotonic_time() - t1;
    gst_video_frame_unmap(&vf); /* pageable memory: submit() has copied the picture out; pinned memory of our pool: the transfer is in flight and the
                                   codec frame keeps the buffer until the picture is collected */
    if (r != MI355ENC_OK) {
        GST_ELEMENT_ERROR(s, STREAM, ENCODE, ("mi355h264enc: submit failed: %s", mi355enc_strerror(r)), ("mi355enc_submit returned %d", r));
        gst_video_encoder_finish_frame(ve, frame);
        return GST_FLOW_ERROR;
    }
    GstFlowReturn fr = GST_FLOW_OK;
    if (mi355enc_pending(s->enc) > s->open_depth) {
        GstVideoCodecFrame *old = gst_video_encoder_get_oldest_frame(ve);
        if (old) fr = collect_into(s, old); /* finish_frame() consumes the reference */
    }
    gst_video_codec_frame_unref(frame);
    return fr;
}
static GstFlowReturn drain(GstMi355H264Enc *s, gboolean push) {
    GstVideoEncoder *ve = GST_VIDEO_ENCODER(s);
    GstFlowReturn fr = GST_FLOW_OK;
    while (s->enc && mi355enc_pending(s->enc) > 0) {
        GstVideoCodecFrame *old = gst_video_encoder_get_oldest_frame(ve);
        if (!old) break;
        if (push) fr = collect_into(s, old);
        else {
            size_t n = 0;
            mi355enc_collect(s->enc, s->au_buf, s->max_au, &n, NULL, NULL, NULL);
            gst_video_encoder_finish_frame(ve, old); /* no output buffer: dropped */
        }
        if (fr != GST_FLOW_OK) break;
    }
    return fr;
}
static GstFlowReturn enc_finish(GstVideoEncoder *ve) { return drain(GST_MI355H264ENC(ve), TRUE); }
static gboolean enc_flush(GstVideoEncoder *ve) { drain(GST_MI355H264ENC(ve), FALSE); return TRUE; }
/* ---- pinned input memory for the upstream element.  The ALLOCATION query is answered with a buffer pool whose memory comes from
 * mi355enc_host_alloc(): a source that takes the offer (videotestsrc, v4l2src in its copying modes, decoders, videoconvert) writes its
 * pictures straight into pinned memory and mi355enc_submit() transfers them from there asynchronously -- no staging pass over the picture
 * on the streaming thread.  A source that declines (its own pool, mmap'ed capture buffers) is staged by submit() as before. */
typedef struct { GstAllocator parent; } GstMi355PinAllocator;
typedef struct { GstAllocatorClass parent_class; } GstMi355PinAllocatorClass;
G_DEFINE_TYPE(GstMi355PinAllocator, gst_mi355_pin_allocator, GST_TYPE_ALLOCATOR)
static GstMemory *pin_alloc(GstAllocator *a, gsize size, GstAllocationParams *params) {
    (void)a;
    const gsize align = params->align | 63u, maxsize = size + params->prefix + params->padding + align;
    guint8 *data = (guint8 *)mi355enc_host_alloc(maxsize);
    if (!data) return NULL; /* GStreamer falls back to the default allocator */
    const gsize off = ((gsize)(-(gintptr)(data + params->prefix)) & align) + params->prefix; /* data + off is aligned */
    return gst_memory_new_wrapped((GstMemoryFlags)0, data, maxsize, off, size, data, (GDestroyNotify)mi355enc_host_free); /* sysmem semantics; freed through the notify */
}
static void pin_free(GstAllocator *a, GstMemory *m) { (void)a; (void)m; } /* never reached: wrapped memory belongs to the system allocator */
static void gst_mi355_pin_allocator_class_init(GstMi355PinAllocatorClass *k) { GST_ALLOCATOR_CLASS(k)->alloc = pin_alloc; GST_ALLOCATOR_CLASS(k)->free = pin_free; }
static void gst_mi355_pin_allocator_init(GstMi355PinAllocator *a) { GST_OBJECT_FLAG_SET(a, GST_ALLOCATOR_FLAG_CUSTOM_ALLOC); }

static gboolean enc_propose_allocation(GstVideoEncoder *ve, GstQuery *q) {
    GstMi355H264Enc *s = GST_MI355H264ENC(ve);
    GstCaps *caps = NULL;
    gboolean need_pool = FALSE;
    GstVideoInfo vi;
    gst_query_parse_allocation(q, &caps, &need_pool);
    if (s->pinned_input && caps && gst_video_info_from_caps(&vi, caps)) {
        GstAllocator *alloc = (GstAllocator *)g_object_new(gst_mi355_pin_allocator_get_type(), NULL);
        GstAllocationParams params;
        gst_allocation_params_init(&params);
        params.align = 63;
        gst_query_add_allocation_param(q, alloc, &params);
        const guint min = (guint)s->pipeline_depth + 2; /* pictures in flight hold their buffers until collect() */
        GstBufferPool *pool = gst_video_buffer_pool_new();
        GstStructure *cfg = gst_buffer_pool_get_config(pool);
        gst_buffer_pool_config_set_params(cfg, caps, (guint)GST_VIDEO_INFO_SIZE(&vi), min, 0);
        gst_buffer_pool_config_set_allocator(cfg, alloc, &params);
        gst_buffer_pool_config_add_option(cfg, GST_BUFFER_POOL_OPTION_VIDEO_META);
        if (gst_buffer_pool_set_config(pool, cfg)) gst_query_add_allocation_pool(q, pool, (guint)GST_VIDEO_INFO_SIZE(&vi), min, 0);
        gst_object_unref(pool);
        gst_object_unref(alloc);
    }
    gst_query_add_allocation_meta(q, GST_VIDEO_META_API_TYPE, NULL);
    return GST_VIDEO_ENCODER_CLASS(gst_mi355h264enc_parent_class)->propose_allocation(ve, q);
}
static void finalize(GObject *obj) {
    enc_stop(GST_VIDEO_ENCODER(obj));
    G_OBJECT_CLASS(gst_mi355h264enc_parent_class)->finalize(obj);
}

static void gst_mi355h264enc_class_init(GstMi355H264EncClass *k) {
    GObjectClass *g = G_OBJECT_CLASS(k);
    GstElementClass *e = GST_ELEMENT_CLASS(k);
    GstVideoEncoderClass *v = GST_VIDEO_ENCODER_CLASS(k);
    const GParamFlags F = (GParamFlags)(G_PARAM_READWRITE | G_PARAM_STATIC_STRINGS | GST_PARAM_MUTABLE_PLAYING);
    g->set_property = set_property; g->get_property = get_property; g->finalize = finalize;
    g_object_class_install_property(g, PROP_BPS, g_param_spec_uint("bps", "Bitrate (bit/s)",
        "Target bitrate; the property ceracoder's encoder_control writes.  In bit/s -- except on an element named venc_kbps, which the reference sends kbit/s (encoder_control.c:29-32,53)", 1, 1000000000, 2048000, F));
    g_object_class_install_property(g, PROP_BITRATE, g_param_spec_uint("bitrate", "Bitrate (kbit/s)",
        "Target bitrate in kbit/s (x264enc-compatible alias of bps)", 1, 1000000, 2048, F));
    g_object_class_install_property(g, PROP_KEY_INT_MAX, g_param_spec_uint("key-int-max", "Key-frame interval",
        "Maximal distance between two IDR pictures (0 = 250)", 0, 100000, 60, F));
    g_object_class_install_property(g, PROP_DEVICE_ID, g_param_spec_int("device-id", "HIP device", "GPU ordinal (one stream per GPU)", 0, 63, 0, F));
    g_object_class_install_property(g, PROP_ME_RANGE, g_param_spec_int("me-range", "Motion search range", "Full-search radius, integer pels", 1, 16, 16, F));
    g_object_class_install_property(g, PROP_QP, g_param_spec_int("qp", "Constant QP", "-1: rate control on; 0..51: constant quantiser", -1, 51, -1, F));
    g_object_class_install_property(g, PROP_PIPELINE_DEPTH, g_param_spec_int("pipeline-depth", "Pipeline depth",
        "0: output each picture before taking the next; 1: overlap host entropy coding with the next picture (+1 frame latency); 2: three pictures in flight (+2 frames)", 0, 2, 0, F));
    g_object_class_install_property(g, PROP_SPEED_PRESET, g_param_spec_enum("speed-preset", "Speed preset",
        "x264enc's presets mapped onto this encoder's tools: None / ultrafast = Constrained Baseline, one quantiser per picture; superfast = dct8x8 + i8x8 + aq-mode=1 "
        "(High profile), and so do the slower ones.  A tool property set explicitly wins over the preset", speed_preset_type(), 0, F));
    g_object_class_install_property(g, PROP_THREADS, g_param_spec_int("threads", "Entropy-coding threads",
        "Host threads that code one slice row-parallel (bit-identical output); like x264enc's property of the same name, 0 = automatic (a quarter of the CPUs, at most 8), 1 = streaming thread only", 0, 64, 0, F));
    g_object_class_install_property(g, PROP_VBV, g_param_spec_uint("vbv-buf-capacity", "VBV buffer (ms)",
        "Rate control's buffer model in milliseconds of stream at the setpoint (x264enc's property of the same name and default)", 100, 10000, 600, F));
    g_object_class_install_property(g, PROP_INTRA_IN_P, g_param_spec_int("intra-in-p", "Intra macroblocks in P pictures",
        "Macroblocks of P pictures may be coded intra (uncovered regions, partial scene changes): 0 never, 1 Intra_16x16, 2 Intra_4x4 as well", 0, 2, 1, F));
    g_object_class_install_property(g, PROP_EXCLUSIVE, g_param_spec_boolean("exclusive-gpu", "This stream has the GPU to itself",
        "One stream per GPU: kernels may wait on the device for other kernels' progress (a P picture's motion-compensation stage beside the previous picture's deblocking, the deblocker beside the intra macroblocks: about 10 % more frames/s); leave false when other processes encode on the same GPU", FALSE, F));
    g_object_class_install_property(g, PROP_AQ_MODE, g_param_spec_int("aq-mode", "Adaptive quantisation",
        "0: one quantiser per picture; 1: a QP offset per macroblock from the variance of its source samples (x264enc's aq-mode 1 in spirit), coded with mb_qp_delta", 0, 1, 0, F));
    g_object_class_install_property(g, PROP_INTRA_SLICES, g_param_spec_int("intra-slices", "Slices per I picture",
        "Slices per IDR picture, one NAL unit each (0: about 17 macroblock rows per slice, 4 at 1080p): the slices are reconstructed side by side", 0, 64, 0, F));
    g_object_class_install_property(g, PROP_SLICES, g_param_spec_int("slices", "Slices per P picture",
        "Slices per P picture, one NAL unit each (0: automatic, like intra-slices -- about 17 macroblock rows per slice, 4 at 1080p; 1: one slice): vector and intra prediction stop at a "
        "slice's first row; with slice-deblock the slices are independent dependency chains for the deblocking launch, which sets the picture period (x264enc: what threads / "
        "sliced-threads do to a picture).  Costs 2-4 % bitrate on panning content", 0, 64, 0, F));
    g_object_class_install_property(g, PROP_SLICE_DEBLOCK, g_param_spec_boolean("slice-deblock", "Slice-local deblocking",
        "The deblocking filter stops at slice boundaries (disable_deblocking_filter_idc 2) in I and P pictures; slice heights become multiples of four macroblock rows", TRUE, F));
    g_object_class_install_property(g, PROP_I8X8, g_param_spec_boolean("i8x8", "Intra 8x8",
        "With dct8x8: the macroblocks of I pictures may be Intra_8x8 (x264enc: part of dct8x8; here a switch of its own: IDR pictures 1.4 - 3.9 % smaller, a key-int 60 stream 1 - 3 % slower)", FALSE, F));
    g_object_class_install_property(g, PROP_SINGLE_STREAM, g_param_spec_boolean("single-stream", "One HIP stream",
        "Run every stage of this encoder in order on one HIP stream (one hardware queue): for many encoders sharing a GPU", FALSE, F));
    g_object_class_install_property(g, PROP_PINNED_INPUT, g_param_spec_boolean("pinned-input", "Offer pinned input buffers",
        "Answer the upstream ALLOCATION query with a buffer pool in pinned host memory: a source that takes it writes pictures the GPU can fetch without a staging copy", TRUE, F));
    g_object_class_install_property(g, PROP_SCENECUT, g_param_spec_boolean("scenecut", "Scene-cut recovery",
        "Code an IDR picture two pictures after a scene cut (detected from the summed motion cost; x264 decides inside its lookahead instead)", TRUE, F));
    g_object_class_install_property(g, PROP_DCT8X8, g_param_spec_boolean("dct8x8", "8x8 transform",
        "Adaptive spatial transform size as in x264enc: High-profile stream, P macroblocks use the 8x8 transform", FALSE, F));
    g_object_class_install_property(g, PROP_STATS, g_param_spec_boolean("stats", "Print stats", "Print a JSON line with counters when the encoder closes", FALSE, F));
    gst_element_class_add_static_pad_template(e, &sink_tmpl);
    gst_element_class_add_static_pad_template(e, &src_tmpl);
    gst_element_class_set_static_metadata(e, "MI355X H.264 encoder", "Codec/Encoder/Video/Hardware",
        "H.264 (Constrained Baseline) encoder on AMD Instinct MI355X via hand-written HIP kernels", "ceracoder-amd");
    v->start = enc_start; v->stop = enc_stop; v->set_format = enc_set_format; v->handle_frame = enc_handle_frame;
    v->finish = enc_finish; v->flush = enc_flush; v->propose_allocation = enc_propose_allocation;
}
static void gst_mi355h264enc_init(GstMi355H264Enc *s) {
    s->rate_raw = 2048; s->rate_is_bps = FALSE; s->key_int_max = 60; s->device_id = 0; s->me_range = 16; s->qp = -1; s->pipeline_depth = 0; s->speed_preset = 0;
    s->stats = FALSE; s->dct8x8 = -1; s->threads = 0; s->scenecut = TRUE; s->exclusive_gpu = FALSE; s->vbv_ms = 600; s->intra_in_p = -1; s->pinned_input = TRUE; s->aq_mode = -1; s->slices = -1; s->slice_deblock = -1; s->intra_slices = 0; s->i8x8 = -1; s->single_stream = FALSE; s->enc = NULL; s->input_state = NULL; s->max_au = 0; s->au_buf = NULL; s->last_pts = GST_CLOCK_TIME_NONE;
    s->us_map = s->us_submit = s->us_collect = s->us_output = s->us_push = s->us_frames = 0;
}

GType gst_mi355tsmux_get_type(void); /* gstmi355tsmux.c */
static gboolean plugin_init(GstPlugin *p) {
    GST_DEBUG_CATEGORY_INIT(mi355_debug, "mi355h264enc", 0, "MI355X H.264 encoder");
    return gst_element_register(p, "mi355h264enc", GST_RANK_NONE, GST_TYPE_MI355H264ENC) &&
           gst_element_register(p, "mi355tsmux", GST_RANK_NONE, gst_mi355tsmux_get_type());
}
GST_PLUGIN_DEFINE(GST_VERSION_MAJOR, GST_VERSION_MINOR, mi355h264enc, "MI355X-native H.264 encoder for ceracoder", plugin_init, VERSION,
                  "LGPL", PACKAGE, "https://github.com/CERALIVE/ceracoder")
