/*
 * mi355ts.h -- C ABI of the minimal MPEG-2 transport-stream muxer that follows the encoder
 * ("downstream wire formats without plugins-bad").
 *
 * What it replaces: the `h264parse config-interval=-1 ! ... ! mpegtsmux name=mux` tail of the
 * reference's pipeline files (/root/reference/pipeline/generic/x264_superfast_camlink:6-10)
 * for the video elementary stream, producing the 188-byte packets that
 * /root/reference/src/ceracoder.c:48-51 (TS_PKT_SIZE, 7 packets per SRT payload) and
 * new_buf_cb (/root/reference/src/ceracoder.c:297-339) regroup into 1316-byte sends.
 * gst-plugins-bad (h264parse, mpegtsmux) is absent from this image; this muxer has no
 * dependency beyond libc.  One program, one H.264 video stream (stream_type 0x1B), PCR on the
 * video PID, PAT/PMT repeated before every key frame and at least every 100 ms of stream time.
 *
 * Plain C, no GStreamer types; the element `mi355tsmux` (ceracoder_amd/csrc/gstmi355tsmux.c)
 * is a thin wrapper.  Host-only code: no HIP call is made by these functions.
 */
#ifndef MI355TS_H
#define MI355TS_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define MI355TS_PACKET 188
#define MI355TS_PID_PAT 0x0000
#define MI355TS_PID_PMT 0x1000
#define MI355TS_PID_VIDEO 0x0100

typedef struct mi355ts mi355ts_t;

mi355ts_t *mi355ts_open(void);
void mi355ts_close(mi355ts_t *m);
/* upper bound of the bytes mi355ts_mux() can emit for an access unit of `au_len` bytes */
size_t mi355ts_bound(size_t au_len);
/* Mux one Annex-B access unit (all NAL units of one picture).  pts_ns: presentation time in
 * nanoseconds (DTS = PTS: the encoder emits no B pictures).  An access-unit delimiter is
 * prepended when the access unit does not start with one (ISO/IEC 13818-1 2.14.1).
 * Returns 0, or -1 on bad arguments / -2 when `cap` is too small.  Output is a whole number of
 * 188-byte packets. */
int mi355ts_mux(mi355ts_t *m, const uint8_t *au, size_t au_len, int64_t pts_ns, int keyframe, uint8_t *out, size_t cap, size_t *out_len);
/* MPEG-2 CRC-32 (polynomial 0x04C11DB7, initial value all ones, no reflection) -- exported for tests */
uint32_t mi355ts_crc32(const uint8_t *p, size_t n);

#ifdef __cplusplus
}
#endif
#endif
