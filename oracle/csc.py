"""oracle/csc.py -- TEST INFRASTRUCTURE ONLY: numpy restatement of the raw-format conversions the
product does on the device (ceracoder_amd/csrc/k_handover.hip, csc_kernel).

PARITY UNPINNED: the reference converts with GStreamer's `videoconvert`
(/root/reference/pipeline/generic/x264_superfast_camlink:4), which is not part of its tree; the
layouts follow the GStreamer raw-video format definitions (I420: Y, U, V planes, chroma 2x2
subsampled; NV12: Y plane + interleaved U,V; YUY2: Y0 U Y1 V; UYVY: U Y0 V Y1).  Going from
4:2:2 to 4:2:0 this repo takes the rounded mean of the two chroma rows (its own choice).  The
coded-size margin replicates the last visible row / column (pairs), like the encoder's staging.
Only tests/ may import this."""
import numpy as np

FMT_NV12, FMT_I420, FMT_YUY2, FMT_UYVY = range(4)


def _pad(plane, H, W, pair):
    """Replicate the last row and the last column (or the last column PAIR, for interleaved chroma)."""
    h, w = plane.shape
    out = np.empty((H, W), np.uint8)
    out[:h, :w] = plane
    if W > w:
        if pair:
            out[:h, w:] = np.tile(plane[:, w - 2:w], (1, (W - w) // 2))
        else:
            out[:h, w:] = plane[:, w - 1:w]
    if H > h:
        out[h:] = out[h - 1]
    return out


def to_nv12(fmt, planes, width, height):
    """Returns coded-size (multiples of 16) NV12 surfaces (Y, UV)."""
    W, H = (width + 15) // 16 * 16, (height + 15) // 16 * 16
    if fmt == FMT_I420:
        y, u, v = (np.asarray(p, np.uint8) for p in planes)
        uv = np.empty((height // 2, width), np.uint8)
        uv[:, 0::2] = u[:height // 2, :width // 2]
        uv[:, 1::2] = v[:height // 2, :width // 2]
        y = y[:height, :width]
    elif fmt in (FMT_YUY2, FMT_UYVY):
        p = np.asarray(planes[0], np.uint8)[:height, :2 * width]
        yo, uo, vo = (0, 1, 3) if fmt == FMT_YUY2 else (1, 0, 2)
        y = p[:, yo::2]
        c = np.empty((height, width), np.uint16)
        c[:, 0::2] = p[:, uo::4]
        c[:, 1::2] = p[:, vo::4]
        uv = ((c[0::2] + c[1::2] + 1) >> 1).astype(np.uint8)
    else:
        raise ValueError(fmt)
    return _pad(np.ascontiguousarray(y), H, W, False), _pad(np.ascontiguousarray(uv), H // 2, W, True)
