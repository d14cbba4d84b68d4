"""Shared helpers for the parity tests."""
import numpy as np

from ceracoder_amd import synth


def pad_planes(y, uv):
    """Replicate the last row/column up to the coded (multiple-of-16) size, as the encoder does."""
    h, w = y.shape
    H, W = (h + 15) // 16 * 16, (w + 15) // 16 * 16
    yp = np.empty((H, W), np.uint8)
    yp[:h, :w] = y
    yp[:h, w:] = y[:, w - 1:w]
    yp[h:, :] = yp[h - 1:h, :]
    up = np.empty((H // 2, W), np.uint8)
    up[:h // 2, :w] = uv
    for x in range(w, W, 2):
        up[:h // 2, x] = uv[:, w - 2]
        up[:h // 2, x + 1] = uv[:, w - 1]
    up[h // 2:, :] = up[h // 2 - 1:h // 2, :]
    return yp, up


def frames(w, h, n, kind="s2"):
    gen = synth.s2_frames(w, h, n) if kind == "s2" else synth.s3_frames(w, h, n)
    return [pad_planes(y, uv) + (y, uv) for y, uv in gen]


def mbinfo_equal(a, b, fields):
    return all(np.array_equal(a[f], b[f]) for f in fields)


def first_diff(a, b):
    d = np.argwhere(np.asarray(a) != np.asarray(b))
    return None if len(d) == 0 else (tuple(d[0]), len(d))


def cut_clip(w, h, n, cut):
    """S2 clip with a hard scene change at picture `cut`: from there on the pictures come from the S3 (noise) generator --
    nothing in the new scene is predictable from the old one, as at a real cut."""
    a = list(synth.s2_frames(w, h, n))
    b = list(synth.s3_frames(w, h, n))
    return [(np.ascontiguousarray(y), np.ascontiguousarray(uv)) for y, uv in (a[:cut] + b[cut:])]


def half_static_clip(w, h, n, static_lines):
    """S2 clip whose top `static_lines` lines never change (a still background above a moving scene): whole deblocking
    bands of the P pictures have no edge to filter."""
    fr = list(synth.s2_frames(w, h, n))
    y0, uv0 = fr[0]
    out = []
    for y, uv in fr:
        y, uv = y.copy(), uv.copy()
        y[:static_lines] = y0[:static_lines]
        uv[:static_lines // 2] = uv0[:static_lines // 2]
        out.append((y, uv))
    return out
