"""GPU: raw-format conversion kernel (I420 / YUY2 / UYVY -> NV12) against the numpy restatement in
oracle/csc.py, bit-exact, and a whole encode from I420 input equal to the NV12 path."""
import numpy as np
import pytest

from oracle import csc as O

pytestmark = pytest.mark.gpu

SIZES = [(16, 16), (64, 48), (322, 182), (1280, 720), (1920, 1080), (3840, 2160), (18, 34)]


def raw(fmt, w, h, seed, stride_pad=0):
    rng = np.random.default_rng(seed)
    if fmt == O.FMT_I420:
        mk = lambda ww, hh: np.ascontiguousarray(rng.integers(0, 256, (hh, ww + stride_pad), dtype=np.uint8)[:, :ww]) if not stride_pad else \
            rng.integers(0, 256, (hh, ww + stride_pad), dtype=np.uint8)[:, :ww]
        return [mk(w, h), mk(w // 2, h // 2), mk(w // 2, h // 2)]
    return [rng.integers(0, 256, (h, 2 * w + stride_pad), dtype=np.uint8)[:, :2 * w]]


@pytest.mark.parametrize("w,h", SIZES)
@pytest.mark.parametrize("fmt", [O.FMT_I420, O.FMT_YUY2, O.FMT_UYVY])
def test_csc_kernel_matches_numpy(E, fmt, w, h):
    e = E.Encoder(w, h, fixed_qp=30)
    planes = raw(fmt, w, h, seed=fmt * 100 + w)
    dy, duv = e.stage_csc(fmt, planes)
    oy, ouv = O.to_nv12(fmt, planes, w, h)
    assert np.array_equal(dy, oy), np.argwhere(dy != oy)[:4]
    assert np.array_equal(duv, ouv), np.argwhere(duv != ouv)[:4]
    e.close()


def test_csc_strided_and_misaligned_planes(E):
    """Row strides that are not multiples of 16 take the byte-wise path of the kernel's host copy; result is the same."""
    w, h = 322, 182
    e = E.Encoder(w, h, fixed_qp=30)
    for fmt in (O.FMT_I420, O.FMT_YUY2, O.FMT_UYVY):
        planes = raw(fmt, w, h, seed=7, stride_pad=5)
        assert not planes[0].flags["C_CONTIGUOUS"]
        strided = [p for p in planes]
        arrs = [np.ascontiguousarray(p) for p in planes]
        oy, ouv = O.to_nv12(fmt, arrs, w, h)
        dy, duv = e.stage_csc(fmt, arrs)
        assert np.array_equal(dy, oy) and np.array_equal(duv, ouv)
    e.close()


def test_encode_from_i420_equals_encode_from_nv12(E):
    from tests.util import frames
    w, h = 322, 182
    a, b = E.Encoder(w, h, gop=4, fixed_qp=28), E.Encoder(w, h, gop=4, fixed_qp=28)
    for i, (_, _, y, uv) in enumerate(frames(w, h, 5)):
        yy, cc = y[:h, :w], uv[:h // 2, :w]
        u, v = np.ascontiguousarray(cc[:, 0::2]), np.ascontiguousarray(cc[:, 1::2])
        au_nv12, k1 = a.encode(yy, cc, pts=i)
        b.submit_fmt(E.FMT_I420, [yy, u, v], pts=i)
        au_i420, k2, _, _ = b.collect()
        assert k1 == k2 and au_nv12 == au_i420, i
    a.close(); b.close()
