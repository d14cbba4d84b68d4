"""ctypes loader for the CPU oracle (TEST INFRASTRUCTURE ONLY -- see h264_oracle.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle_h264.so")

LEVELS_PER_MB = 408
MBINFO_DTYPE = np.dtype(
    [("mvx", "<i2"), ("mvy", "<i2"), ("mb_type", "u1"), ("i16_mode", "u1"), ("chroma_mode", "u1"),
     ("qp", "u1"), ("nzmask", "<u4"), ("cost", "<u4")]
)
assert MBINFO_DTYPE.itemsize == 16
IDEC = np.dtype([("modes4", "u1", (16,)), ("mode16", "u1"), ("cmode", "u1"), ("use_i4", "u1"), ("pad", "u1"), ("cost", "<u4"), ("cost_luma", "<u4"), ("rsv", "<u4")])
assert IDEC.itemsize == 32
IMV_DTYPE = np.dtype([("mvx", "<i2"), ("mvy", "<i2"), ("sad", "<u2"), ("bits", "<u2")])
SURF = 33 * 33
DROP_MAX, DROP_SKIP = 12, 255
F_MVDCOST, F_SKIPPROBE, F_DECIMATE, F_SATD, F_INTRAP, F_ALL = 1, 2, 4, 8, 16, 31
F_I4P, F_PART = 32, 64  # Intra_4x4 in P pictures; inter partitions (oracle-side groundwork: not produced by the device yet)


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("h264_enc_oracle.c", "h264_dec_oracle.c", "h264_oracle.h", "h264_tables_enc.h")]
    if force or not os.path.exists(_LIB) or any(os.path.getmtime(s) > os.path.getmtime(_LIB) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "liboracle_h264.so"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference/src/core"):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB)
        u8p, i16p, vp = C.POINTER(C.c_uint8), C.POINTER(C.c_int16), C.c_void_p
        L.orc_enc_open.restype = vp
        L.orc_enc_open.argtypes = [C.c_int] * 7
        L.orc_enc_close.argtypes = [vp]
        L.orc_enc_frame.restype = C.c_int
        L.orc_enc_frame.argtypes = [vp, vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, vp, C.c_size_t,
                                    C.POINTER(C.c_size_t), C.POINTER(C.c_int)]
        for n in ("recon_y", "recon_uv", "prefilter_y", "prefilter_uv", "mbinfo", "levels"):
            f = getattr(L, "orc_enc_" + n)
            f.restype = vp
            f.argtypes = [vp]
        L.orc_enc_mbw.argtypes = [vp]
        L.orc_enc_mbh.argtypes = [vp]
        L.orc_me_frame.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int]
        L.orc_me_frame.restype = None
        L.orc_me_select.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int]
        L.orc_me_select.restype = None
        L.orc_enc_set_me_iters.argtypes = [vp, C.c_int]
        L.orc_enc_set_me_iters.restype = None
        L.orc_enc_frame2.restype = C.c_int
        L.orc_enc_frame2.argtypes = [vp, vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_int)]
        for n in ("imv", "idec"):
            f = getattr(L, "orc_enc_" + n)
            f.restype = vp
            f.argtypes = [vp]
        L.orc_pmb_frame.argtypes = [vp] * 6 + [C.c_int] * 6 + [vp, vp, vp, vp, vp, C.c_int]
        L.orc_pmb_frame.restype = None
        L.orc_intra_p_frame.argtypes = [vp] * 4 + [C.c_int] * 4 + [vp, vp, vp]
        L.orc_intra_p_frame.restype = None
        L.orc_set_features.argtypes = [C.c_int]
        L.orc_set_features.restype = None
        L.orc_drop_threshold.restype = C.c_uint32
        L.orc_drop_threshold.argtypes = [C.c_int]
        L.orc_decimate_score.argtypes = [vp, C.c_int]
        L.orc_satd16.restype = C.c_uint32
        L.orc_satd16.argtypes = [vp, C.c_int, vp]
        L.orc_subpel_frame.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int]
        L.orc_subpel_frame.restype = None
        L.orc_enc_set_subpel.argtypes = [vp, C.c_int]
        L.orc_enc_set_subpel.restype = None
        L.orc_enc_set_scenecut.argtypes = [vp, C.c_int]
        L.orc_enc_set_aq.argtypes = [vp, C.c_int]
        L.orc_enc_set_aq.restype = None
        L.orc_aq_offsets.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp]
        L.orc_aq_offsets.restype = None
        L.orc_qp_chain.argtypes = [vp, C.c_int, C.c_int]
        L.orc_qp_chain.restype = None
        L.orc_qp_chain_slices.argtypes = [vp, C.c_int, C.c_int, C.c_int]
        L.orc_qp_chain_slices.restype = None
        L.orc_set_i8x8.argtypes = [C.c_int]
        L.orc_set_i8x8.restype = None
        L.orc_set_part_levels.argtypes = [vp]
        L.orc_set_part_levels.restype = None
        L.orc_set_slice_rows.argtypes = [C.c_int]
        L.orc_set_slice_rows.restype = None
        L.orc_auto_intra_slices.argtypes = [C.c_int]
        L.orc_enc_set_intra_slices.argtypes = [vp, C.c_int]
        L.orc_enc_set_intra_slices.restype = None
        L.orc_enc_set_p_slices.argtypes = [vp, C.c_int]
        L.orc_enc_set_p_slices.restype = None
        L.orc_enc_set_slice_deblock.argtypes = [vp, C.c_int]
        L.orc_enc_set_slice_deblock.restype = None
        L.orc_set_slice_deblock.argtypes = [C.c_int]
        L.orc_set_slice_deblock.restype = None
        L.orc_slice_rows_for.argtypes = [C.c_int, C.c_int, C.c_int]
        L.orc_enc_set_scenecut.restype = None
        L.orc_enc_set_sc_lag.argtypes = [vp, C.c_int]
        L.orc_enc_set_sc_lag.restype = None
        L.orc_set_transform8x8.argtypes = [C.c_int]
        L.orc_set_transform8x8.restype = None
        L.orc_fdct8.argtypes = [C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_idct8.argtypes = [C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_quant8.argtypes = [C.c_int] * 4
        L.orc_dequant8.argtypes = [C.c_int] * 3
        L.orc_zigzag8.argtypes = [C.c_int]
        L.orc_dec_zz8.argtypes = [C.c_int]
        L.orc_set_i4x4.argtypes = [C.c_int]
        L.orc_set_i4x4.restype = None
        L.orc_inter_frame.argtypes = [vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]
        L.orc_inter_frame.restype = None
        L.orc_intra_analyse.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, vp]
        L.orc_intra_analyse.restype = None
        L.orc_intra_decide.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]
        L.orc_intra_decide.restype = None
        L.orc_intra_frame.argtypes = [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]
        L.orc_intra_frame.restype = None
        L.orc_deblock_frame.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, vp]
        L.orc_deblock_frame.restype = None
        L.orc_write_headers.restype = C.c_size_t
        L.orc_write_headers.argtypes = [vp, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_write_slice.restype = C.c_size_t
        L.orc_write_slice.argtypes = [vp, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]
        L.orc_dec_open.restype = vp
        L.orc_dec_close.argtypes = [vp]
        L.orc_dec_decode.argtypes = [vp, vp, C.c_size_t]
        for n in ("width", "height", "coded_width", "coded_height"):
            getattr(L, "orc_dec_" + n).argtypes = [vp]
        L.orc_dec_y.restype = vp
        L.orc_dec_y.argtypes = [vp]
        L.orc_dec_uv.restype = vp
        L.orc_dec_uv.argtypes = [vp]
        L.orc_dec_error.restype = C.c_char_p
        L.orc_dec_error.argtypes = [vp]
        L.orc_fdct4.argtypes = [i16p, i16p]
        L.orc_idct4_add.argtypes = [C.POINTER(C.c_int32), u8p, C.c_int]
        L.orc_quant4.argtypes = [C.c_int] * 4
        L.orc_dequant4.argtypes = [C.c_int] * 3
        L.orc_nal_escape.restype = C.c_size_t
        L.orc_nal_escape.argtypes = [vp, C.c_size_t, vp, C.c_size_t]
        L.orc_ue_bits.argtypes = [C.c_uint32, C.POINTER(C.c_uint32)]
        L.orc_table_checksum.restype = C.c_uint32
        L.orc_table_checksum.argtypes = [C.c_int]
        L.orc_me_lambda.argtypes = [C.c_int]
        for n in ("orc_enc_vlc", "orc_dec_vlc"):
            getattr(L, n).argtypes = [C.c_int] * 4 + [C.POINTER(C.c_int)] * 2
        L.orc_enc_cbp_codenum.argtypes = [C.c_int, C.c_int]
        L.orc_dec_cbp.argtypes = [C.c_int, C.c_int]
        L.orc_dec_const.argtypes = [C.c_int, C.c_int]
        L.orc_cavlc_block_bits.argtypes = [vp, C.c_int, C.c_int, vp, C.c_size_t]
        L.orc_dec_set_capture.argtypes = [vp, vp, vp]
        L.orc_dec_set_capture.restype = None
        _lib = L
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _view(ptr, shape, dtype):
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    buf = (C.c_uint8 * n).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype).reshape(shape)


class Encoder:
    """Whole-encoder oracle: one NV12 frame + QP in, Annex-B access unit + stage outputs out."""

    def __init__(self, width, height, fps=60, gop=60, me_range=16, threads=1, subpel=True, scenecut=True, me_iters=None, sc_lag=2, aq=False, intra_slices=0, p_slices=0, slice_deblock_local=False):
        self.L = lib()
        self.h = self.L.orc_enc_open(width, height, fps, 1, gop, me_range, threads)
        if not self.h:
            raise ValueError("orc_enc_open failed")
        self.L.orc_enc_set_subpel(self.h, int(subpel))
        self.L.orc_enc_set_scenecut(self.h, int(scenecut))
        self.L.orc_enc_set_aq(self.h, int(aq))
        self.L.orc_enc_set_intra_slices(self.h, int(intra_slices))  # 0: the default (about 17 macroblock rows per slice)
        self.L.orc_enc_set_p_slices(self.h, int(p_slices))          # slices per P picture (0 / 1: one slice)
        self.L.orc_enc_set_slice_deblock(self.h, int(slice_deblock_local))  # the deblocking filter stops at slice boundaries (disable_deblocking_filter_idc 2)
        self.L.orc_enc_set_sc_lag(self.h, int(sc_lag))
        if me_iters is not None:
            self.L.orc_enc_set_me_iters(self.h, int(me_iters))
        self.width, self.height = width, height
        self.mbw, self.mbh = self.L.orc_enc_mbw(self.h), self.L.orc_enc_mbh(self.h)
        self._out = np.empty(self.mbw * self.mbh * 1024 + 4096, np.uint8)

    def encode(self, y, uv, qp, force_idr=False, drop=0):
        """drop: rate control's ladder below QP 51 for P pictures (0 .. DROP_MAX), DROP_SKIP = an all-skip picture."""
        y = np.ascontiguousarray(y, np.uint8)
        uv = np.ascontiguousarray(uv, np.uint8)
        assert y.shape == (self.height, self.width) and uv.shape == (self.height // 2, self.width)
        n, idr = C.c_size_t(0), C.c_int(0)
        r = self.L.orc_enc_frame2(self.h, _ptr(y), self.width, _ptr(uv), self.width, qp, int(drop), int(force_idr),
                                  _ptr(self._out), self._out.size, C.byref(n), C.byref(idr))
        if r:
            raise RuntimeError("orc_enc_frame2 -> %d" % r)
        return bytes(self._out[: n.value]), bool(idr.value)

    @property
    def imv(self):
        return _view(self.L.orc_enc_imv(self.h), (self.mbh * self.mbw,), IMV_DTYPE).copy()

    @property
    def idec(self):
        return _view(self.L.orc_enc_idec(self.h), (self.mbh * self.mbw,), IDEC).copy()

    def _plane(self, fn, rows):
        return _view(fn(self.h), (rows, self.mbw * 16), np.uint8).copy()

    @property
    def recon_y(self):
        return self._plane(self.L.orc_enc_recon_y, self.mbh * 16)

    @property
    def recon_uv(self):
        return self._plane(self.L.orc_enc_recon_uv, self.mbh * 8)

    @property
    def prefilter_y(self):
        return self._plane(self.L.orc_enc_prefilter_y, self.mbh * 16)

    @property
    def prefilter_uv(self):
        return self._plane(self.L.orc_enc_prefilter_uv, self.mbh * 8)

    @property
    def mbinfo(self):
        return _view(self.L.orc_enc_mbinfo(self.h), (self.mbh * self.mbw,), MBINFO_DTYPE).copy()

    @property
    def levels(self):
        return _view(self.L.orc_enc_levels(self.h), (self.mbh * self.mbw, LEVELS_PER_MB), np.int16).copy()

    def close(self):
        if self.h:
            self.L.orc_enc_close(self.h)
            self.h = None

    __del__ = close


class Decoder:
    def __init__(self):
        self.L = lib()
        self.h = self.L.orc_dec_open()

    def decode(self, au):
        buf = np.frombuffer(au, np.uint8)
        r = self.L.orc_dec_decode(self.h, _ptr(buf), buf.size)
        if r < 0:
            raise RuntimeError("oracle decoder: " + self.L.orc_dec_error(self.h).decode())
        if r == 0:
            return None
        cw, ch = self.L.orc_dec_coded_width(self.h), self.L.orc_dec_coded_height(self.h)
        y = _view(self.L.orc_dec_y(self.h), (ch, cw), np.uint8).copy()
        uv = _view(self.L.orc_dec_uv(self.h), (ch // 2, cw), np.uint8).copy()
        return y, uv

    def capture(self, n_mb):
        """Have the decoder record the syntax it parses (records + levels in the encoder's layout) from now on."""
        self.cap_mbi = np.zeros(n_mb, MBINFO_DTYPE)
        self.cap_lev = np.zeros((n_mb, LEVELS_PER_MB), np.int16)
        self.L.orc_dec_set_capture(self.h, _ptr(self.cap_mbi), _ptr(self.cap_lev))
        return self.cap_mbi, self.cap_lev

    @property
    def size(self):
        return self.L.orc_dec_width(self.h), self.L.orc_dec_height(self.h)

    def close(self):
        if self.h:
            self.L.orc_dec_close(self.h)
            self.h = None

    __del__ = close


def cavlc_block(coef, maxnum, nC):
    """One residual block through the encoder oracle's cavlc_block; returns the bits as a '0'/'1' string."""
    L = lib()
    c = np.ascontiguousarray(coef, np.int16)
    assert c.size == maxnum
    out = np.zeros(64, np.uint8)
    n = L.orc_cavlc_block_bits(_ptr(c), maxnum, nC, _ptr(out), out.size)
    if n < 0:
        raise RuntimeError("orc_cavlc_block_bits failed")
    return "".join("{:08b}".format(b) for b in out)[:n]


def me_frame(cur_y, ref_y, rng, qp, threads=1):
    """Whole-sample search: (SAD surfaces (n_mb, SURF) uint16, first selection IMV_DTYPE (n_mb,))."""
    L = lib()
    H, W = cur_y.shape
    n = (H // 16) * (W // 16)
    imv, surf = np.zeros(n, IMV_DTYPE), np.zeros((n, SURF), np.uint16)
    L.orc_me_frame(_ptr(np.ascontiguousarray(cur_y)), _ptr(np.ascontiguousarray(ref_y)), W, W // 16, H // 16, rng, qp, _ptr(surf), _ptr(imv), threads)
    return surf, imv


def me_select(surf, imv, mbw, mbh, rng, qp, threads=1):
    """One Jacobi iteration of the selection (bits against the median of the neighbours' vectors in imv)."""
    out = np.zeros(imv.size, IMV_DTYPE)
    lib().orc_me_select(_ptr(np.ascontiguousarray(surf)), mbw, mbh, rng, qp, _ptr(np.ascontiguousarray(imv)), _ptr(out), threads)
    return out


def se_bits(v):
    v = np.asarray(v, np.int64)
    k = np.where(v > 0, 2 * v - 1, -2 * v) + 1
    return 2 * np.floor(np.log2(k)).astype(np.int64) + 1


def imv_to_mbinfo(imv, qp):
    """Records for the two-stage (8x8-transform) path: vector + the absolute-vector cost its refinement compares."""
    m = np.zeros(imv.size, MBINFO_DTYPE)
    m["mvx"], m["mvy"] = imv["mvx"], imv["mvy"]
    m["cost"] = imv["sad"].astype(np.int64) + lib().orc_me_lambda(qp) * (se_bits(imv["mvx"]) + se_bits(imv["mvy"]))
    return m


def surf_to_device(surf):
    """(n_mb, 1089) oracle surfaces -> the device layout (n_mb, 35, 36); cells outside the 33 x 33 range are zero."""
    out = np.zeros((surf.shape[0], 35, 36), np.uint16)
    out[:, :33, :33] = surf.reshape(-1, 33, 33)
    return out


def pmb_frame(src_y, src_uv, ref_y, ref_uv, imv, surf, qp, drop=0, refine=True, idec=None, threads=1):
    """The fused P-macroblock stage (then the intra macroblocks it decided, if idec is given): rec_y, rec_uv, records, levels."""
    L = lib()
    H, W = src_y.shape
    src_y, src_uv, ref_y, ref_uv = (np.ascontiguousarray(a) for a in (src_y, src_uv, ref_y, ref_uv))
    rec_y, rec_uv = np.zeros_like(src_y), np.zeros_like(src_uv)
    mbi = np.zeros(imv.size, MBINFO_DTYPE)
    lev = np.zeros((imv.size, LEVELS_PER_MB), np.int16)
    imv = np.ascontiguousarray(imv)
    dec = np.ascontiguousarray(idec) if idec is not None else None
    L.orc_pmb_frame(_ptr(src_y), _ptr(src_uv), _ptr(ref_y), _ptr(ref_uv), _ptr(rec_y), _ptr(rec_uv), W, W // 16, H // 16, qp, drop, int(refine),
                    _ptr(imv), _ptr(np.ascontiguousarray(surf)), _ptr(dec) if dec is not None else None, _ptr(mbi), _ptr(lev), threads)
    pre = (mbi.copy(), rec_y.copy(), rec_uv.copy())
    if dec is not None:
        L.orc_intra_p_frame(_ptr(src_y), _ptr(src_uv), _ptr(rec_y), _ptr(rec_uv), W, W // 16, H // 16, qp, _ptr(dec), _ptr(mbi), _ptr(lev))
    return rec_y, rec_uv, mbi, lev, pre


_part_levels_keep = None


def set_part_levels(levels):
    """Stage functions (deblock_frame, write_slice): the levels of the picture, where the vectors of partitions 1 .. 3 of its inter macroblocks lie
    (None: every inter macroblock is one 16x16 partition).  The array is kept alive until the next call."""
    global _part_levels_keep
    _part_levels_keep = None if levels is None else np.ascontiguousarray(levels, np.int16)
    lib().orc_set_part_levels(None if levels is None else _ptr(_part_levels_keep))


def set_i8x8(on):
    """Process-wide (default off): Intra_8x8 macroblocks in the I pictures of a stream with the 8x8 transform (the product's cfg.i8x8; it needs intra_mode 0)."""
    lib().orc_set_i8x8(int(on))


def set_slice_rows(rows):
    """Stage functions: the picture being coded is cut into slices of `rows` macroblock rows (0: one slice)."""
    lib().orc_set_slice_rows(int(rows))


def set_slice_deblock(idc):
    """Stage functions (deblock_frame, write_slice): disable_deblocking_filter_idc of the picture's slices, 0 (across boundaries) or 2 (slice-local)."""
    lib().orc_set_slice_deblock(int(idc))


def slice_rows_for(mbh, slices=0, local_deblock=False):
    """Rows per slice the encoders use for `slices` slices per picture (0: the default number of an I picture); 0 = one slice.  With slice-local
    deblocking a multiple of four rows."""
    n = slices if slices > 0 else lib().orc_auto_intra_slices(int(mbh))
    return lib().orc_slice_rows_for(int(mbh), int(n), int(local_deblock))


def auto_slices(mbh):
    """The default number of slices per picture (about 17 macroblock rows each, at most 8): what the product uses for I pictures, and since r04 for P pictures."""
    return int(lib().orc_auto_intra_slices(int(mbh)))


def set_features(mask):
    """Process-wide ablation switches of the P-macroblock stage (default F_ALL)."""
    lib().orc_set_features(int(mask))


def drop_threshold(drop):
    return lib().orc_drop_threshold(int(drop))


def decimate_score(lev16, first=0):
    a = np.ascontiguousarray(lev16, np.int16)
    return lib().orc_decimate_score(_ptr(a), first)


def satd16(src16, pred16):
    a, b = np.ascontiguousarray(src16, np.uint8), np.ascontiguousarray(pred16, np.uint8)
    return lib().orc_satd16(_ptr(a), 16, _ptr(b))


def subpel_frame(cur_y, ref_y, mbi, qp, threads=1):
    L = lib()
    H, W = cur_y.shape
    mbi = np.ascontiguousarray(mbi).copy()
    L.orc_subpel_frame(_ptr(np.ascontiguousarray(cur_y)), _ptr(np.ascontiguousarray(ref_y)), W, W // 16, H // 16, qp, _ptr(mbi), threads)
    return mbi


def inter_frame(src_y, src_uv, ref_y, ref_uv, mbi, qp):
    L = lib()
    H, W = src_y.shape
    rec_y, rec_uv = np.zeros_like(src_y), np.zeros_like(src_uv)
    mbi = mbi.copy()
    lev = np.zeros((mbi.size, LEVELS_PER_MB), np.int16)
    L.orc_inter_frame(_ptr(src_y), _ptr(src_uv), _ptr(ref_y), _ptr(ref_uv), _ptr(rec_y), _ptr(rec_uv), W, W // 16,
                      H // 16, qp, _ptr(mbi), _ptr(lev))
    return rec_y, rec_uv, mbi, lev


def intra_analyse(src_y, src_uv):
    """SADs of all intra candidates against source-neighbour predictions: (n_mb, 152) uint16, 0xFFFF = unavailable."""
    L = lib()
    H, W = src_y.shape
    out = np.empty(((H // 16) * (W // 16), 152), np.uint16)
    L.orc_intra_analyse(_ptr(np.ascontiguousarray(src_y)), _ptr(np.ascontiguousarray(src_uv)), W, W // 16, H // 16, _ptr(out))
    return out





def intra_decide(isad, mbw, mbh, qp, i4x4=True):
    """Mode decisions of an I picture from the analysed SADs alone: structured array (n_mb,) of IDEC."""
    L = lib()
    out = np.zeros(mbw * mbh, IDEC)
    L.orc_intra_decide(_ptr(np.ascontiguousarray(isad)), mbw, mbh, qp, int(i4x4), _ptr(out))
    return out


def set_transform8x8(on):
    """Process-wide oracle switch (default off): High-profile stream, 8x8 transform for P macroblocks."""
    lib().orc_set_transform8x8(int(on))


def set_i4x4(on):
    """Process-wide oracle switch (default on): try Intra_4x4 in I pictures."""
    lib().orc_set_i4x4(int(on))


def intra_frame(src_y, src_uv, qp, drop=0):
    L = lib()
    H, W = src_y.shape
    rec_y, rec_uv = np.zeros_like(src_y), np.zeros_like(src_uv)
    mbi = np.zeros((H // 16) * (W // 16), MBINFO_DTYPE)
    lev = np.zeros((mbi.size, LEVELS_PER_MB), np.int16)
    L.orc_intra_frame(_ptr(src_y), _ptr(src_uv), _ptr(rec_y), _ptr(rec_uv), W, W // 16, H // 16, qp, drop, _ptr(mbi), _ptr(lev))
    return rec_y, rec_uv, mbi, lev


def deblock_frame(rec_y, rec_uv, mbi):
    L = lib()
    H, W = rec_y.shape
    y, uv = rec_y.copy(), rec_uv.copy()
    L.orc_deblock_frame(_ptr(y), _ptr(uv), W, W // 16, H // 16, _ptr(np.ascontiguousarray(mbi)))
    return y, uv


def write_headers(width, height, fps):
    L = lib()
    out = np.empty(256, np.uint8)
    n = L.orc_write_headers(_ptr(out), out.size, width, height, fps, 1)
    return bytes(out[:n])


def write_slice(mbw, mbh, is_idr, frame_num, idr_pic_id, qp, mbi, levels):
    L = lib()
    out = np.empty(mbw * mbh * 1024 + 4096, np.uint8)
    n = L.orc_write_slice(_ptr(out), out.size, mbw, mbh, int(is_idr), frame_num, idr_pic_id, qp,
                          _ptr(np.ascontiguousarray(mbi)), _ptr(np.ascontiguousarray(levels)))
    if not n:
        raise RuntimeError("orc_write_slice failed")
    return bytes(out[:n])
