"""ctypes mirror of include/mi355enc.h -- plumbing only: every call below is one C-ABI call.

The library is the product; there is no Python or CPU fallback.  If libmi355enc.so is
missing or no HIP device is usable, construction raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MI355ENC_LIB") or os.path.join(_HERE, "libmi355enc.so")  # (MI355ENC_LIB: development builds, tools/build_variant.sh)

LEVELS_PER_MB = 408
MBINFO_DTYPE = np.dtype(
    [("mvx", "<i2"), ("mvy", "<i2"), ("mb_type", "u1"), ("i16_mode", "u1"), ("chroma_mode", "u1"),
     ("qp", "u1"), ("nzmask", "<u4"), ("cost", "<u4")]
)

FETCH_RECON_Y, FETCH_RECON_UV, FETCH_PREFILTER_Y, FETCH_PREFILTER_UV, FETCH_MBINFO, FETCH_LEVELS = range(6)
FMT_NV12, FMT_I420, FMT_YUY2, FMT_UYVY = range(4)
IDEC = np.dtype([("modes4", "u1", (16,)), ("mode16", "u1"), ("cmode", "u1"), ("use_i4", "u1"), ("pad", "u1"), ("cost", "<u4"), ("cost_luma", "<u4"), ("rsv", "<u4")])
IMV_DTYPE = np.dtype([("mvx", "<i2"), ("mvy", "<i2"), ("sad", "<u2"), ("bits", "<u2")])
SURF_ROWS, SURF_COLS = 35, 36
DROP_MAX, DROP_SKIP = 12, 255
STAGE_ME, STAGE_INTER, STAGE_INTRA, STAGE_DEBLOCK, STAGE_SUBPEL, STAGE_CSC_I420, STAGE_CSC_YUY2, STAGE_CSC_UYVY, STAGE_ME_SELECT, STAGE_PMB, STAGE_INTRA_P = range(11)

EXPORTS = [
    "mi355enc_abi_version", "mi355enc_strerror", "mi355enc_default_cfg", "mi355enc_open", "mi355enc_close",
    "mi355enc_set_bitrate", "mi355enc_get_bitrate", "mi355enc_set_fixed_qp", "mi355enc_set_fixed_drop", "mi355enc_stage_me_select", "mi355enc_stage_me_select_next", "mi355enc_encode", "mi355enc_submit",
    "mi355enc_submit_device", "mi355enc_pending", "mi355enc_collect", "mi355enc_get_stats", "mi355enc_reset_stats",
    "mi355enc_max_au_bytes", "mi355enc_fetch", "mi355enc_mb_width", "mi355enc_mb_height", "mi355enc_stage_me",
    "mi355enc_stage_subpel", "mi355enc_stage_inter", "mi355enc_stage_pmb", "mi355enc_stage_intra", "mi355enc_stage_intra_analyse", "mi355enc_stage_csc", "mi355enc_submit_fmt", "mi355enc_host_write_slice_packed", "mi355enc_stage_deblock", "mi355enc_time_stage",
    "mi355enc_host_write_headers", "mi355enc_host_write_slice", "mi355enc_host_set_slice_rows", "mi355enc_host_set_p_slices", "mi355enc_stage_set_slice_rows", "mi355enc_slice_rows", "mi355enc_p_slice_rows", "mi355enc_stage_set_slice_deblock", "mi355enc_rc_init", "mi355enc_rc_set_bitrate",
    "mi355enc_rc_pick", "mi355enc_rc_update", "mi355enc_host_cavlc_block", "mi355enc_debug_trip_wait", "mi355enc_host_alloc", "mi355enc_host_free",
]


class Cfg(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("fps_num", C.c_int), ("fps_den", C.c_int), ("gop", C.c_int),
                ("me_range", C.c_int), ("bitrate_bps", C.c_uint32), ("device_id", C.c_int), ("fixed_qp", C.c_int),
                ("qp_min", C.c_int), ("qp_max", C.c_int), ("pipeline_depth", C.c_int), ("profile_events", C.c_int),
                ("use_graphs", C.c_int), ("keep_prefilter", C.c_int), ("transform8x8", C.c_int), ("i4x4", C.c_int), ("subpel", C.c_int), ("deblock_mode", C.c_int), ("intra_in_p", C.c_int), ("cavlc_threads", C.c_int), ("intra_mode", C.c_int), ("vbv_ms", C.c_int), ("scenecut", C.c_int), ("exclusive_device", C.c_int), ("aq_mode", C.c_int), ("single_stream", C.c_int), ("intra_slices", C.c_int), ("partitions", C.c_int), ("profile_overlap", C.c_int), ("i8x8", C.c_int), ("slices", C.c_int), ("slice_deblock", C.c_int)]


class Stats(C.Structure):
    _fields_ = [("frames", C.c_uint64), ("idr_frames", C.c_uint64), ("bytes", C.c_uint64), ("last_qp", C.c_uint32),
                ("last_bytes", C.c_uint32), ("target_bps", C.c_uint32), ("ms_me", C.c_double), ("ms_inter", C.c_double),
                ("ms_intra", C.c_double), ("ms_deblock", C.c_double), ("ms_total_gpu", C.c_double), ("ms_subpel", C.c_double), ("n_me", C.c_uint64),
                ("n_inter", C.c_uint64), ("n_intra", C.c_uint64), ("n_deblock", C.c_uint64), ("ms_entropy", C.c_double),
                ("ms_wait", C.c_double), ("n_total_gpu", C.c_uint64), ("ms_deblock_idr", C.c_double), ("n_deblock_idr", C.c_uint64), ("cavlc_threads", C.c_uint32), ("last_drop", C.c_uint32), ("ms_select", C.c_double), ("ms_analyse_p", C.c_double), ("ms_intra_p", C.c_double), ("skip_pictures", C.c_uint64), ("ms_open", C.c_double),
                ("recoveries", C.c_uint32), ("last_error_word", C.c_uint32), ("safe_level", C.c_uint32), ("pinned_inputs", C.c_uint64)]


_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libmi355enc.so not built (run `python -c 'import __graft_entry__ as g; g.build()'`)")
        L = C.CDLL(LIB_PATH)
        vp = C.c_void_p
        L.mi355enc_strerror.restype = C.c_char_p
        L.mi355enc_strerror.argtypes = [C.c_int]
        L.mi355enc_default_cfg.restype = None
        L.mi355enc_default_cfg.argtypes = [C.POINTER(Cfg), C.c_int, C.c_int, C.c_int, C.c_int]
        L.mi355enc_open.argtypes = [C.POINTER(Cfg), C.POINTER(vp)]
        L.mi355enc_close.restype = None
        L.mi355enc_close.argtypes = [vp]
        L.mi355enc_set_bitrate.argtypes = [vp, C.c_uint32]
        L.mi355enc_get_bitrate.restype = C.c_uint32
        L.mi355enc_get_bitrate.argtypes = [vp]
        L.mi355enc_set_fixed_qp.argtypes = [vp, C.c_int]
        L.mi355enc_set_fixed_drop.argtypes = [vp, C.c_int]
        L.mi355enc_encode.argtypes = [vp, vp, C.c_int, vp, C.c_int, C.c_int64, C.c_int, vp, C.c_size_t,
                                      C.POINTER(C.c_size_t), C.POINTER(C.c_int)]
        L.mi355enc_submit.argtypes = [vp, vp, C.c_int, vp, C.c_int, C.c_int64, C.c_int]
        L.mi355enc_submit_device.argtypes = [vp, vp, C.c_int, vp, C.c_int, C.c_int64, C.c_int]
        L.mi355enc_pending.argtypes = [vp]
        L.mi355enc_collect.argtypes = [vp, vp, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_int),
                                       C.POINTER(C.c_int64), C.POINTER(C.c_int)]
        L.mi355enc_get_stats.argtypes = [vp, C.POINTER(Stats)]
        L.mi355enc_reset_stats.restype = None
        L.mi355enc_reset_stats.argtypes = [vp]
        L.mi355enc_max_au_bytes.restype = C.c_size_t
        L.mi355enc_max_au_bytes.argtypes = [vp]
        L.mi355enc_fetch.argtypes = [vp, C.c_int, vp, C.c_size_t]
        L.mi355enc_mb_width.argtypes = [vp]
        L.mi355enc_mb_height.argtypes = [vp]
        L.mi355enc_stage_me.argtypes = [vp, vp, vp, C.c_int, vp, vp]
        L.mi355enc_stage_me_select.argtypes = [vp, vp, vp, C.c_int, vp]
        L.mi355enc_stage_subpel.argtypes = [vp, vp, vp, C.c_int, vp]
        L.mi355enc_stage_inter.argtypes = [vp, vp, vp, vp, vp, C.c_int, vp, vp, vp, vp]
        L.mi355enc_stage_pmb.argtypes = [vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, C.c_int, vp, vp, vp, vp]
        L.mi355enc_stage_intra.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp, vp, vp, vp]
        L.mi355enc_stage_intra_analyse.argtypes = [vp, vp, vp, C.c_int, vp, vp]
        L.mi355enc_stage_deblock.argtypes = [vp, vp, vp, vp]
        L.mi355enc_submit_fmt.argtypes = [vp, C.c_int, vp, vp, C.c_int64, C.c_int]
        L.mi355enc_stage_csc.argtypes = [vp, C.c_int, vp, vp, vp, vp]
        L.mi355enc_debug_trip_wait.argtypes = [vp, C.c_uint]
        L.mi355enc_host_alloc.restype = vp
        L.mi355enc_host_alloc.argtypes = [C.c_size_t]
        L.mi355enc_host_free.restype = None
        L.mi355enc_host_free.argtypes = [vp]
        L.mi355enc_time_stage.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_double)]
        L.mi355enc_host_write_headers.argtypes = [C.c_int] * 5 + [vp, C.c_size_t, C.POINTER(C.c_size_t)]
        L.mi355enc_host_set_slice_rows.argtypes = [C.c_int]
        L.mi355enc_host_set_slice_rows.restype = None
        L.mi355enc_host_set_p_slices.argtypes = [C.c_int, C.c_int]
        L.mi355enc_host_set_p_slices.restype = None
        L.mi355enc_stage_set_slice_rows.argtypes = [vp, C.c_int]
        L.mi355enc_slice_rows.argtypes = [vp]
        L.mi355enc_p_slice_rows.argtypes = [vp]
        L.mi355enc_stage_set_slice_deblock.argtypes = [vp, C.c_int]
        L.mi355enc_host_write_slice.argtypes = [C.c_int] * 7 + [vp, vp, vp, C.c_size_t, C.POINTER(C.c_size_t)]
        L.mi355enc_host_write_slice_packed.argtypes = [C.c_int] * 8 + [vp, vp, vp, C.c_size_t, C.POINTER(C.c_size_t)]
        L.mi355enc_host_cavlc_block.argtypes = [vp, C.c_int, C.c_int, vp, C.c_size_t]
        L.mi355enc_rc_init.restype = None
        L.mi355enc_rc_init.argtypes = [vp, C.c_double, C.c_int, C.c_uint32, C.c_int, C.c_int]
        L.mi355enc_rc_set_bitrate.restype = None
        L.mi355enc_rc_set_bitrate.argtypes = [vp, C.c_uint32]
        L.mi355enc_rc_pick.restype = None
        L.mi355enc_rc_pick.argtypes = [vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.mi355enc_rc_update.restype = None
        L.mi355enc_rc_update.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_size_t]
        _lib = L
    return _lib


def host_cavlc_block(coef, maxnum, nC):
    """One residual block through the product's CAVLC block coder; returns the bits as a '0'/'1' string."""
    L = load()
    c = np.ascontiguousarray(coef, np.int16)
    assert c.size == maxnum
    out = np.zeros(64, np.uint8)
    n = L.mi355enc_host_cavlc_block(c.ctypes.data_as(C.c_void_p), maxnum, nC, out.ctypes.data_as(C.c_void_p), out.size)
    if n < 0:
        raise RuntimeError("mi355enc_host_cavlc_block: %d" % n)
    return "".join("{:08b}".format(b) for b in out)[:n]


def host_write_headers(width, height, fps_num, fps_den=1, transform8x8=False):
    L = load()
    out, n = np.empty(256, np.uint8), C.c_size_t(0)
    r = L.mi355enc_host_write_headers(width, height, fps_num, fps_den, int(transform8x8), out.ctypes.data_as(C.c_void_p), out.size, C.byref(n))
    if r:
        raise RuntimeError("mi355enc_host_write_headers: %d" % r)
    return bytes(out[: n.value])


def host_set_slice_rows(rows):
    """The host stage functions write I pictures as slices of `rows` macroblock rows from now on (0: one slice)."""
    load().mi355enc_host_set_slice_rows(int(rows))


def host_set_p_slices(rows, dbf_idc=0):
    """... and P pictures as slices of `rows` rows (0: one slice); dbf_idc: the disable_deblocking_filter_idc of every slice header (0 or 2)."""
    load().mi355enc_host_set_p_slices(int(rows), int(dbf_idc))


def host_write_slice(mbw, mbh, is_idr, frame_num, idr_pic_id, qp, mbinfo, levels, transform8x8=False):
    L = load()
    out, n = np.empty(mbw * mbh * 1536 + 4096, np.uint8), C.c_size_t(0)
    mbinfo, levels = np.ascontiguousarray(mbinfo), np.ascontiguousarray(levels, np.int16)
    r = L.mi355enc_host_write_slice(mbw, mbh, int(is_idr), frame_num, idr_pic_id, qp, int(transform8x8), mbinfo.ctypes.data_as(C.c_void_p),
                                    levels.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), out.size, C.byref(n))
    if r:
        raise RuntimeError("mi355enc_host_write_slice: %d" % r)
    return bytes(out[: n.value])


def host_write_slice_packed(mbw, mbh, is_idr, frame_num, idr_pic_id, qp, mbinfo, levels, threads=1, transform8x8=False):
    L = load()
    out, n = np.empty(mbw * mbh * 1536 + 4096, np.uint8), C.c_size_t(0)
    mbinfo, levels = np.ascontiguousarray(mbinfo), np.ascontiguousarray(levels, np.int16)
    r = L.mi355enc_host_write_slice_packed(mbw, mbh, int(is_idr), frame_num, idr_pic_id, qp, int(transform8x8), int(threads), mbinfo.ctypes.data_as(C.c_void_p),
                                           levels.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), out.size, C.byref(n))
    if r:
        raise RuntimeError("mi355enc_host_write_slice_packed: %d" % r)
    return bytes(out[: n.value])


RC_BYTES = 512  # include/mi355enc.h MI355ENC_RC_BYTES


class RateControl:
    """The encoder's rate-control model by itself (host logic; no device)."""

    def __init__(self, fps, gop, bps, qp_min=10, qp_max=51):
        self.L = load()
        self.buf = (C.c_uint8 * RC_BYTES)()
        self.L.mi355enc_rc_init(self.buf, float(fps), gop, bps, qp_min, qp_max)

    def set_bitrate(self, bps):
        self.L.mi355enc_rc_set_bitrate(self.buf, int(bps))

    def pick(self, is_idr):
        """-> (qp, drop): drop 0 .. DROP_MAX is the ladder below QP 51, DROP_SKIP an all-skip picture"""
        qp, drop = C.c_int(0), C.c_int(0)
        self.L.mi355enc_rc_pick(self.buf, int(is_idr), C.byref(qp), C.byref(drop))
        return qp.value, drop.value

    def update(self, is_idr, qp, drop, nbytes):
        self.L.mi355enc_rc_update(self.buf, int(is_idr), qp, drop, nbytes)


class PinnedBuffer:
    """nbytes of pinned host memory from mi355enc_host_alloc(), viewed as a numpy uint8 array (`.array`); pictures submitted from it
    are DMA'd in place."""

    def __init__(self, nbytes):
        self.L = load()
        self.ptr = self.L.mi355enc_host_alloc(nbytes)
        if not self.ptr:
            raise EncoderError("mi355enc_host_alloc(%d) failed" % nbytes)
        self.array = np.ctypeslib.as_array((C.c_uint8 * nbytes).from_address(self.ptr))

    def free(self):
        if getattr(self, "ptr", None):
            self.array = None
            self.L.mi355enc_host_free(self.ptr)
            self.ptr = None

    __del__ = free


class EncoderError(RuntimeError):
    pass


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class Encoder:
    """One H.264 stream on one GPU.  Arguments mirror the element's properties
    (bitrate in bits/s as written through `bps`, key-int-max -> gop)."""

    def __init__(self, width, height, fps=60, gop=60, bitrate_bps=6_000_000, device_id=0, fixed_qp=-1, me_range=16,
                 pipeline_depth=0, profile_events=False, use_graphs=True, keep_prefilter=False, fps_den=1, deblock_mode=0, subpel=True, i4x4=True, transform8x8=False, intra_in_p=True, cavlc_threads=0, intra_mode=0, scenecut=True, exclusive=False, aq=False, single_stream=False, intra_slices=0, profile_overlap=False, partitions=False, i8x8=False, slices="mirror", slice_deblock="mirror"):
        self.L = load()
        cfg = Cfg()
        self.L.mi355enc_default_cfg(C.byref(cfg), width, height, fps, fps_den)
        cfg.gop, cfg.bitrate_bps, cfg.device_id, cfg.fixed_qp, cfg.me_range = gop, bitrate_bps, device_id, fixed_qp, me_range
        cfg.pipeline_depth, cfg.profile_events, cfg.use_graphs, cfg.keep_prefilter = (
            pipeline_depth, int(profile_events), int(use_graphs), int(keep_prefilter))
        cfg.deblock_mode = deblock_mode
        cfg.intra_in_p = int(intra_in_p)  # True / 1: Intra_16x16 (the default); 2: Intra_4x4 as well
        cfg.cavlc_threads = int(cavlc_threads)
        cfg.scenecut = int(scenecut)
        cfg.exclusive_device = int(exclusive)  # this encoder has the GPU to itself: kernels may wait on the device for each other (include/mi355enc.h)
        cfg.intra_mode = int(intra_mode)
        cfg.aq_mode = int(aq)
        cfg.single_stream = int(single_stream)
        cfg.profile_overlap = int(profile_overlap)  # sampled P pictures keep the free-running schedule (timers include device-side waits)
        cfg.i8x8 = int(i8x8)  # with transform8x8: Intra_8x8 macroblocks in I pictures (intra_mode 0)
        cfg.partitions = int(partitions)  # P macroblocks may be split into 16x8 / 8x16 / 8x8 partitions
        cfg.intra_slices = int(intra_slices)  # 0: about 17 macroblock rows per slice (1080p: 4 slices per I picture)
        # slices / slice_deblock: this mirror defaults to the one-slice P pictures and the filter across slice boundaries of rounds 1-3 (what the stage-by-stage parity
        # suite was written against); None = the library's own default (mi355enc_default_cfg: P pictures sliced like I pictures, slice-local deblocking)
        # (dev tools: MI355ENC_MIRROR_DEFAULTS=library makes the mirror's default the library's)
        lib_defaults = os.environ.get("MI355ENC_MIRROR_DEFAULTS") == "library"
        if isinstance(slices, str):
            slices = None if lib_defaults else 1
        if isinstance(slice_deblock, str):
            slice_deblock = None if lib_defaults else False
        if slices is not None:
            cfg.slices = int(slices)  # slices per P picture (0: automatic, 1: one)
        if slice_deblock is not None:
            cfg.slice_deblock = int(slice_deblock)  # the deblocking filter stops at slice boundaries (disable_deblocking_filter_idc 2)
        cfg.subpel = int(subpel)
        cfg.i4x4 = int(i4x4)
        cfg.transform8x8 = int(transform8x8)
        self.h = C.c_void_p()
        self._chk(self.L.mi355enc_open(C.byref(cfg), C.byref(self.h)), "open", close_on_fail=True)
        self.width, self.height = width, height
        self.mbw, self.mbh = self.L.mi355enc_mb_width(self.h), self.L.mi355enc_mb_height(self.h)
        self._out = np.empty(self.L.mi355enc_max_au_bytes(self.h), np.uint8)

    def _chk(self, r, what, close_on_fail=False):
        if r != 0:
            msg = self.L.mi355enc_strerror(r).decode()
            if close_on_fail and self.h:
                self.L.mi355enc_close(self.h)
                self.h = None
            raise EncoderError("mi355enc_%s: %s (%d)" % (what, msg, r))

    def close(self):
        if getattr(self, "h", None):
            self.L.mi355enc_close(self.h)
            self.h = None

    __del__ = close

    def set_bitrate(self, bps):
        self._chk(self.L.mi355enc_set_bitrate(self.h, int(bps)), "set_bitrate")

    def set_fixed_qp(self, qp):
        self._chk(self.L.mi355enc_set_fixed_qp(self.h, int(qp)), "set_fixed_qp")

    def set_fixed_drop(self, drop):
        self._chk(self.L.mi355enc_set_fixed_drop(self.h, int(drop)), "set_fixed_drop")

    def encode(self, y, uv, pts=0, force_idr=False):
        y = np.ascontiguousarray(y, np.uint8)
        uv = np.ascontiguousarray(uv, np.uint8)
        n, key = C.c_size_t(0), C.c_int(0)
        self._chk(self.L.mi355enc_encode(self.h, _p(y), y.strides[0], _p(uv), uv.strides[0], pts, int(force_idr),
                                         _p(self._out), self._out.size, C.byref(n), C.byref(key)), "encode")
        return bytes(self._out[: n.value]), bool(key.value)

    def submit(self, y, uv, pts=0, force_idr=False):
        y = np.ascontiguousarray(y, np.uint8)
        uv = np.ascontiguousarray(uv, np.uint8)
        self._chk(self.L.mi355enc_submit(self.h, _p(y), y.strides[0], _p(uv), uv.strides[0], pts, int(force_idr)), "submit")

    def _planes(self, planes):
        arrs = [np.ascontiguousarray(a, np.uint8) for a in planes]
        pp = (C.c_void_p * 3)(*([a.ctypes.data for a in arrs] + [None] * (3 - len(arrs))))
        ss = (C.c_int * 3)(*([a.strides[0] for a in arrs] + [0] * (3 - len(arrs))))
        return arrs, pp, ss

    def submit_fmt(self, fmt, planes, pts=0, force_idr=False):
        """fmt: FMT_I420 (planes Y, U, V), FMT_YUY2 / FMT_UYVY (one packed plane, 2 bytes per pixel), FMT_NV12 (Y, UV)."""
        arrs, pp, ss = self._planes(planes)
        self._chk(self.L.mi355enc_submit_fmt(self.h, fmt, pp, ss, pts, int(force_idr)), "submit_fmt")

    def stage_csc(self, fmt, planes):
        arrs, pp, ss = self._planes(planes)
        oy = np.empty((self.mbh * 16, self.mbw * 16), np.uint8)
        ouv = np.empty((self.mbh * 8, self.mbw * 16), np.uint8)
        self._chk(self.L.mi355enc_stage_csc(self.h, fmt, pp, ss, _p(oy), _p(ouv)), "stage_csc")
        return oy, ouv

    def submit_device(self, y_ptr, y_stride, uv_ptr, uv_stride, pts=0, force_idr=False):
        self._chk(self.L.mi355enc_submit_device(self.h, y_ptr, y_stride, uv_ptr, uv_stride, pts, int(force_idr)), "submit_device")

    def collect(self, copy=True):
        n, key, pts, qp = C.c_size_t(0), C.c_int(0), C.c_int64(0), C.c_int(0)
        self._chk(self.L.mi355enc_collect(self.h, _p(self._out), self._out.size, C.byref(n), C.byref(key), C.byref(pts),
                                          C.byref(qp)), "collect")
        au = bytes(self._out[: n.value]) if copy else n.value
        return au, bool(key.value), pts.value, qp.value

    @property
    def last_drop(self):
        """drop level of the last collected picture (0, 1 .. DROP_MAX, DROP_SKIP)"""
        return int(self.stats().last_drop)

    @property
    def pending(self):
        return self.L.mi355enc_pending(self.h)

    def stats(self):
        s = Stats()
        self._chk(self.L.mi355enc_get_stats(self.h, C.byref(s)), "get_stats")
        return s

    def reset_stats(self):
        self.L.mi355enc_reset_stats(self.h)

    def fetch(self, what):
        H, W, n = self.mbh * 16, self.mbw * 16, self.mbw * self.mbh
        if what in (FETCH_RECON_Y, FETCH_PREFILTER_Y):
            a = np.empty((H, W), np.uint8)
        elif what in (FETCH_RECON_UV, FETCH_PREFILTER_UV):
            a = np.empty((H // 2, W), np.uint8)
        elif what == FETCH_MBINFO:
            a = np.empty(n, MBINFO_DTYPE)
        else:
            a = np.empty((n, LEVELS_PER_MB), np.int16)
        self._chk(self.L.mi355enc_fetch(self.h, what, _p(a), a.nbytes), "fetch")
        return a

    # ---- single-stage entry points (coded-size host planes)
    def stage_me(self, cur_y, ref_y, qp):
        """-> (surfaces (n_mb, 35, 36) uint16: [dy+16][dx+16], first selection IMV_DTYPE (n_mb,))"""
        n = self.mbw * self.mbh
        imv = np.zeros(n, IMV_DTYPE)
        surf = np.zeros((n, SURF_ROWS, SURF_COLS), np.uint16)
        self._chk(self.L.mi355enc_stage_me(self.h, _p(np.ascontiguousarray(cur_y)), _p(np.ascontiguousarray(ref_y)), qp, _p(surf), _p(imv)), "stage_me")
        return surf, imv

    def stage_me_select(self, surf, imv, qp):
        out = np.zeros(imv.size, IMV_DTYPE)
        self._chk(self.L.mi355enc_stage_me_select(self.h, _p(np.ascontiguousarray(surf, np.uint16)), _p(np.ascontiguousarray(imv)), qp, _p(out)), "stage_me_select")
        return out

    def stage_me_select_next(self, surf, imv, prev, qp):
        """the same iteration as the encoder's later passes run it: macroblocks whose predictors are unchanged against `prev` (the field `imv` was selected from) are copied"""
        out = np.zeros(imv.size, IMV_DTYPE)
        self._chk(self.L.mi355enc_stage_me_select_next(self.h, _p(np.ascontiguousarray(surf, np.uint16)), _p(np.ascontiguousarray(imv)), _p(np.ascontiguousarray(prev)), qp, _p(out)), "stage_me_select_next")
        return out

    def stage_subpel(self, cur_y, ref_y, mbi, qp):
        mbi = np.ascontiguousarray(mbi).copy()
        self._chk(self.L.mi355enc_stage_subpel(self.h, _p(np.ascontiguousarray(cur_y)), _p(np.ascontiguousarray(ref_y)), qp, _p(mbi)), "stage_subpel")
        return mbi

    def stage_inter(self, src_y, src_uv, ref_y, ref_uv, mbi, qp):
        mbi = np.ascontiguousarray(mbi).copy()
        rec_y, rec_uv = np.empty_like(src_y), np.empty_like(src_uv)
        lev = np.empty((mbi.size, LEVELS_PER_MB), np.int16)
        self._chk(self.L.mi355enc_stage_inter(self.h, _p(np.ascontiguousarray(src_y)), _p(np.ascontiguousarray(src_uv)),
                                              _p(np.ascontiguousarray(ref_y)), _p(np.ascontiguousarray(ref_uv)), qp, _p(mbi),
                                              _p(rec_y), _p(rec_uv), _p(lev)), "stage_inter")
        return rec_y, rec_uv, mbi, lev

    def stage_pmb(self, src_y, src_uv, ref_y, ref_uv, imv, surf, qp, drop=0, refine=True, idec=None, run_intra_p=True):
        """surf: device layout (n_mb, 35, 36).  -> rec_y, rec_uv, records, levels"""
        n = self.mbw * self.mbh
        mbi = np.zeros(n, MBINFO_DTYPE)
        rec_y, rec_uv = np.empty_like(src_y), np.empty_like(src_uv)
        lev = np.empty((n, LEVELS_PER_MB), np.int16)
        dec = np.ascontiguousarray(idec) if idec is not None else None
        self._chk(self.L.mi355enc_stage_pmb(self.h, _p(np.ascontiguousarray(src_y)), _p(np.ascontiguousarray(src_uv)),
                                            _p(np.ascontiguousarray(ref_y)), _p(np.ascontiguousarray(ref_uv)), qp, int(drop), int(refine),
                                            _p(np.ascontiguousarray(imv)), _p(np.ascontiguousarray(surf, np.uint16)), _p(dec) if dec is not None else None,
                                            int(run_intra_p), _p(mbi), _p(rec_y), _p(rec_uv), _p(lev)), "stage_pmb")
        return rec_y, rec_uv, mbi, lev

    def stage_intra(self, src_y, src_uv, qp, drop=0):
        mbi = np.zeros(self.mbw * self.mbh, MBINFO_DTYPE)
        rec_y, rec_uv = np.empty_like(src_y), np.empty_like(src_uv)
        lev = np.empty((mbi.size, LEVELS_PER_MB), np.int16)
        self._chk(self.L.mi355enc_stage_intra(self.h, _p(np.ascontiguousarray(src_y)), _p(np.ascontiguousarray(src_uv)), qp, int(drop),
                                              _p(mbi), _p(rec_y), _p(rec_uv), _p(lev)), "stage_intra")
        return rec_y, rec_uv, mbi, lev

    def stage_intra_analyse(self, src_y, src_uv, qp=30):
        out = np.empty((self.mbw * self.mbh, 152), np.uint16)
        dec = np.zeros(self.mbw * self.mbh, IDEC)
        self._chk(self.L.mi355enc_stage_intra_analyse(self.h, _p(np.ascontiguousarray(src_y)), _p(np.ascontiguousarray(src_uv)), qp, _p(out), _p(dec)), "stage_intra_analyse")
        return out, dec

    def stage_deblock(self, rec_y, rec_uv, mbi):
        y, uv = np.ascontiguousarray(rec_y).copy(), np.ascontiguousarray(rec_uv).copy()
        self._chk(self.L.mi355enc_stage_deblock(self.h, _p(y), _p(uv), _p(np.ascontiguousarray(mbi))), "stage_deblock")
        return y, uv

    @property
    def slice_rows(self):
        """macroblock rows per slice of this encoder's I pictures (0: one slice)"""
        return int(self.L.mi355enc_slice_rows(self.h))

    @property
    def p_slice_rows(self):
        """... and of its P pictures"""
        return int(self.L.mi355enc_p_slice_rows(self.h))

    def stage_set_slice_deblock(self, idc):
        """the single-stage entry points: disable_deblocking_filter_idc of the picture's slices (0 or 2)"""
        self._chk(self.L.mi355enc_stage_set_slice_deblock(self.h, int(idc)), "stage_set_slice_deblock")

    def stage_set_slice_rows(self, rows):
        """the single-stage entry points treat the picture as slices of `rows` macroblock rows (0, the default: one slice)"""
        self._chk(self.L.mi355enc_stage_set_slice_rows(self.h, int(rows)), "stage_set_slice_rows")

    def debug_trip_wait(self, code):
        """fault injection: as if a bounded device-side wait had just run out (include/mi355enc.h)"""
        self._chk(self.L.mi355enc_debug_trip_wait(self.h, int(code)), "debug_trip_wait")

    def time_stage(self, stage, iters=20):
        ms = C.c_double(0)
        self._chk(self.L.mi355enc_time_stage(self.h, stage, iters, C.byref(ms)), "time_stage")
        return ms.value
