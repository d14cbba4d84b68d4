"""mi355ts: the minimal MPEG-TS muxer behind include/mi355ts.h, checked with an
independent demultiplexer (tests/tsdemux.py).  Host-only: runs without a GPU."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from ceracoder_amd import enc as E
from tests.tsdemux import crc32_mpeg2, demux

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
AUD = b"\x00\x00\x00\x01\x09\xF0"


class Mux:
    def __init__(self):
        self.L = C.CDLL(E.LIB_PATH)
        self.L.mi355ts_open.restype = C.c_void_p
        self.L.mi355ts_close.argtypes = [C.c_void_p]
        self.L.mi355ts_bound.restype = C.c_size_t
        self.L.mi355ts_bound.argtypes = [C.c_size_t]
        self.L.mi355ts_mux.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_int64, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        self.L.mi355ts_crc32.restype = C.c_uint32
        self.L.mi355ts_crc32.argtypes = [C.c_char_p, C.c_size_t]
        self.h = self.L.mi355ts_open()
        assert self.h

    def mux(self, au, pts_ns, key, cap=None):
        cap = self.L.mi355ts_bound(len(au)) if cap is None else cap
        out = (C.c_uint8 * max(cap, 1))()
        n = C.c_size_t()
        rc = self.L.mi355ts_mux(self.h, au, len(au), pts_ns, int(key), out, cap, C.byref(n))
        return rc, bytes(out[:n.value])

    def close(self):
        self.L.mi355ts_close(self.h)


def fake_au(rng, n, key):
    body = bytes(rng.integers(1, 256, max(0, n - 5), dtype=np.uint8))  # no zero bytes: no accidental start codes
    return b"\x00\x00\x00\x01" + (b"\x65" if key else b"\x41") + body


def test_header_symbols_exported():
    hdr = open(os.path.join(ROOT, "include", "mi355ts.h")).read()
    declared = sorted(set(re.findall(r"\b(mi355ts_[a-z0-9_]+)\s*\(", hdr)))
    assert declared == ["mi355ts_bound", "mi355ts_close", "mi355ts_crc32", "mi355ts_mux", "mi355ts_open"]
    L = C.CDLL(E.LIB_PATH)
    assert all(hasattr(L, n) for n in declared)


def test_crc32_known_answer():
    m = Mux()
    assert m.L.mi355ts_crc32(b"123456789", 9) == 0x0376E6E7  # CRC-32/MPEG-2 check value
    assert crc32_mpeg2(b"123456789") == 0x0376E6E7
    m.close()


def test_every_small_length_round_trips():
    """Lengths around the 176/184-byte payload boundaries exercise every stuffing case."""
    rng = np.random.default_rng(1)
    m = Mux()
    ts, aus = b"", []
    for i, n in enumerate(list(range(5, 420)) + [5000, 65536, 70001]):
        au = fake_au(rng, n, i % 60 == 0)
        rc, out = m.mux(au, i * 16_666_667, i % 60 == 0)
        assert rc == 0 and len(out) % 188 == 0 and len(out) <= m.L.mi355ts_bound(len(au))
        ts += out
        aus.append(au)
    d = demux(ts)
    assert len(d["pes"]) == len(aus)
    for p, au in zip(d["pes"], aus):
        assert bytes(p["data"]) == AUD + au
    m.close()


def test_tables_timing_and_flags():
    rng = np.random.default_rng(2)
    m = Mux()
    ts, keys = b"", []
    for i in range(180):  # 3 s at 60 fps, key frame every 60
        key = i % 60 == 0
        rc, out = m.mux(fake_au(rng, 3000 if key else 400, key), i * 1_000_000_000 // 60, key)
        assert rc == 0
        ts += out
        keys.append(key)
    d = demux(ts)
    assert d["pat"][0] == {"program": 1, "pmt_pid": 0x1000, "tsid": 1}
    assert d["pmt"][0] == {"pcr_pid": 0x100, "streams": [(0x1B, 0x100)]}
    assert d["video_pid"] == 0x100
    # PAT+PMT directly before every key frame, and at least every 100 ms in between
    pes_pos = {idx: k for k, (kind, idx) in enumerate(d["order"]) if kind == "pes"}
    last_psi_pts = None
    for i, p in enumerate(d["pes"]):
        k = pes_pos[i]
        preceded = k >= 2 and d["order"][k - 2][0] == "pat" and d["order"][k - 1][0] == "pmt"
        if keys[i]:
            assert preceded and p["rai"]
        else:
            assert not p["rai"]
        if preceded:
            last_psi_pts = p["pts"]
        assert p["pts"] - last_psi_pts < 9000 + 1500
    pts = [p["pts"] for p in d["pes"]]
    assert pts == [90000 + (i * 1_000_000_000 // 60) * 9 // 100000 for i in range(180)]  # 90 kHz, one second base offset
    for p in d["pes"]:
        assert p["pcr"] % 300 == 0 and p["pts"] - p["pcr"] // 300 == 11250       # PTS leads PCR by 125 ms
    m.close()


def test_existing_aud_is_not_duplicated_and_errors():
    m = Mux()
    au = AUD + b"\x00\x00\x00\x01\x41\x9a\x22"
    rc, out = m.mux(au, 0, False)
    assert rc == 0 and bytes(demux(out)["pes"][0]["data"]) == au
    assert m.mux(b"", 0, False)[0] == -1
    assert m.mux(au, -5, False)[0] == -1
    assert m.mux(au, 0, False, cap=188)[0] == -2
    m.close()


def test_real_access_units_survive_mux_demux_and_decode(oracle=None):
    """Oracle encoder -> muxer -> independent demux -> independent decoder == encoder reconstruction."""
    from oracle import oracle as O
    from tests.util import frames
    w, h = 176, 144
    oe, dec, m = O.Encoder(w, h, gop=4, threads=4), O.Decoder(), Mux()
    ts, recs = b"", []
    for i, (_, _, y, uv) in enumerate(frames(w, h, 6)):
        au, key = oe.encode(y, uv, 30)
        rc, out = m.mux(au, i * 33_333_333, key)
        assert rc == 0
        ts += out
        recs.append((oe.recon_y.copy(), oe.recon_uv.copy()))
    for p, (ry, ruv) in zip(demux(ts)["pes"], recs):
        data = bytes(p["data"])
        assert data.startswith(AUD)
        dy, duv = dec.decode(data[len(AUD):])
        assert np.array_equal(dy, ry) and np.array_equal(duv, ruv)
    m.close()


@pytest.mark.skipif(not os.path.exists("/opt/conda/bin/gst-launch-1.0"), reason="GStreamer 1.14 of this image not found")
def test_gst_element_wraps_a_buffer(tmp_path):
    from tests.test_boundary_cpu import gst_env
    au = b"\x00\x00\x00\x01\x65" + bytes(range(1, 250)) * 3
    (tmp_path / "in.h264").write_bytes(au)
    r = subprocess.run(["/opt/conda/bin/gst-launch-1.0", "-q", "filesrc", "location=%s" % (tmp_path / "in.h264"), "blocksize=%d" % len(au), "!",
                        "video/x-h264,stream-format=byte-stream,alignment=au", "!", "mi355tsmux", "!", "filesink", "location=%s" % (tmp_path / "out.ts")],
                       env=gst_env(), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    d = demux((tmp_path / "out.ts").read_bytes())
    assert len(d["pat"]) == 1 and len(d["pes"]) == 1 and bytes(d["pes"][0]["data"]) == AUD + au
