"""GPU: the element inside a real GStreamer graph, driven by the reference's own
pipeline_loader.c + encoder_control.c (oracle/_ref/ref_harness), like ceracoder's main()."""
import json
import os
import struct
import subprocess

import numpy as np
import pytest

from tests.test_boundary_cpu import HARNESS, ROOT, gst_env

pytestmark = pytest.mark.gpu


def read_records(path):
    data = open(path, "rb").read()
    out, o = [], 0
    while o < len(data):
        n, pts = struct.unpack_from("<IQ", data, o)
        out.append((pts, data[o + 12:o + 12 + n]))
        o += 12 + n
    return out


@pytest.mark.skipif(not os.path.exists(HARNESS), reason="oracle/_ref/ref_harness not shipped")
def test_pipeline_file_runs_and_stream_decodes(tmp_path, oracle):
    """pipeline/mi355x/es_test_pattern_360p30: videotestsrc -> mi355h264enc name=venc_bps -> appsink."""
    out = tmp_path / "out.bin"
    r = subprocess.run([HARNESS, os.path.join(ROOT, "pipeline", "mi355x", "es_test_pattern_360p30"), str(out)], env=gst_env(),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    summary = json.loads(r.stdout.splitlines()[-1])
    assert summary["samples"] == 90
    recs = read_records(str(out))
    pts = [p for p, _ in recs]
    assert pts == sorted(pts) and pts[1] - pts[0] == 33333333
    dec = oracle.Decoder()
    for i, (_, au) in enumerate(recs):
        assert au[:5] == (b"\x00\x00\x00\x01\x67" if i % 30 == 0 else b"\x00\x00\x00\x01\x41"), i  # SPS before IDR, else non-IDR slice
        y, uv = dec.decode(au)
    assert dec.size == (640, 360)
    assert 40 < float(y[:360, :640].mean()) < 200  # SMPTE bars, not garbage


@pytest.mark.skipif(not os.path.exists(HARNESS), reason="oracle/_ref/ref_harness not shipped")
def test_balancer_script_drives_bitrate_through_reference_encoder_control(tmp_path):
    """Golden setpoints of the reference's adaptive balancer (tests/golden/balancer_adaptive.txt, 15 s
    of control time) replayed through encoder_control_set_bitrate on the GLib main thread while the
    streaming thread encodes a live 30 fps source: the produced rate follows 6.0 -> 5.3 -> 6.0 Mbit/s."""
    pf = tmp_path / "pipe"
    pf.write_text("videotestsrc is-live=true num-buffers=450 pattern=snow ! video/x-raw,width=640,height=368,framerate=30/1,format=NV12 ! queue ! "
                  "mi355h264enc key-int-max=30 name=venc_bps ! appsink name=appsink sync=false\n")
    out = tmp_path / "out.bin"
    script = os.path.join(ROOT, "tests", "golden", "balancer_adaptive.txt")
    r = subprocess.run([HARNESS, str(pf), str(out), script], env=gst_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    summary = json.loads(r.stdout.splitlines()[-1])
    assert summary["samples"] == 450 and summary["setpoints_applied"] >= 30
    recs = read_records(str(out))
    sizes = np.array([len(a) for _, a in recs], float)
    rate = lambda a, b: sizes[a:b].sum() * 8 * 30 / (b - a)
    # control time == wall time == stream time (live source): 0-5 s at 6000k, 5.25-7.5 s at 5300k, then back up
    assert abs(rate(60, 150) - 6.0e6) / 6.0e6 < 0.10, rate(60, 150)
    assert abs(rate(165, 225) - 5.3e6) / 5.3e6 < 0.12, rate(165, 225)
    assert abs(rate(330, 450) - 6.0e6) / 6.0e6 < 0.10, rate(330, 450)


@pytest.mark.skipif(not os.path.exists(HARNESS), reason="oracle/_ref/ref_harness not shipped")
def test_ts_pipeline_demuxes_and_decodes(tmp_path, oracle):
    """pipeline/mi355x/ts_test_pattern_720p30: ... ! mi355h264enc ! mi355tsmux ! appsink -- the plugins-bad-free tail
    (SURVEY 8f N4).  The harness regroups the packets into 1316-byte datagrams like new_buf_cb; an independent
    demultiplexer and the independent decoder must recover every picture."""
    from tests.tsdemux import demux
    out = tmp_path / "out.bin"
    r = subprocess.run([HARNESS, os.path.join(ROOT, "pipeline", "mi355x", "ts_test_pattern_720p30"), str(out)], env=gst_env(),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    summary = json.loads(r.stdout.splitlines()[-1])
    assert summary["samples"] == 150
    recs = read_records(str(out))
    ts = b"".join(a for _, a in recs)
    assert summary["datagrams_1316"] == len(ts) // 1316
    d = demux(ts)
    assert len(d["pes"]) == 150 and len(d["pat"]) >= 5
    assert [p["pts"] for p in d["pes"]] == [90000 + (pts // 100000 * 9 + pts % 100000 * 9 // 100000) for pts, _ in recs]
    dec = oracle.Decoder()
    for i, p in enumerate(d["pes"]):
        data = bytes(p["data"])
        assert data[:6] == b"\x00\x00\x00\x01\x09\xF0" and p["rai"] == (i % 30 == 0)
        y, uv = dec.decode(data[6:])
    assert dec.size == (1280, 720)
    lat = summary["ms_encoder_sink_to_appsink"]
    assert lat["n"] >= 60 and 0 < lat["p50"] < 50, lat


@pytest.mark.skipif(not os.path.exists(HARNESS), reason="oracle/_ref/ref_harness not shipped")
@pytest.mark.parametrize("fmt", ["I420", "YUY2", "UYVY"])
def test_element_takes_raw_formats_without_videoconvert(tmp_path, oracle, fmt):
    """The reference's graphs put `videoconvert` in front of the encoder; with I420 / packed 4:2:2 accepted directly
    (device-side conversion) the same graph negotiates without a CPU conversion.  The stream must decode to the
    videotestsrc picture."""
    pf = tmp_path / "pipe"
    pf.write_text("videotestsrc num-buffers=12 ! video/x-raw,width=320,height=180,framerate=30/1,format=%s ! "
                  "mi355h264enc key-int-max=30 qp=24 name=venc_bps ! appsink name=appsink sync=false\n" % fmt)
    out = tmp_path / "out.bin"
    r = subprocess.run([HARNESS, str(pf), str(out)], env=gst_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    recs = read_records(str(out))
    assert len(recs) == 12
    dec = oracle.Decoder()
    for _, au in recs:
        y, uv = dec.decode(au)
    assert dec.size == (320, 180)
    # SMPTE bars: the left-most bar is white or light grey, the 7th blue (Y ~ 35); chroma of the blue bar: Cb high, Cr low
    assert float(y[20:100, 5:35].mean()) > 150 and float(y[20:100, 280:310].mean()) < 70
    assert float(uv[10:50, 280:310:2].mean()) > 170 and float(uv[10:50, 281:311:2].mean()) < 128


@pytest.mark.skipif(not os.path.exists(HARNESS), reason="oracle/_ref/ref_harness not shipped")
def test_dct8x8_and_i8x8_properties_give_a_high_profile_stream_that_decodes(tmp_path, oracle):
    """x264enc's `dct8x8` = High profile with the 8x8 transform; its Intra_8x8 half is the property `i8x8` here.  Both streams carry a High-profile SPS and
    go through the independent decoder; the I pictures differ (Intra_8x8 macroblocks) and are smaller with it."""
    streams = {}
    for i8 in ("false", "true"):
        pf = tmp_path / ("pipe_" + i8)
        pf.write_text("videotestsrc num-buffers=8 pattern=zone-plate kx2=12 ky2=12 kt=2 ! video/x-raw,width=640,height=368,framerate=30/1,format=NV12 ! "
                      "mi355h264enc key-int-max=4 qp=28 dct8x8=true i8x8=%s name=venc_bps ! appsink name=appsink sync=false\n" % i8)
        out = tmp_path / ("out_%s.bin" % i8)
        r = subprocess.run([HARNESS, str(pf), str(out)], env=gst_env(), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-3000:]
        recs = read_records(str(out))
        assert len(recs) == 8
        assert recs[0][1][:5] == b"\x00\x00\x00\x01\x67" and recs[0][1][5] == 100  # profile_idc of the SPS: High
        dec = oracle.Decoder()
        for _, au in recs:
            y, uv = dec.decode(au)
        assert dec.size == (640, 368)
        streams[i8] = [au for _, au in recs]
    assert streams["true"][0] != streams["false"][0]
    assert len(streams["true"][0]) < len(streams["false"][0])


@pytest.mark.skipif(not os.path.exists(HARNESS), reason="oracle/_ref/ref_harness not shipped")
@pytest.mark.parametrize("props,profile_idc,p_slices", [("speed-preset=3 key-int-max=60", 100, 2), ("speed-preset=2 key-int-max=60", 100, 2), ("speed-preset=1 key-int-max=60", 66, 2),
                                                         ("key-int-max=60", 66, 2), ("speed-preset=3 dct8x8=false key-int-max=60", 66, 2),
                                                         ("speed-preset=veryfast slices=1 slice-deblock=false key-int-max=60", 100, 1), ("speed-preset=2 slices=3 key-int-max=60", 100, 3)])
def test_speed_preset_and_slices_reach_the_stream(tmp_path, oracle, props, profile_idc, p_slices):
    """What the reference's files pass (`speed-preset=2` / `=3`: /root/reference/pipeline/generic/x264_superfast_camlink:5, x264_veryfast_camlink:5) lands on the High-profile
    toolset (SPS profile_idc 100), ultrafast / none on Constrained Baseline (66), an explicit `dct8x8=false` wins; P pictures are cut into slices by default (720p: two,
    disable_deblocking_filter_idc 2) unless told otherwise.  Every stream decodes with the independent decoder."""
    pf = tmp_path / "pipe"
    pf.write_text("videotestsrc num-buffers=6 pattern=zone-plate kx2=12 ky2=12 kt=2 ! video/x-raw,width=1280,height=720,framerate=30/1,format=NV12 ! "
                  "mi355h264enc %s qp=30 name=venc_kbps ! appsink name=appsink sync=false\n" % props)
    out = tmp_path / "out.bin"
    r = subprocess.run([HARNESS, str(pf), str(out)], env=gst_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    recs = read_records(str(out))
    assert len(recs) == 6
    assert recs[0][1][:5] == b"\x00\x00\x00\x01\x67" and recs[0][1][5] == profile_idc
    dec = oracle.Decoder()
    for i, (_, au) in enumerate(recs):
        dec.decode(au)
        if i:
            assert au.count(b"\x00\x00\x01\x41") == p_slices, (i, au.count(b"\x00\x00\x01\x41"))  # P slices: nal_ref_idc 2, type 1
    assert dec.size == (1280, 720)


@pytest.mark.skipif(not os.path.exists(HARNESS), reason="oracle/_ref/ref_harness not shipped")
def test_reference_x264_line_with_only_the_factory_token_changed_runs_at_the_written_rate(tmp_path):
    """The encoder hop of pipeline/mi355x/x264_superfast_camlink (the reference's file, `x264enc` -> `mi355h264enc`, still
    `name=venc_kbps`) between a test source and the appsink: the reference's encoder_control divides by 1000 for that name
    (encoder_control.c:29-32,53) and the element takes "bps" in kbit/s there, so 4.3 Mbit/s is what comes out."""
    from tests.test_boundary_cpu import encoder_line_of_reference_file
    pf = tmp_path / "pipe"
    pf.write_text("videotestsrc num-buffers=240 pattern=snow ! video/x-raw,width=640,height=368,framerate=30/1 ! videoconvert ! \n"
                  + encoder_line_of_reference_file() + " ! \nappsink name=appsink sync=false\n")
    script = tmp_path / "script"
    script.write_text("0 4300000\n")
    out = tmp_path / "out.bin"
    r = subprocess.run([HARNESS, str(pf), str(out), str(script)], env=gst_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    info = json.loads([l for l in r.stderr.splitlines() if l.startswith("{\"encoder_found\"")][0])
    assert info["bitrate_div"] == 1000 and info["bps_after_null_state_write"] == 4300 and info["bitrate_kbps"] == 4300
    sizes = np.array([len(a) for _, a in read_records(str(out))], float)
    rate = sizes[60:240].sum() * 8 * 30 / 180  # three GOPs of 60 after the first
    assert abs(rate - 4.3e6) / 4.3e6 < 0.10, rate


@pytest.mark.skipif(not os.path.exists(HARNESS), reason="oracle/_ref/ref_harness not shipped")
@pytest.mark.parametrize("fixup", [False, True])
@pytest.mark.parametrize("depth", [0, 1, 2])
def test_damaged_timestamps_leave_monotone_with_dts_equal_pts(tmp_path, oracle, fixup, depth):
    """upstream of the encoder the reference rewrites PTS, zeroes DTS and flags early pictures DROPPABLE
    (ceracoder.c:371-423).  With timestamps damaged the way a capture device damages them (jitter, repeats, a picture two
    periods early, garbage DTS) -- repaired by the reference's ptsfixup logic or not -- every sample leaves with DTS == PTS,
    PTS never runs backwards, DROPPABLE pictures are not coded, and the stream still decodes."""
    pf = tmp_path / "pipe"
    pf.write_text("videotestsrc num-buffers=120 pattern=ball ! video/x-raw,width=320,height=192,framerate=30/1,format=NV12 ! "
                  "identity name=jitter signal-handoffs=TRUE ! " + ("identity name=ptsfixup signal-handoffs=TRUE ! " if fixup else "") +
                  "queue ! mi355h264enc key-int-max=30 pipeline-depth=%d exclusive-gpu=%s name=venc_bps ! appsink name=appsink sync=false\n" % (depth, "true" if depth == 2 else "false"))
    out = tmp_path / "out.bin"
    r = subprocess.run([HARNESS, str(pf), str(out)], env=gst_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    s = json.loads(r.stdout.splitlines()[-1])
    assert s["dts_ne_pts"] == 0 and s["pts_backwards"] == 0, s
    assert s["samples"] == 120 - s["droppable_in"]
    if fixup:
        assert s["droppable_in"] >= 1 and s["pts_repeated"] == 0, s   # the early pictures were flagged upstream and not coded
    recs = read_records(str(out))
    pts = [p for p, _ in recs]
    assert pts == sorted(pts)
    dec = oracle.Decoder()
    for _, au in recs:
        dec.decode(au)
    assert dec.size == (320, 192)


@pytest.mark.skipif(not os.path.exists(HARNESS), reason="oracle/_ref/ref_harness not shipped")
def test_device_id_without_a_device_fails_through_the_bus(tmp_path):
    """`device-id` beyond the box's GPUs is MI355ENC_ERR_NO_DEVICE -> bus ERROR -> the harness's
    cb_pipeline equivalent stops with code 3 (ceracoder.c:425-438); nothing is encoded."""
    import torch
    pf = tmp_path / "pipe"
    pf.write_text("videotestsrc num-buffers=5 ! video/x-raw,width=320,height=192,framerate=30/1,format=NV12 ! "
                  "mi355h264enc device-id=%d name=venc_bps ! appsink name=appsink sync=false\n" % torch.cuda.device_count())
    r = subprocess.run([HARNESS, str(pf), str(tmp_path / "out.bin")], env=gst_env(), capture_output=True, text=True, timeout=120)
    assert r.returncode == 3 and "no usable HIP device" in r.stderr, r.stderr[-2000:]
    assert json.loads(r.stdout.splitlines()[-1])["samples"] == 0


@pytest.mark.skipif(not os.path.exists(HARNESS), reason="oracle/_ref/ref_harness not shipped")
def test_open_is_far_below_the_stall_watchdog_tick(tmp_path):
    """stall_check (ceracoder.c:152-200) stops the app when the position stands still for a 1 s tick; the element
    opens the device in set_format.  mi355enc_open() must take well under 500 ms, first use of the GPU in the process included."""
    pf = tmp_path / "pipe"
    pf.write_text("videotestsrc num-buffers=10 ! video/x-raw,width=1920,height=1080,framerate=60/1,format=NV12 ! "
                  "mi355h264enc stats=true name=venc_bps ! appsink name=appsink sync=false\n")
    r = subprocess.run([HARNESS, str(pf), str(tmp_path / "out.bin")], env=gst_env(), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    st = json.loads([l for l in r.stderr.splitlines() if l.startswith("{\"element\"")][0])
    assert st["frames"] == 10 and 0 < st["open_ms"] < 500, st
