"""The drop-in boundary exercised by the reference's own code (compiled unmodified into
oracle/_ref): pipeline text -> gst_parse_launch -> element found by name -> "bps" written in
state NULL (ceracoder.c:514-518).  Golden bitrate scripts come from the reference balancer."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HARNESS = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
GOLD = os.path.join(ROOT, "tests", "golden")


def gst_env():
    env = dict(os.environ)
    env.update(GST_PLUGIN_SYSTEM_PATH="/opt/conda/lib/gstreamer-1.0", GST_PLUGIN_SCANNER="/opt/conda/libexec/gstreamer-1.0/gst-plugin-scanner",
               GST_REGISTRY="/tmp/ceracoder_amd_gst_registry.bin", GST_PLUGIN_PATH=os.path.join(ROOT, "ceracoder_amd", "gst-plugins"),
               LD_PRELOAD="/usr/lib/x86_64-linux-gnu/libstdc++.so.6")
    return env


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


needs_gst = pytest.mark.skipif(not os.path.exists("/opt/conda/bin/gst-inspect-1.0"), reason="GStreamer 1.14 of this image not found")


def test_golden_balancer_scripts_match_survey():
    """SURVEY.md 8c lists what the reference's balancer prints for test_integration.c:151-225."""
    def kbps(name):
        return [int(l.split()[1]) // 1000 for l in open(os.path.join(GOLD, name))]
    assert kbps("balancer_adaptive.txt") == [6000] * 10 + [5300] * 10 + [5500, 5500, 5700, 5700, 5900, 5900] + [6000] * 9
    assert kbps("balancer_aimd.txt") == [6000] * 10 + [4500, 3300, 2500, 1800, 1400, 1000, 800, 600, 500, 500] + \
        [500, 500, 600, 600, 600, 600, 700, 700, 700, 700, 800, 800, 800, 800, 900]
    assert kbps("balancer_fixed.txt") == [6000] * 35
    ts = [int(l.split()[0]) for l in open(os.path.join(GOLD, "balancer_adaptive.txt"))]
    assert ts[:10] == list(range(500, 5001, 500)) and ts[10] == 5250 and ts[-1] == 15000


@pytest.mark.skipif(not os.path.isdir("/root/reference/src/core"), reason="reference tree only exists in the build container")
def test_golden_balancer_scripts_regenerate_identically():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    for algo in ("adaptive", "aimd", "fixed"):
        out = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "gen_balancer_script"), algo], capture_output=True, text=True, check=True).stdout
        assert out == open(os.path.join(GOLD, "balancer_%s.txt" % algo)).read()


@needs_gst
def test_plugin_registers_with_expected_properties():
    out = subprocess.run(["/opt/conda/bin/gst-inspect-1.0", "mi355h264enc"], env=gst_env(), capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    for needle in ("bps ", "bitrate ", "key-int-max", "device-id", "speed-preset", "pipeline-depth", "exclusive-gpu", "video/x-h264", "byte-stream", "NV12"):
        assert needle in out.stdout, needle


@needs_gst
@pytest.mark.skipif(not os.path.exists(HARNESS), reason="oracle/_ref not built")
@pytest.mark.parametrize("name,div", [("venc_bps", 1), ("venc_kbps", 1000)])
def test_reference_encoder_control_drives_bps_in_null_state(tmp_path, name, div):
    """encoder_control.c finds the element by name and writes "bps" before PLAYING; the element must
    hold the value.  Named venc_kbps (as in every x264 pipeline file of the reference) the reference writes
    bitrate / 1000 into "bps": the element then takes "bps" in kbit/s, so both names end at 4.3 Mbit/s."""
    pf = tmp_path / "pipe"
    pf.write_text("videotestsrc num-buffers=3 ! video/x-raw,width=320,height=192,framerate=30/1,format=NV12 ! "
                  "mi355h264enc key-int-max=30 speed-preset=superfast name=%s ! appsink name=appsink sync=false\n" % name)
    script = tmp_path / "script"
    script.write_text("0 4300000\n")
    r = subprocess.run([HARNESS, str(pf), str(tmp_path / "out.bin"), str(script)], env=gst_env(), capture_output=True, text=True, timeout=120)
    info = json.loads([l for l in r.stderr.splitlines() if l.startswith("{\"encoder_found\"")][0])
    assert info == {"encoder_found": 1, "bitrate_div": div, "bps_after_null_state_write": 4300000 // div, "bitrate_kbps": 4300}
    if _has_gpu():
        assert r.returncode == 0, r.stderr
        assert json.loads(r.stdout.splitlines()[-1])["samples"] == 3
    else:  # no device: a bus ERROR from the element, never a silent CPU encode
        assert r.returncode == 3 and "no usable HIP device" in r.stderr, r.stderr


@needs_gst
@pytest.mark.skipif(not os.path.exists(HARNESS), reason="oracle/_ref not built")
@pytest.mark.parametrize("line,bps,kbps", [("mi355h264enc bps=6000 name=venc_kbps", 6000, 6000), ("mi355h264enc name=venc_kbps bps=6000", 6000, 6000),
                                           ("mi355h264enc bps=5000000 name=venc_bps", 5000000, 5000), ("mi355h264enc bitrate=3500 name=venc_kbps", 3500, 3500),
                                           ("mi355h264enc bitrate=3500 name=venc_bps", 3500000, 3500)])
def test_bps_unit_is_resolved_when_used_not_when_written(tmp_path, line, bps, kbps):
    """gst_parse_launch sets properties in text order: with `bps=6000 name=venc_kbps` the value arrives before the element has the name that
    gives it its unit (encoder_control.c:29-36).  The element keeps what was written and applies the unit where the target is consumed."""
    pf = tmp_path / "pipe"
    pf.write_text("videotestsrc num-buffers=1 ! video/x-raw,width=320,height=192,framerate=30/1,format=NV12 ! %s ! appsink name=appsink sync=false\n" % line)
    script = tmp_path / "script"
    script.write_text("0 4300000\n")
    r = subprocess.run([HARNESS, str(pf), str(tmp_path / "out.bin"), str(script)], env=gst_env(), capture_output=True, text=True, timeout=120)
    info = json.loads([l for l in r.stderr.splitlines() if l.startswith("{\"bps_from_pipeline_text\"")][0])
    assert info == {"bps_from_pipeline_text": bps, "bitrate_kbps_from_pipeline_text": kbps}


PROBE = os.path.join(ROOT, "ceracoder_amd", "mi355_gst_probe")


@needs_gst
@pytest.mark.skipif(not os.path.exists(PROBE), reason="probe not built")
@pytest.mark.parametrize("line,want", [
    ("mi355h264enc", dict(speed_preset=0, dct8x8=0, i8x8=0, aq_mode=0, intra_in_p=1, slices=0, slice_deblock=1, has_partitions_property=0)),
    ("mi355h264enc speed-preset=1", dict(dct8x8=0, i8x8=0, aq_mode=0, intra_in_p=1)),
    ("mi355h264enc speed-preset=2 key-int-max=60", dict(speed_preset=2, dct8x8=1, i8x8=1, aq_mode=1, intra_in_p=1)),  # the reference's line, /root/reference/pipeline/generic/x264_superfast_camlink:5
    ("mi355h264enc speed-preset=3 key-int-max=60", dict(speed_preset=3, dct8x8=1, i8x8=1, aq_mode=1, intra_in_p=1)),                # .../x264_veryfast_camlink:5
    ("mi355h264enc speed-preset=veryfast", dict(speed_preset=3, dct8x8=1)),
    ("mi355h264enc speed-preset=medium intra-in-p=2", dict(speed_preset=6, dct8x8=1, i8x8=1, aq_mode=1, intra_in_p=2)),
    ("mi355h264enc speed-preset=2 aq-mode=0 dct8x8=false", dict(dct8x8=0, i8x8=0, aq_mode=0)),            # explicit properties win, in either order
    ("mi355h264enc dct8x8=true speed-preset=1", dict(dct8x8=1, i8x8=0, aq_mode=0)),
    ("mi355h264enc speed-preset=2 i8x8=false slices=4 slice-deblock=false intra-slices=2", dict(dct8x8=1, i8x8=0, slices=4, slice_deblock=0, intra_slices=2)),
])
def test_speed_preset_selects_a_toolset_and_explicit_properties_win(line, want):
    """The reference's pipeline files pass x264enc's speed-preset (=2 superfast, =3 veryfast: /root/reference/pipeline/generic/x264_superfast_camlink:5,
    x264_veryfast_camlink:5; /root/reference/bindings/typescript/src/pipeline/generic-builder.ts:43-55): the element maps it onto the tools it has instead of ignoring it.
    Read back through GObject as the encoder will use them; no device involved."""
    r = subprocess.run([PROBE, "videotestsrc ! %s name=venc_kbps ! appsink name=appsink" % line, "--props"], env=gst_env(), capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    got = json.loads(r.stdout.splitlines()[-1])
    for k, v in want.items():
        assert got[k] == v, (k, got)


def test_pipeline_files_name_the_element_like_the_reference():
    """The swap point is one token: same line shape as pipeline/generic/x264_superfast_*:5-6."""
    d = os.path.join(ROOT, "pipeline", "mi355x")
    files = sorted(os.listdir(d))
    assert "h264_test_pattern_1080p60" in files and "h264_test_pattern_2160p60" in files and "h264_test_pattern_720p30" in files
    for f in files:
        text = open(os.path.join(d, f)).read()
        assert "mi355h264enc" in text and ("name=venc_bps" in text or "name=venc_kbps" in text) and "appsink name=appsink" in text, f
        assert "x264enc" not in text


@pytest.mark.skipif(not os.path.isdir("/root/reference/pipeline/generic"), reason="reference tree only exists in the build container")
def test_reference_pipeline_file_converts_by_changing_one_token():
    """pipeline/mi355x/x264_superfast_camlink is /root/reference/pipeline/generic/x264_superfast_camlink with the factory token
    changed and nothing else: same properties (speed-preset=2 key-int-max=60), same element name (venc_kbps)."""
    ref = open("/root/reference/pipeline/generic/x264_superfast_camlink").read().split()
    ours = open(os.path.join(ROOT, "pipeline", "mi355x", "x264_superfast_camlink")).read().split()
    assert len(ref) == len(ours)
    diff = [(a, b) for a, b in zip(ref, ours) if a != b]
    assert diff == [("x264enc", "mi355h264enc")]


def encoder_line_of_reference_file():
    """The encoder hop of the converted reference file, verbatim: `mi355h264enc speed-preset=2 key-int-max=60 name=venc_kbps`."""
    text = open(os.path.join(ROOT, "pipeline", "mi355x", "x264_superfast_camlink")).read()
    line = [l for l in text.splitlines() if l.startswith("mi355h264enc")][0]
    assert line.rstrip(" !") == "mi355h264enc speed-preset=2 key-int-max=60 name=venc_kbps"
    return line.rstrip(" !")


@needs_gst
def test_latency_probe_runs_the_graph_and_fails_loudly_without_a_device():
    """ceracoder_amd/mi355_gst_probe (bench.py's M2 leg): parses a description, needs the two named elements, and on a
    machine without a HIP device ends with the element's bus ERROR instead of producing samples."""
    probe = os.path.join(ROOT, "ceracoder_amd", "mi355_gst_probe")
    assert os.path.exists(probe), "run __graft_entry__.build()"
    r = subprocess.run([probe, "videotestsrc num-buffers=2 ! fakesink"], env=gst_env(), capture_output=True, text=True, timeout=60)
    assert r.returncode == 2 and "venc_bps" in r.stderr
    desc = ("videotestsrc num-buffers=3 ! video/x-raw,width=320,height=192,framerate=30/1,format=I420 ! mi355h264enc name=venc_bps ! "
            "mi355tsmux ! appsink name=appsink sync=false")
    r = subprocess.run([probe, desc], env=gst_env(), capture_output=True, text=True, timeout=120)
    if _has_gpu():
        assert r.returncode == 0 and json.loads(r.stdout.splitlines()[-1])["samples"] == 3, r.stderr
    else:
        assert r.returncode == 3 and "no usable HIP device" in r.stderr, r.stderr


@needs_gst
def test_probe_feeds_pictures_through_appsrc():
    """bench.py's through-the-element throughput leg with the probe as the source (--appsrc): every picture pushed arrives at the sink, and
    the rate over the pictures after the first GOP is reported."""
    probe = os.path.join(ROOT, "ceracoder_amd", "mi355_gst_probe")
    desc = "appsrc name=src ! video/x-raw,width=320,height=192,framerate=60/1,format=NV12 ! queue ! appsink name=appsink sync=false"
    r = subprocess.run([probe, desc, "--appsrc", "200", "320", "192", "--no-encoder"], env=gst_env(), capture_output=True, text=True, timeout=60)
    out = json.loads(r.stdout.splitlines()[-1])
    assert r.returncode == 0 and out["samples"] == 200 and out["buffers_timed"] == 139 and out["fps_after_first_gop"] > 0


def test_committed_bench_line_keeps_the_contract():
    """profiles/r03_bench_1080p_ippp.json is the line bench.py printed on the GPU box: the keys the driver reads, the roofline
    object (algorithmic bytes / live launch time, agreeing with the rocprofv3 kernel trace and the PMC pass), the CPU baseline, and
    (r03) the host-input and through-the-element rates beside the HBM-resident one."""
    import json
    d = json.load(open(os.path.join(ROOT, "profiles", "r03_bench_1080p_ippp.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in d, k
    assert d["unit"] == "frames/s" and d["n_gpus"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None and d["dtype"] == "u8"
    assert d["config"]["workload"] == "1080p_ippp" and "model" not in d["config"]
    assert abs(d["value"] * d["ms_per_step"] / 1e3 - 1.0) < 0.02  # value = pictures / time
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_us"] * 1e-6) / 1e9) < 0.05
    assert r["algorithmic_bytes_per_launch"] == int(3.0625 * 1920 * 1088)             # deblocking, 3.0625 P
    assert r["traffic"] and r["traffic"] >= r["algorithmic_bytes_per_launch"]          # PMC bytes, at least the algorithmic ones
    assert abs(r["kernel_trace_avg_us"] - r["avg_launch_us"]) / r["avg_launch_us"] < 0.15  # HIP events (prep + band kernel) vs kernel trace (band kernel)
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "frames/s" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    assert any(k["kernel"] == "me_kernel" for k in d["roofline_kernels"])              # the kernel north_star names is listed
    assert all(k["traffic"] for k in d["roofline_kernels"])                            # every listed kernel has its PMC traffic attached
    assert d["idr_in_timed_region"] == d["steps"] // d["config"]["gop"] and d["config"]["h2d_in_timed_region"] is False  # whole GOPs in the timed region ...
    assert abs(d["gop_weighted_frames_per_s"] - d["value"]) < 0.01 * d["value"]          # ... so the GOP-weighted rate is the headline itself
    assert d["psnr_db"]["pictures"] >= 60 and d["psnr_db"]["y"] > 30.0                 # mean over all pictures of two GOPs at the 6 Mbit/s setpoint
    for k in ("host_input_depth2_frames_per_s", "host_pinned_input_depth2_frames_per_s", "gst_frames_per_s", "gst_appsrc_frames_per_s"):
        assert d[k] and 0 < d[k] < 1.05 * d["value"], k                                # PCIe / the element in the loop: never more than the HBM-resident rate
    assert d["device_wait_recoveries"] == 0 and d["safe_level"] == 0
    for wl in ("2160p_ippp", "1080p_intra"):                                           # configs[3] and [1]: PMC traffic attached as well
        o = json.load(open(os.path.join(ROOT, "profiles", "r03_bench_%s.json" % wl)))
        assert o["roofline"]["traffic"] and o["config"]["workload"] == wl
