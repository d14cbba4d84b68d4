#!/usr/bin/env python3
"""bench.py -- encoded frames/s of the MI355X H.264 hot path (BASELINE.json metric).

A "step" is one picture through the whole hot path: motion search / intra prediction,
transform+quant, reconstruction, in-loop deblocking on the GPU, CAVLC on the host thread.
Source pictures (synthetic S2) are resident in HBM before the timed region
and handed over by device pointer (mi355enc_submit_device).  One independent stream per
GPU (north_star: one live stream does not shard; no RCCL on the data path) -- ranks only
meet at the barriers that bracket the timed region.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload 1080p_ippp|1080p_intra|2160p_ippp|720p_ippp]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (width, height, fps, gop, bitrate_bps)
    "1080p_ippp": (1920, 1080, 60, 60, 6_000_000),   # BASELINE.json configs[2] (headline: IPPP + full SAD search)
    "1080p_intra": (1920, 1080, 60, 1, 6_000_000),   # configs[1]
    "2160p_ippp": (3840, 2160, 60, 60, 20_000_000),  # configs[3]
    "720p_ippp": (1280, 720, 30, 60, 6_000_000),     # configs[0] geometry
}
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md "HBM3E peak BW" (spec)


def coded(v):
    return (v + 15) // 16 * 16


def make_source(width, height, n_unique):
    from ceracoder_amd import synth
    frames = np.empty((n_unique, height * 3 // 2, width), np.uint8)
    for i, (y, uv) in enumerate(synth.s2_frames(width, height, n_unique)):
        frames[i, :height] = y
        frames[i, height:] = uv
    return frames


def bounce(i, n):
    """0,1,..,n-1,n-2,..,1,0,1,.. -- plays the clip back and forth so motion stays continuous."""
    if n == 1:
        return 0
    p = i % (2 * n - 2)
    return p if p < n else 2 * n - 2 - p


def cpu_baseline(frames_np, width, height, fps, gop, qps, budget_s=20.0):
    """The oracle (scalar CPU restatement, OpenMP over macroblock rows in the motion search
    only) on a bounded sample of the same pictures with the QPs the GPU run chose."""
    from oracle import oracle as O
    threads = max(1, min(os.cpu_count() or 1, 16))
    enc = O.Encoder(width, height, fps=fps, gop=gop, threads=threads)
    n, t0 = 0, time.perf_counter()
    while n < len(qps) and (n < 4 or time.perf_counter() - t0 < budget_s):
        f = frames_np[bounce(n, len(frames_np))]
        enc.encode(f[:height], f[height:], int(qps[n]))
        n += 1
    dt = time.perf_counter() - t0
    enc.close()
    return {"value": round(n / dt, 3), "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": "first %d pictures of the same workload, oracle/h264_enc_oracle.c (own CPU restatement, not x264; "
                      "x264enc is not installed on this image), OpenMP motion search on %d threads, rest scalar" % (n, threads)}


def gst_latency(width, height, fps, gop, bps, dev, seconds=4):
    """M2 as SURVEY 8d defines it: a LIVE source at the nominal frame rate through the element in a real GStreamer graph,
    stamped at the encoder's sink pad, at the appsink callback and after the last 1316-byte datagram (UDP loopback; libsrt
    is not in the image).  Uses the product's own probe (ceracoder_amd/mi355_gst_probe), nothing of oracle/."""
    import subprocess
    probe = os.path.join(ROOT, "ceracoder_amd", "mi355_gst_probe")
    if not (os.path.exists(probe) and os.path.exists("/opt/conda/lib/gstreamer-1.0")):
        return {"unavailable": "GStreamer 1.14 of this image or the probe binary not found"}
    env = dict(os.environ)
    env.update(GST_PLUGIN_SYSTEM_PATH="/opt/conda/lib/gstreamer-1.0", GST_PLUGIN_SCANNER="/opt/conda/libexec/gstreamer-1.0/gst-plugin-scanner",
               GST_REGISTRY="/tmp/ceracoder_amd_gst_registry_bench.bin", GST_PLUGIN_PATH=os.path.join(ROOT, "ceracoder_amd", "gst-plugins"),
               LD_PRELOAD="/usr/lib/x86_64-linux-gnu/libstdc++.so.6")
    desc = ("videotestsrc is-live=true num-buffers=%d pattern=smpte horizontal-speed=5 ! video/x-raw,width=%d,height=%d,framerate=%d/1,format=NV12 ! queue ! "
            "mi355h264enc key-int-max=%d bps=%d device-id=%d exclusive-gpu=true name=venc_bps ! mi355tsmux ! appsink name=appsink sync=false"
            % (seconds * fps, width, height, fps, gop, bps, dev))
    try:
        r = subprocess.run([probe, desc], env=env, capture_output=True, text=True, timeout=60 + 3 * seconds)
        out = json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as e:  # the headline number must not depend on this extra
        return {"unavailable": "probe failed: %s" % e}
    out["path"] = ("live videotestsrc (scrolling SMPTE bars) %dx%d@%d -> mi355h264enc -> mi355tsmux -> appsink -> 7x188-byte datagrams over UDP loopback "
                   "(same packetisation as ceracoder.c new_buf_cb; not SRT)" % (width, height, fps))
    return out


def gst_throughput(width, height, fps, gop, bps, dev, depth, buffers=660, clip=None):
    """M1 through the element (SURVEY 8d "Timing method"): a NON-live source, the element instantiated by gst_parse_launch the way
    /root/reference/src/io/pipeline_loader.c:59 does, wall-clock between buffers arriving at the sink behind it, first GOP discarded.
    The source is videotestsrc (what the reference's own test pipelines use); its own ceiling -- the same source into the same sink
    without the encoder -- is measured beside it, because at 1080p it paints slower than the encoder codes."""
    import subprocess
    probe = os.path.join(ROOT, "ceracoder_amd", "mi355_gst_probe")
    if not (os.path.exists(probe) and os.path.exists("/opt/conda/lib/gstreamer-1.0")):
        return {"unavailable": "GStreamer 1.14 of this image or the probe binary not found"}
    env = dict(os.environ)
    env.update(GST_PLUGIN_SYSTEM_PATH="/opt/conda/lib/gstreamer-1.0", GST_PLUGIN_SCANNER="/opt/conda/libexec/gstreamer-1.0/gst-plugin-scanner",
               GST_REGISTRY="/tmp/ceracoder_amd_gst_registry_bench.bin", GST_PLUGIN_PATH=os.path.join(ROOT, "ceracoder_amd", "gst-plugins"),
               LD_PRELOAD="/usr/lib/x86_64-linux-gnu/libstdc++.so.6")
    src = "videotestsrc is-live=false num-buffers=%d pattern=smpte horizontal-speed=5 ! video/x-raw,width=%d,height=%d,framerate=%d/1,format=NV12" % (buffers, width, height, fps)
    encoder = "mi355h264enc key-int-max=%d bps=%d device-id=%d pipeline-depth=%d exclusive-gpu=true name=venc_bps" % (gop, bps, dev, depth)
    out = {}
    for key, desc, extra in (("element", "%s ! queue ! %s ! appsink name=appsink sync=false" % (src, encoder), []),
                             ("element_pageable_input", "%s ! queue ! %s pinned-input=false ! appsink name=appsink sync=false" % (src, encoder), []),
                             ("source_alone", "%s ! queue ! appsink name=appsink sync=false" % src, ["--no-encoder"])):
        try:
            r = subprocess.run([probe, desc] + extra, env=env, capture_output=True, text=True, timeout=180)
            j = json.loads(r.stdout.strip().splitlines()[-1])
            out[key] = {"frames_per_s": j.get("fps_after_first_gop"), "buffers_timed": j.get("buffers_timed"), "samples": j.get("samples")}
        except Exception as e:
            out[key] = {"unavailable": "probe failed: %s" % e}
    # ... and with the probe feeding pre-rendered pictures through appsrc (no copy, no painting): what the element itself sustains
    asrc = "appsrc name=src ! video/x-raw,width=%d,height=%d,framerate=%d/1,format=NV12" % (width & ~3, height, fps)
    clip_args = []
    if clip is not None and (width & 3) == 0:  # the same pictures the C-ABI legs code (the first 16 of the clip, walked forwards and backwards), as one raw NV12 file
        path = "/tmp/mi355_bench_clip_%dx%d.nv12" % (width, height)
        with open(path, "wb") as f:
            f.write(np.ascontiguousarray(clip[:16]).tobytes())  # (n, height * 3 / 2, width): NV12 pictures one after the other
        clip_args = ["--clip", path]
    # (each twice, the better run reported and both listed: a 0.4 s run now and then lands on a box hiccup -- one r03 run measured 541 where every other gave 3700-3850)
    for key, extra in (("element_appsrc_pageable", []), ("element_appsrc_pinned", ["pinned"])):
        try:
            runs = []
            for _ in range(2):
                r = subprocess.run([probe, "%s ! queue ! %s ! appsink name=appsink sync=false" % (asrc, encoder), "--appsrc", str(2 * buffers), str(width), str(height)] + extra + clip_args,
                                   env=env, capture_output=True, text=True, timeout=180)
                runs.append(json.loads(r.stdout.strip().splitlines()[-1]))
            j = max(runs, key=lambda x: x.get("fps_after_first_gop") or 0.0)
            out[key] = {"frames_per_s": j.get("fps_after_first_gop"), "buffers_timed": j.get("buffers_timed"), "samples": j.get("samples"), "runs": [x.get("fps_after_first_gop") for x in runs],
                        "content": "the bench's own clip (first 16 pictures, forwards and backwards)" if clip_args else "the probe's panning texture"}
        except Exception as e:
            out[key] = {"unavailable": "probe failed: %s" % e}
    out["pipeline"] = "%s ! queue ! %s ! appsink sync=false (non-live; wall-clock between buffers at the sink, first 60 discarded)" % (src, encoder)
    return out


def third_party_probe():
    """x264enc / an H.264 decoder / ffmpeg on the box would allow an x264 baseline and a third-party decode of
    our stream.  None of them is part of this image; record everything that was looked for instead of failing: programs,
    GStreamer elements, shared libraries (software and hardware decoders: libavcodec, openh264, rocDecode, VA-API, AMF) and
    the render nodes a VA-API / rocDecode decoder would need."""
    import ctypes.util
    import glob
    import shutil
    import subprocess
    found = {p: bool(shutil.which(p)) for p in ("ffmpeg", "ffprobe", "x264", "vainfo", "gst-launch-1.0")}
    insp = "/opt/conda/bin/gst-inspect-1.0" if os.path.exists("/opt/conda/bin/gst-inspect-1.0") else shutil.which("gst-inspect-1.0")
    for el in ("x264enc", "avdec_h264", "openh264dec", "vaapih264dec", "vah264dec", "h264parse", "mpegtsmux"):
        ok = False
        if insp:
            try:
                env = dict(os.environ, GST_PLUGIN_SYSTEM_PATH="/opt/conda/lib/gstreamer-1.0", GST_REGISTRY="/tmp/ceracoder_amd_gst_registry_bench.bin",
                           GST_PLUGIN_SCANNER="/opt/conda/libexec/gstreamer-1.0/gst-plugin-scanner")
                ok = subprocess.run([insp, "--exists", el], env=env, capture_output=True, timeout=30).returncode == 0
            except Exception:
                ok = False
        found[el] = ok
    libs = {}
    for lib in ("avcodec", "x264", "openh264", "rocdecode", "va", "va-drm", "amfrt64"):
        hit = ctypes.util.find_library(lib) or next(iter(glob.glob("/opt/rocm/lib/lib%s.so*" % lib) + glob.glob("/opt/conda/lib/lib%s.so*" % lib)), None)
        libs[lib] = hit or False
    found["libraries"] = libs
    found["render_nodes"] = sorted(glob.glob("/dev/dri/renderD*"))
    found["decoder_available"] = bool(found["ffmpeg"] or found["avdec_h264"] or found["openh264dec"] or found["vaapih264dec"] or found["vah264dec"] or
                                      libs["avcodec"] or libs["openh264"] or libs["rocdecode"])
    found["note"] = ("x264enc absent: cpu_baseline is this repo's own CPU restatement, not x264" if not found["x264enc"] else "x264enc present")
    return found


def third_party_decode(aus, width, height):
    """Decode the access units with ffmpeg when the box has one: returns the decoded NV12 frames (list of (y, uv)) or None.
    The only third-party decoder path this repository can drive without linking anything; never available on this image."""
    import shutil
    import subprocess
    if not shutil.which("ffmpeg"):
        return None
    r = subprocess.run(["ffmpeg", "-v", "error", "-f", "h264", "-i", "pipe:0", "-f", "rawvideo", "-pix_fmt", "nv12", "pipe:1"], input=b"".join(aus),
                       capture_output=True, timeout=600)
    if r.returncode:
        raise RuntimeError("ffmpeg failed to decode the stream: " + r.stderr.decode(errors="replace")[-500:])
    fsz = width * height * 3 // 2
    raw = np.frombuffer(r.stdout, np.uint8)
    return [(raw[i * fsz:i * fsz + width * height].reshape(height, width), raw[i * fsz + width * height:(i + 1) * fsz].reshape(height // 2, width))
            for i in range(len(raw) // fsz)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=600)
    ap.add_argument("--warmup", type=int, default=60)
    ap.add_argument("--workload", default="1080p_ippp", choices=sorted(WORKLOADS))
    ap.add_argument("--unique", type=int, default=32, help="distinct synthetic pictures kept in HBM")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gst-latency", action="store_true", help="skip the live GStreamer latency probe (M2)")
    ap.add_argument("--no-extras", action="store_true", help="skip the untimed host-input / latency / IDR-probe legs as well (for kernel traces: the trace then holds the timed "
                    "region's schedule, its warm-up and the quality pass only)")
    ap.add_argument("--no-graphs", action="store_true")
    ap.add_argument("--fixed-qp", type=int, default=-1)
    ap.add_argument("--depth", type=int, default=2, help="pipeline_depth: pictures in flight - 1 (2: the device never waits for the host)")
    ap.add_argument("--deblock-mode", type=int, default=0)
    ap.add_argument("--shared-gpu", action="store_true", help="other processes encode on the same GPU: do not set exclusive_device (include/mi355enc.h); the default is the metric's one stream per GPU")
    ap.add_argument("--sample-in-order", action="store_true", help="sampled P pictures run their stages strictly in order (every timer is one kernel alone; such a picture costs "
                    "the stream about two periods) instead of keeping the free-running schedule (timers then include what a launch waits for on the device, as a kernel trace shows it)")
    ap.add_argument("--sample", type=int, default=29, help="stage timers (HIP events) on every k-th picture (and every IDR): a sampled picture costs ~12 event records of ~5 us queue time each and runs its stages strictly in order (no deblocking beside the intra macroblocks of a P picture)")
    ap.add_argument("--streams-per-gpu", type=int, default=1, help="independent streams encoded concurrently on each GPU (one host thread each); the headline configuration is 1")
    ap.add_argument("--cavlc-threads", type=int, default=0, help="host threads coding one slice row-parallel (bit-identical output); 0 = the encoder's default (automatic, at most 8)")
    ap.add_argument("--rate-script", default="auto", help="bitrate setpoints written to the encoder while it runs: 'none', or a balancer of the reference "
                    "(adaptive|aimd|fixed: tests/golden/balancer_<name>.txt, generated from the reference's own balancer code); auto = adaptive for 1080p_ippp "
                    "(BASELINE.json configs[2]: 'balancer driving the bitrate property'), none elsewhere")
    ap.add_argument("--dct8x8", type=int, default=0, help="1: High profile, 8x8 transform for P macroblocks (x264enc dct8x8)")
    ap.add_argument("--i8x8", type=int, default=0, help="1 (with --dct8x8 1): Intra_8x8 macroblocks in I pictures (picture QP <= 37)")
    ap.add_argument("--aq", type=int, default=0, help="1: adaptive quantisation (aq-mode 1)")
    ap.add_argument("--single-stream", type=int, default=-1, help="1: every encoder runs its stages in order on ONE HIP stream (cfg.single_stream); -1: automatic (two or more streams per GPU: 2 streams 4 170 -> 4 840 frames/s in r04; ranks sharing a GPU: two ranks 5 771 -> 6 282)")
    ap.add_argument("--slices", type=int, default=-1, help="slices per P picture (cfg.slices: 0 automatic, 1 one slice); -1: the library's default (automatic: 4 at 1080p)")
    ap.add_argument("--slice-deblock", type=int, default=-1, help="1: the deblocking filter stops at slice boundaries (disable_deblocking_filter_idc 2), 0: runs across them; -1: the library's default (1)")
    args = ap.parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        # One process per stream, as the reference runs them (bindings/typescript/src/process.ts:129-170): start the ranks ourselves, each a fresh
        # child with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*, its device and its CPU set (ceracoder_amd/multistream.py).  Nothing has touched
        # the GPU in this process (no torch, no HIP so far), and the children are started, never exec'ed into.
        from ceracoder_amd import multistream
        code, out0, errs, plans = multistream.launch(args.gpus, [os.path.abspath(__file__)] + sys.argv[1:], timeout=3000)
        line = [l for l in (out0 or "").splitlines() if l.startswith("{")]
        if code or not line:
            for r, e in sorted(errs.items()):
                if e.strip():
                    sys.stderr.write("--- rank %d stderr (tail) ---\n%s\n" % (r, e))
            raise SystemExit(code or 1)
        print(line[-1], flush=True)
        return
    from ceracoder_amd import multistream as _ms
    _ms.self_plan()                   # ranks started by torch.distributed.run: same device / CPU plan as our own launcher's
    rank_cpus = _ms.apply_affinity()  # before the first GPU call: threads created from here on (HIP's, the entropy coders) inherit it
    if args.streams_per_gpu > 1:  # every encoder owns five HIP streams; the runtime's default of 4 hardware queues would serialise them
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

    import torch
    from ceracoder_amd.multistream import Ranks
    ranks = Ranks()  # control plane only (barrier + max of wall time) over gloo; no collective touches the data path
    rank, local_rank, world = ranks.rank, ranks.local_rank, ranks.world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the encoder has no CPU path")
    n_dev = torch.cuda.device_count()
    dev = int(os.environ.get("MI355_BENCH_DEVICE", local_rank)) % n_dev
    shared_gpu = args.shared_gpu or world > n_dev  # ranks beyond the box's GPUs share devices: no kernel may then wait on the device for another
    torch.cuda.set_device(dev)

    from ceracoder_amd import enc as E
    width, height, fps, gop, bps = WORKLOADS[args.workload]
    frames_np = make_source(width, height, args.unique)   # same clip on every rank would hide nothing: offset per rank
    if rank:
        frames_np = np.roll(frames_np, 7 * rank, axis=2)
    frames = torch.from_numpy(frames_np).to("cuda:%d" % dev)
    torch.cuda.synchronize()
    stride = width
    base = frames.data_ptr()
    fbytes = frames.stride(0)

    S = max(1, args.streams_per_gpu)
    # The reference's control loop (src/ceracoder.c:237-264) writes a new setpoint every balancer tick; the committed script holds
    # (time in ms, bit/s) pairs of one of its balancers for a scripted link, replayed here against stream time (picture index / fps).
    script_name = args.rate_script if args.rate_script != "auto" else ("adaptive" if args.workload == "1080p_ippp" else "none")
    script = []
    if script_name != "none" and args.fixed_qp < 0:
        here = os.path.dirname(os.path.abspath(__file__))
        for line in open(os.path.join(here, "tests", "golden", "balancer_%s.txt" % script_name)):
            t_ms, b = line.split()
            script.append((int(t_ms), int(b)))
    script_end = script[-1][0] if script else 0

    def setpoint(index):  # the script repeats when the run is longer than it
        t = (index * 1000 // fps) % script_end
        cur = script[0][1]
        for t_ms, b in script:
            if t_ms <= t:
                cur = b
        return cur

    if args.sample > 0:  # a short run (the driver's --steps 20) still gets at least four sampled pictures inside the timed region
        args.sample = max(1, min(args.sample, args.steps // 4 if args.steps >= 4 else 1))
    # the coding tools, the same in every encoder this script opens (None: the library's default)
    tools = dict(transform8x8=bool(args.dct8x8), i8x8=bool(args.i8x8), aq=bool(args.aq), slices=None if args.slices < 0 else args.slices,
                 slice_deblock=None if args.slice_deblock < 0 else bool(args.slice_deblock))
    def make_encoder():
        return E.Encoder(width, height, fps=fps, gop=gop, bitrate_bps=bps, device_id=dev, fixed_qp=args.fixed_qp,
                         pipeline_depth=args.depth, profile_events=args.sample, use_graphs=not args.no_graphs, deblock_mode=args.deblock_mode,
                         cavlc_threads=args.cavlc_threads, **tools, exclusive=(S == 1 and not shared_gpu), single_stream=(S >= 2 or shared_gpu) if args.single_stream < 0 else bool(args.single_stream), profile_overlap=not args.sample_in_order)

    encs = [make_encoder() for _ in range(S)]
    e = encs[0]

    def run_one(enc, n, first_index, shift):
        qps, nbytes, last_bps = [], 0, None
        for i in range(n):
            if script:
                b = setpoint(first_index + i)
                if b != last_bps:
                    enc.set_bitrate(b)
                    last_bps = b
            p = base + bounce(first_index + i + shift, args.unique) * fbytes
            enc.submit_device(p, stride, p + height * stride, stride, pts=first_index + i)
            if enc.pending > args.depth:
                sz, _, _, qp = enc.collect(copy=False)
                qps.append(qp)
                nbytes += sz
        while enc.pending:
            sz, _, _, qp = enc.collect(copy=False)
            qps.append(qp)
            nbytes += sz
        return qps, nbytes

    def run(n, first_index):
        if S == 1:
            return run_one(e, n, first_index, 0)
        import threading  # one host thread per stream; the C ABI releases the GIL for the duration of every call
        res = [None] * S
        th = [threading.Thread(target=lambda k=k: res.__setitem__(k, run_one(encs[k], n, first_index, 5 * k))) for k in range(S)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        return res[0][0], sum(r[1] for r in res)

    run(args.warmup, 0)
    for enc in encs:
        enc.reset_stats()
    dt, (qps, nbytes) = ranks.timed(lambda: run(args.steps, args.warmup), sync=torch.cuda.synchronize)
    st = e.stats()
    p_rows, i_rows = e.p_slice_rows, e.slice_rows  # rows per slice of P / I pictures (0: one slice)
    for enc in encs:  # (closed before the untimed extras open encoders of their own: several encoders in one process run their stages in order)
        enc.close()

    extra = {}
    if rank == 0:
        # Untimed extras.  (1) quality: a separate pass over two GOPs of the same clip with the same rate control (and the same
        # setpoint script), reconstruction fetched after every picture -- mean PSNR of Y, Cb, Cr over ALL its pictures.
        from ceracoder_amd import synth
        q_enc = E.Encoder(width, height, fps=fps, gop=gop, bitrate_bps=bps, device_id=dev, fixed_qp=args.fixed_qp, pipeline_depth=0, cavlc_threads=args.cavlc_threads, **tools)
        nq, ps, qbytes, last_b = (2 * gop if gop > 1 else 30), [], 0, None
        for i in range(nq):
            if script:
                b_ = setpoint(i)
                if b_ != last_b:
                    q_enc.set_bitrate(b_)
                    last_b = b_
            f = frames_np[bounce(i, args.unique)]
            qbytes += len(q_enc.encode(f[:height], f[height:], pts=i)[0])
            ry, ruv = q_enc.fetch(E.FETCH_RECON_Y)[:height, :width], q_enc.fetch(E.FETCH_RECON_UV)[:height // 2, :width]
            ps.append((synth.psnr(f[:height], ry), synth.psnr(f[height:, 0::2], ruv[:, 0::2]), synth.psnr(f[height:, 1::2], ruv[:, 1::2])))
        q_enc.close()
        pm = np.mean(np.array(ps), axis=0)
        extra["psnr_db"] = {"y": round(float(pm[0]), 2), "u": round(float(pm[1]), 2), "v": round(float(pm[2]), 2), "pictures": nq,
                            "bitrate_bps": round(qbytes * 8 * fps / nq), "min_y": round(float(np.min(np.array(ps)[:, 0])), 2),
                            "note": "mean over all pictures of an untimed pass (two GOPs, same clip, same rate control); encoder reconstruction vs source"}
        if not args.no_extras:
            # (2) per-picture latency of the synchronous path an element in a live graph uses (pipeline_depth 0):
            # host NV12 in -> H2D -> kernels -> D2H -> CAVLC -> access unit out, PCIe included.
            lat_enc = E.Encoder(width, height, fps=fps, gop=gop, bitrate_bps=bps, device_id=dev, fixed_qp=args.fixed_qp, pipeline_depth=0, cavlc_threads=args.cavlc_threads,
                                exclusive=not shared_gpu, **tools)
            def lat_run(n, gap_s):
                v = []
                for i in range(n):
                    f = frames_np[bounce(i, args.unique)]
                    if gap_s:
                        time.sleep(gap_s)
                    t0 = time.perf_counter()
                    lat_enc.encode(f[:height], f[height:], pts=i)
                    v.append((time.perf_counter() - t0) * 1e3)
                return np.sort(np.array(v[30:]))
            lat = lat_run(150, 0.0)         # back to back: a picture queues behind the deblocking of the one before (throughput-bound)
            live = lat_run(120, 1.0 / fps)  # paced like a live source at the workload's frame rate: the device is idle when a picture arrives
            lat_enc.close()
            extra["latency_ms"] = {"p50": round(float(live[len(live) // 2]), 3), "p95": round(float(live[int(len(live) * 0.95)]), 3),
                                   "p50_back_to_back": round(float(lat[len(lat) // 2]), 3), "p95_back_to_back": round(float(lat[int(len(lat) * 0.95)]), 3),
                                   "path": "pipeline_depth=0: host NV12 -> H2D -> GPU -> D2H -> host CAVLC -> AU (the element's handle_frame), pictures arriving at "
                                           "%d fps (p50/p95) or back to back; appsink->SRT segment not measurable here (no libsrt / mpegtsmux in the image)" % fps,
                                   "host_input_frames_per_s": round(1e3 / float(lat.mean()), 1)}

            # (3) host input with three pictures in flight (pipeline_depth 2), through mi355enc_submit: from ordinary (pageable) memory -- one
            # staging pass of the calling thread per picture, then an asynchronous transfer beside the kernels -- and from memory obtained with
            # mi355enc_host_alloc (what the element offers its upstream through the ALLOCATION query): transferred in place.  PCIe included.
            def host_run(src_frames, n):
                he = E.Encoder(width, height, fps=fps, gop=gop, bitrate_bps=bps, device_id=dev, fixed_qp=args.fixed_qp, pipeline_depth=2, cavlc_threads=args.cavlc_threads,
                               exclusive=not shared_gpu, **tools)
                def go(k, first):
                    for i in range(k):
                        f = src_frames[bounce(first + i, len(src_frames))]
                        he.submit(f[:height], f[height:], pts=first + i)
                        if he.pending > 2:
                            he.collect(copy=False)
                    while he.pending:
                        he.collect(copy=False)
                go(gop + 10, 0)  # warm-up: first GOP
                t0 = time.perf_counter()
                go(n, gop + 10)
                dt_h = time.perf_counter() - t0
                pinned = int(he.stats().pinned_inputs)
                he.close()
                return n / dt_h, pinned
            n_host = min(args.steps, 300) if args.steps >= 60 else 120
            r_page, _ = host_run(frames_np, n_host)
            extra["host_input_depth2_frames_per_s"] = round(r_page, 1)
            try:
                nu = min(args.unique, 16)
                pin = E.PinnedBuffer(nu * frames_np[0].nbytes)
                pframes = pin.array.reshape((nu,) + frames_np[0].shape)
                pframes[:] = frames_np[:nu]
                r_pin, n_pinned = host_run(pframes, n_host)
                extra["host_pinned_input_depth2_frames_per_s"] = round(r_pin, 1)
                extra["host_input_note"] = ("mi355enc_submit from host memory, pipeline_depth 2, PCIe in the loop, %d pictures after a warm-up GOP: pageable numpy planes "
                                            "(staged once by the calling thread) / planes in mi355enc_host_alloc memory (%d of them transferred in place)" % (n_host, n_pinned))
                del pframes
                pin.free()
            except E.EncoderError as ex:
                extra["host_pinned_input_depth2_frames_per_s"] = None
                extra["host_input_note"] = "pinned leg failed: %s" % ex
        # (4) an IDR picture's device time at this run's operating point, for the GOP-weighted rate of a timed region that holds no IDR picture
        # (the driver's --steps 20): fixed QP = the timed region's mean, stage timers on, a forced IDR picture every third picture.
        idr_probe = None
        if gop > 1:
            ie = E.Encoder(width, height, fps=fps, gop=10 ** 6, bitrate_bps=bps, device_id=dev, fixed_qp=int(round(float(np.mean(qps)))), pipeline_depth=0,
                           profile_events=1, cavlc_threads=args.cavlc_threads, **tools)
            for i in range(3):   # cold start: code objects, first launches
                f = frames_np[bounce(i, args.unique)]
                ie.encode(f[:height], f[height:], pts=i)
            ie.reset_stats()
            for i in range(3, 3 + 18):
                f = frames_np[bounce(i, args.unique)]
                ie.encode(f[:height], f[height:], pts=i, force_idr=(i % 3 == 0))
            ist = ie.stats()
            ie.close()
            if ist.n_intra and ist.n_deblock_idr:
                idr_probe = {"ms_intra": ist.ms_intra / ist.n_intra, "ms_deblock": ist.ms_deblock_idr / ist.n_deblock_idr, "samples": int(ist.n_intra),
                             "qp": int(round(float(np.mean(qps))))}
        extra["idr_probe"] = idr_probe

    if rank == 0 and world == 1 and not args.no_gst_latency:
        extra["latency_gst_ms"] = gst_latency(width, height, fps, gop, bps, dev)
        g = gst_throughput(width, height, fps, gop, bps, dev, args.depth, clip=frames_np)
        extra["gst_throughput"] = g
        extra["gst_frames_per_s"] = (g.get("element") or {}).get("frames_per_s")                     # videotestsrc in front: the source's own painting rate bounds it
        extra["gst_appsrc_frames_per_s"] = (g.get("element_appsrc_pinned") or {}).get("frames_per_s")  # pre-rendered pictures in pinned memory through appsrc
        extra["third_party"] = third_party_probe()

    if rank == 0:
        # HBM traffic of the kernels comes from separate rocprofv3 --pmc passes (cannot be collected from inside this
        # process); the committed summary of the latest pass for this workload is attached for cross-checking.
        prof = None
        try:
            cands = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("pmc_hbm_traffic_%s.json" % args.workload))
            if cands:
                prof = json.load(open(os.path.join(ROOT, "profiles", cands[-1])))["kernels"]
        except Exception:
            prof = None
        # ... and the kernel trace itself (profiles/*kernel_stats_<workload>.csv: one row per template instantiation).  A kernel's launch time is compared with the instantiation
        # that ran most often in the traced command -- for the fused P stage and the deblocking launch that is the free-running one, which waits on the device for the rows /
        # bands it follows and is what the stage timers of a free-running sampled picture bracket as well.
        trace_rows = {}
        try:
            import csv as _csv, re as _re
            cands = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("kernel_stats_%s.csv" % args.workload))
            if cands:
                for row in _csv.DictReader(open(os.path.join(ROOT, "profiles", cands[-1]))):
                    m = _re.match(r"^_Z(\d+)", row["Name"])
                    short = row["Name"][m.end():m.end() + int(m.group(1))] if m else row["Name"]
                    trace_rows.setdefault(short, []).append((int(row["Calls"]), float(row["TotalDurationNs"]) / int(row["Calls"]) / 1e3, row["Name"].replace(".kd", "")))
        except Exception:
            trace_rows = {}
        P = coded(width) * coded(height)
        # SURVEY.md 8(d) algorithmic bytes per launch (P = coded luma pixels).  Kernels the survey has no line for: sub-pel
        # refinement = cur luma P + reference window P + 16 B/MB record read and written = 2.125 P; one vector-selection
        # iteration = the macroblock's SAD surface read once (35 x 36 uint16) + the three neighbours' vectors + its own 8-byte
        # result = 2552 B/MB = 9.97 P (the stage this design adds to make the motion search's HBM output do the regularising).
        SEL = "vector selection: me_select_kernel (first pass, every surface) + 2 x me_select_sparse_kernel (later passes: changed macroblocks only)"
        ALG = {"me_kernel": 2.03125, SEL: 2552.0 / 256, "subpel_kernel": 2.125, "inter_kernel": 7.5625,
               "intra (analyse + x+y wavefront)": 6.0625, "deblock (prep + band kernel)": 3.0625}
        # what THIS design moves per launch where that differs from the survey's figure: me_kernel also writes the SAD surfaces (2520 B/MB) and the
        # source copy the next picture searches against -- traffic is to be compared with this one
        DESIGN = {"me_kernel": 3.0 + 2528.0 / 256}
        db_p = (st.ms_deblock - st.ms_deblock_idr, st.n_deblock - st.n_deblock_idr) if st.n_deblock > st.n_deblock_idr else (st.ms_deblock, st.n_deblock)
        FUSED = "pmb_kernel (skip probe + refinement + intra-or-inter + residual, fused)"
        fused = not args.dct8x8  # the 8x8-transform path keeps subpel_kernel + inter_kernel
        ALG[FUSED] = 7.5625      # the inter stage's bytes; probe and refinement re-read the same reference window
        pmb_ms = st.ms_inter - st.ms_analyse_p - st.ms_intra_p if fused else st.ms_inter
        # (the selection's three passes are bracketed by one event pair: one entry for the three launches of a picture)
        per = {"me_kernel": (st.ms_me, st.n_me), SEL: (st.ms_select, st.n_me), "subpel_kernel": (st.ms_subpel, 0 if fused else st.n_me),
               ("inter_kernel" if not fused else FUSED): (pmb_ms, st.n_inter),
               "intra (analyse + x+y wavefront)": (st.ms_intra, st.n_intra), "deblock (prep + band kernel)": db_p}
        bound = {"me_kernel": "VALU SAD issue rate (~142 T abs-diff/s chip-wide, tools/ubench_sad.hip): 1089*P abs-diffs -> >=17.6 us @1080p; besides the survey's "
                              "2.03 P it writes the 9.84 P of SAD surfaces (2520 B per macroblock) that the selection iterations and the fused stage read",
                 SEL: "first pass: HBM / Infinity Cache (streams every macroblock's SAD surface once); later passes: latency of the few changed macroblocks' surface reads (a wave checks eight macroblocks and walks the changed ones)",
                 "subpel_kernel": "LDS-staged 6-tap planes, latency/LDS", "inter_kernel": "launch + byte stores of interleaved chroma",
                 FUSED: "VALU issue: one wave per macroblock (skip probe; 6-tap planes, 8 SAD + 9 SATD candidates; transforms on all 64 lanes; decimation)",
                 "intra (analyse + x+y wavefront)": "dependency chain: Intra_4x4's left-neighbour dependency lets a macroblock start 4 block sub-steps (~0.8 us each) after the one before it, "
                                                    "so a row of mbw macroblocks is ~4*mbw sub-steps however many rows run side by side; + the lag between the rows of a slice (4 slices at 1080p)",
                 "deblock (prep + band kernel)": "dependency chain of the normative filter order inside one persistent launch (three waves per macroblock row: filter / mover / storer): per slice "
                                                 "%d + %d - 1 dependent steps of ~0.4-0.5 us + ~4 us per band hand-over, the slices side by side%s" % (
                                                     coded(width) // 16, p_rows if (p_rows and args.slice_deblock != 0) else coded(height) // 16,
                                                     " (slice-local deblocking: every slice is a wavefront of its own)" if (p_rows and args.slice_deblock != 0) else "")}
        pmc_name = {"me_kernel": "me_kernel", SEL: "me_select_kernel", FUSED: "pmb_kernel", "intra (analyse + x+y wavefront)": "intra_rows_kernel",
                    "deblock (prep + band kernel)": "deblock_rows3_kernel"}
        kernels = []
        n_idr, n_p = int(st.idr_frames), int(st.frames - st.idr_frames - st.skip_pictures)
        weight = {"me_kernel": n_p, SEL: n_p, "subpel_kernel": n_p, "inter_kernel": n_p, FUSED: n_p, "intra (analyse + x+y wavefront)": n_idr,
                  "deblock (prep + band kernel)": n_idr + n_p}  # launches in the timed region (timers are sampled)
        db_i_avg = st.ms_deblock_idr / st.n_deblock_idr if st.n_deblock_idr else 0.0
        if n_p and st.n_deblock_idr:  # deblocking of P pictures is the roofline entry; IDR pictures are added to the total separately
            weight["deblock (prep + band kernel)"] = n_p
        other_p = (st.ms_analyse_p + st.ms_intra_p) / max(1, st.n_inter)  # gated intra analysis + intra macroblocks of P pictures
        est_total = (sum(weight[k] * (ms / n) for k, (ms, n) in per.items() if n) + (n_idr * db_i_avg if n_p else 0.0) + n_p * other_p) or 1e-9
        period_us = dt / (S * args.steps) * 1e6
        for name, (ms, n) in per.items():
            if not n or ms <= 0:  # (a stage whose timer pair bracketed nothing on the sampled pictures: e.g. the stand-alone selection timer when adaptive quantisation reorders the front stream's records)
                continue
            us = ms / n * 1e3
            pk = (prof or {}).get(pmc_name.get(name, ""), {})
            alg_bytes, alg_note = ALG[name] * P, None
            if name == SEL and prof:  # the three launches of a picture together: the dense first pass + two sparse ones
                d_, s_ = prof.get("me_select_kernel", {}), prof.get("me_select_sparse_kernel", {})
                if d_.get("hbm_bytes_per_launch_corrected") and s_.get("hbm_bytes_per_launch_corrected"):
                    pk = {"hbm_bytes_per_launch_corrected": d_["hbm_bytes_per_launch_corrected"] + 2 * s_["hbm_bytes_per_launch_corrected"],
                          "kernel_trace_avg_us": round(d_.get("kernel_trace_avg_us", 0) + 2 * s_.get("kernel_trace_avg_us", 0), 3) if d_.get("kernel_trace_avg_us") and s_.get("kernel_trace_avg_us") else None}
            if name == SEL:
                # The selection copies a macroblock whose four predictor values did not change (24 bytes instead of its 2520-byte surface), so what a launch has to
                # read depends on the content: the upper bound (every surface, 9.97 P) is what iteration 1 reads; the mean over the three iterations is taken from
                # the committed counter pass of this workload (profiles/*pmc_hbm_traffic*), never from bytes that were not moved.
                if pk.get("hbm_bytes_per_launch_corrected"):
                    alg_bytes, alg_note = float(pk["hbm_bytes_per_launch_corrected"]), "data-dependent: bytes = memory-side traffic of a picture's three selection launches in the committed PMC pass (first pass + 2 x later pass: unchanged macroblocks are copied, their surfaces not read); the first pass alone reads every surface: %d B" % int(ALG[name] * P)
                else:
                    alg_note = "upper bound (every surface read); no PMC pass of this workload under profiles/ to take the mean from"
            ach = alg_bytes / (us * 1e-6) / 1e9
            tr_us, tr_inst = pk.get("kernel_trace_avg_us"), None
            rows_ = trace_rows.get(pmc_name.get(name, ""), [])
            if rows_ and name != SEL:  # the instantiation launched most often in the traced command
                calls_, tr_us, tr_inst = max(rows_)
                tr_us = round(tr_us, 3)
            k = {"kernel": name, "launches_timed": int(n), "avg_launch_us": round(us, 2), "algorithmic_bytes_per_launch": int(alg_bytes),
                 "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5),
                 "design_bytes_per_launch": int(DESIGN.get(name, ALG[name]) * P),
                 "traffic": pk.get("hbm_bytes_per_launch_corrected"), "kernel_trace_avg_us": tr_us, "kernel_trace_instantiation": tr_inst,
                 # the same fraction from the committed kernel trace's mean launch time (profiles/*kernel_stats*: what a reader can recompute without this run)
                 "frac_from_kernel_trace": round(alg_bytes / (tr_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 5) if tr_us else None,
                 "time_share": round(weight[name] * (ms / n) / est_total, 4), "bounded_by": bound[name]}
            if alg_note:
                k["algorithmic_bytes_note"] = alg_note
            if us > period_us and weight[name] and name != "intra (analyse + x+y wavefront)":
                k["avg_launch_note"] = ("longer than the picture period (%.1f us): the event pair brackets the launch on a SAMPLED picture (every %d-th), whose event records and device-side "
                                        "waits lengthen it; the kernel trace's mean is the unperturbed figure" % (period_us, args.sample))
            kernels.append(k)
        kernels.sort(key=lambda k: -k["time_share"])
        if not kernels:
            raise SystemExit("bench.py: no stage timers were sampled (--sample 0?): the roofline block needs them")
        dom = kernels[0]
        roof = {"bound": "hbm", "kernel": dom["kernel"], "achieved": dom["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": dom["frac"],
                "traffic": dom["traffic"], "avg_launch_us": dom["avg_launch_us"], "algorithmic_bytes_per_launch": dom["algorithmic_bytes_per_launch"],
                "launches": dom["launches_timed"], "kernel_trace_avg_us": dom["kernel_trace_avg_us"], "frac_from_kernel_trace": dom["frac_from_kernel_trace"],
                "avg_launch_note": dom.get("avg_launch_note"), "time_share": dom["time_share"],
                "note": "dominant kernel by GPU time; " + dom["bounded_by"] + ". Every kernel of the path is listed in roofline_kernels "
                        "(the motion search north_star names is 'me_kernel'; the HBM-shaped one is the vector selection)."}
        # What the device sustains over whole GOPs, whatever share of IDR pictures the timed region happened to hold: an IDR picture's stages run
        # strictly in order, so its time is its stage timers' sum; a P picture's stages overlap (front stream beside the previous picture's
        # deblocking, intra macroblocks beside the deblocker), so its effective period is what is left of the measured wall time.
        ip = extra.get("idr_probe")
        if st.n_intra:
            t_i, t_i_src = st.ms_intra / max(1, st.n_intra) + db_i_avg, "IDR pictures of the timed region"
        elif ip:
            t_i, t_i_src = ip["ms_intra"] + ip["ms_deblock"], "untimed IDR probe at the timed region's mean QP (%d), %d IDR pictures" % (ip["qp"], ip["samples"])
        else:
            t_i, t_i_src = 0.0, None
        n_p_all = int(st.frames - st.idr_frames - st.skip_pictures)
        t_p = (dt * 1e3 - n_idr * t_i) / max(1, n_p_all)  # (per encoder: with several streams per GPU each one coded its own `steps` pictures in dt)
        gop_fps = (1e3 * gop / ((gop - 1) * t_p + t_i)) if (gop > 1 and st.n_me and t_i > 0 and t_p > 0) else None
        out = {
            "metric": "1080p H.264 encoded frames/sec per GPU" if args.workload.startswith("1080p") else "H.264 encoded frames/sec per GPU",
            "value": round(world * S * (args.steps - int(st.skip_pictures)) / dt, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / (S * args.steps) * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic (S2: panning texture + 12 moving rectangles, seed 0x5EED), resident in HBM",
            "idr_in_timed_region": n_idr, "skip_pictures_in_timed_region": int(st.skip_pictures),
            "device_wait_recoveries": int(st.recoveries), "safe_level": int(st.safe_level),  # must be 0 / 0: a recovery means a bounded device-side wait ran out
            "value_kind": ("P pictures only: the timed region holds no IDR picture -- gop_weighted_frames_per_s is the rate over whole GOPs" if (n_idr == 0 and gop > 1) else
                           "whole GOPs" if gop > 1 and args.steps % gop == 0 else "all-intra" if gop == 1 else "%d IDR pictures among %d" % (n_idr, args.steps)),
            "value_note": ("over whole GOPs the device sustains gop_weighted_frames_per_s (below); " if (n_idr == 0 and gop > 1) else "") +
                          "value = pictures that went through the device per second: rate control's all-skip pictures (one P_Skip run written by the host, no kernel) are not counted",
            "frames_per_s_including_skip_pictures": round(world * S * args.steps / dt, 2),
            "gop_weighted_frames_per_s": round(gop_fps, 1) if gop_fps else None,
            "gop_weighted_note": "gop / ((gop-1) * t_P + t_IDR): t_IDR = an IDR picture's stage timers (its stages run in order), t_P = (wall time - IDR pictures * t_IDR) / "
                                 "P pictures of the timed region -- what the device sustains over whole GOPs, however many IDR pictures the timed region happened to contain",
            "gop_weighted_t_idr_ms": round(t_i, 4) if t_i else None, "gop_weighted_t_idr_source": t_i_src,
            "config": {"workload": args.workload, "width": width, "height": height, "fps_nominal": fps, "gop": gop, "h2d_in_timed_region": False,
                       "rate_control": ("fixed qp %d" % args.fixed_qp) if args.fixed_qp >= 0 else ("cbr %d bit/s" % bps) if not script else
                       "cbr, setpoint driven by the reference's '%s' balancer script (%d..%d kbit/s, tests/golden/balancer_%s.txt)" % (
                           script_name, min(b for _, b in script) // 1000, max(b for _, b in script) // 1000, script_name),
                       "me": "full search +-16 integer-pel SAD (surfaces kept) + 3 median-regularised selection iterations + half-sample SAD / quarter-sample SATD refinement", "streams_per_gpu": S, "hip_streams_per_encoder": 1 if ((S >= 2 or shared_gpu) if args.single_stream < 0 else bool(args.single_stream)) else 4, "parallelism": "%d independent streams" % (world * S),
                       "pipeline_depth": args.depth, "exclusive_device": bool(S == 1 and not shared_gpu), "devices_on_box": n_dev, "ranks_share_devices": bool(world > n_dev),
                       "rank0_cpus": rank_cpus if world > 1 else None, "rank0_numa_node": int(os.environ.get("MI355_BENCH_NUMA_NODE", "-1")), "dct8x8": bool(args.dct8x8), "i8x8": bool(args.i8x8), "aq_mode": int(args.aq), "cavlc_threads": int(st.cavlc_threads),
                       "p_slice_rows": int(p_rows), "i_slice_rows": int(i_rows), "slices_per_p_picture": (coded(height) // 16 + p_rows - 1) // p_rows if p_rows else 1,
                       "slice_local_deblocking": (args.slice_deblock != 0)},
            "roofline": roof,
            "roofline_kernels": kernels,
            "stage_timers": ("HIP events on every %d-th P picture, on the streams the kernels are launched on" % args.sample) + (
                "; the sampled picture runs its stages strictly in order (every timer is one kernel alone)" if args.sample_in_order else
                "; the sampled picture keeps the free-running schedule: a launch that waits on the device for another kernel's rows (the gated P stage, the deblocking launch) "
                "is timed with that wait, as the rocprofv3 kernel trace shows it") + "; IDR pictures: every other one, in order",
            "stage_ms_per_picture": {"me": round(st.ms_me / max(1, st.n_me), 4), "me_select_x3": round(st.ms_select / max(1, st.n_me), 4),
                                     ("p_stage_total" if fused else "inter"): round(st.ms_inter / max(1, st.n_inter), 4),
                                     "p_intra_analyse_gated": round(st.ms_analyse_p / max(1, st.n_inter), 4), "p_intra_macroblocks": round(st.ms_intra_p / max(1, st.n_inter), 4),
                                     "subpel": round(st.ms_subpel / max(1, st.n_me), 4),
                                     "intra_wavefront": round(st.ms_intra / max(1, st.n_intra), 4),
                                     "deblock_wavefront": round(db_p[0] / max(1, db_p[1]), 4), "deblock_wavefront_idr": round(db_i_avg, 4),
                                     "gpu_total": round(est_total / max(1, st.frames), 4),
                                     "host_cavlc": round(st.ms_entropy / max(1, st.frames), 4),
                                     "host_wait": round(st.ms_wait / max(1, st.frames), 4)},
            "bitrate_out_bps": round(nbytes * 8 * fps / (S * args.steps)),
            "bitrate_setpoint_mean_bps": round(float(np.mean([setpoint(args.warmup + i) for i in range(args.steps)]))) if script else bps, "mean_qp": round(float(np.mean(qps)), 2),
        }
        out.update(extra)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(frames_np, width, height, fps, gop, qps)
        print(json.dumps(out), flush=True)
    ranks.close()


if __name__ == "__main__":
    main()
