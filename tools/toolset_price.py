#!/usr/bin/env python3
"""GPU: what the coding tools cost and buy.
 (a) toolsets: the element's speed-preset mapping -- none / ultrafast (Constrained Baseline, one QP per picture) against superfast and slower (dct8x8 + i8x8 + aq-mode 1)
     -- at 1080p and 2160p: pictures/s (three in flight, exclusive device, sources resident in HBM), and at EQUAL fixed QP the bitrate and PSNR of the same clip;
 (b) slices: one slice per picture against the default (P pictures cut like I pictures, slice-local deblocking) under CBR at 2 / 6 / 12 Mbit/s, 1080p S2: produced
     bitrate, PSNR-Y of the whole picture and of the seam rows (the 16 luma lines either side of every slice boundary), pictures/s.
    python tools/toolset_price.py [out.md]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ceracoder_amd import enc as E, synth

lines = []
def say(s=""):
    print(s, flush=True); lines.append(s)

def speed(w, h, n, **kw):
    clip = list(synth.s2_frames(w, h, 16))
    bufs = [torch.from_numpy(np.concatenate([y.reshape(-1), uv.reshape(-1)])).cuda() for y, uv in clip]
    torch.cuda.synchronize()
    e = E.Encoder(w, h, fps=60, gop=60, pipeline_depth=2, exclusive=True, **kw)
    def run(cnt, base):
        nb = 0
        for i in range(cnt):
            k = (base + i) % 30
            p = bufs[k if k < 16 else 30 - k].data_ptr()
            e.submit_device(p, w, p + w * h, w, pts=base + i)
            if e.pending > 2: nb += e.collect(copy=False)[0]
        while e.pending: nb += e.collect(copy=False)[0]
        return nb
    run(60, 0)
    t0 = time.perf_counter(); nb = run(n, 60); t = time.perf_counter() - t0
    rec = e.stats().recoveries
    e.close()
    assert rec == 0
    return n / t, nb * 8 * 60 / n

def quality(w, h, n, seam_rows=0, **kw):
    clip = list(synth.s2_frames(w, h, 16))
    e = E.Encoder(w, h, fps=60, gop=60, pipeline_depth=0, **kw)
    mask = np.zeros(h, bool)
    if seam_rows:
        for s in range(seam_rows * 16, h, seam_rows * 16): mask[max(0, s - 16):min(h, s + 16)] = True
    ps, seam, nb, qps = [], [], 0, []
    for i in range(n):
        k = i % 30
        y, uv = clip[k if k < 16 else 30 - k]
        e.submit(y, uv, pts=i)
        au, key, pts, qp = e.collect(copy=False)
        if i >= 60 or n <= 60:
            ry = e.fetch(E.FETCH_RECON_Y)[:h, :w]
            ps.append(synth.psnr(y, ry)); nb += au; qps.append(qp)
            if mask.any(): seam.append(synth.psnr(y[mask], ry[mask]))
    e.close()
    cnt = len(ps)
    return nb * 8 * 60 / cnt, float(np.mean(ps)), (float(np.mean(seam)) if seam else float("nan")), float(np.mean(qps))

BASE = dict(slices=None, slice_deblock=None)
say("## (a) toolsets (speed-preset): none / ultrafast against superfast and slower (dct8x8 + i8x8 + aq-mode 1); library-default slices in both")
say("| geometry | toolset | pictures/s (CBR, 3 in flight) | fixed QP | kbit/s | PSNR-Y |")
say("|---|---|---|---|---|---|")
for (w, h, bps, n) in ((1920, 1080, 6_000_000, 600), (3840, 2160, 20_000_000, 300)):
    for name, tools in (("baseline (speed-preset 0/1)", {}), ("dct8x8 + i8x8 + aq-mode 1 (speed-preset >= 2)", dict(transform8x8=True, i8x8=True, aq=True))):
        fps, _ = speed(w, h, n, bitrate_bps=bps, **BASE, **tools)
        for qp in ((26, 32, 38) if w == 1920 else (32,)):
            rate, p, _, _ = quality(w, h, 60 if w == 1920 else 30, fixed_qp=qp, **BASE, **tools)
            say("| %dx%d | %s | %.0f | %d | %.0f | %.2f |" % (w, h, name, fps, qp, rate / 1e3, p))
say()
say("## (b) slices under CBR, 1080p60 S2 (two GOPs measured after one of settling)")
say("| setpoint Mbit/s | slices | produced Mbit/s | mean QP | PSNR-Y | PSNR-Y of the seam rows | pictures/s |")
say("|---|---|---|---|---|---|---|")
w, h = 1920, 1080
probe = E.Encoder(w, h, slices=None, slice_deblock=None)
rows = probe.p_slice_rows
probe.close()
for bps in (2_000_000, 6_000_000, 12_000_000):
    for name, kw in (("one slice, filter across (r03)", dict(slices=1, slice_deblock=False, intra_slices=1)), ("default: %d-row slices, slice-local deblocking" % rows, BASE)):
        rate, p, ps, q = quality(w, h, 180, seam_rows=rows, bitrate_bps=bps, **kw)
        fps, _ = speed(w, h, 600, bitrate_bps=bps, **kw)
        say("| %.0f | %s | %.2f | %.1f | %.2f | %.2f | %.0f |" % (bps / 1e6, name, rate / 1e6, q, p, ps, fps))
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write("\n".join(lines) + "\n")
