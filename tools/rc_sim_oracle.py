#!/usr/bin/env python3
"""dev tool (CPU): the product's rate control (ratecontrol.c through the C ABI's host entry points) in closed loop with the CPU oracle encoder -- the device's
bytes, bit for bit -- at a small picture size with the setpoints scaled by the picture area: tests/test_ratecontrol_gpu.py's steps without a GPU, a few seconds per
run.  Prints, per step, the rate of the GOP that starts with the step and of the one after it as fractions of the setpoint.
    python tools/rc_sim_oracle.py [--size 640x368] [--clip s2|s4] [--delay 0|1|2] [--trace STEP]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ceracoder_amd import enc as E, synth
from oracle import oracle as O

ap = argparse.ArgumentParser()
ap.add_argument("--size", default="640x368")
ap.add_argument("--clip", default="s2")
ap.add_argument("--delay", default="1", help="comma list: picture sizes still unknown at every pick (pipeline_depth)")
ap.add_argument("--steps", default="6000,300,1000,1500,20000,30000,6000", help="kbit/s at 1920x1080; scaled by the picture area")
ap.add_argument("--trace", type=int, default=-1)
args = ap.parse_args()
w, h = (int(v) for v in args.size.split("x"))
scale = (w * h) / (1920.0 * 1080.0)
steps = [max(1000, int(float(s) * 1000 * scale)) for s in args.steps.split(",")]
fps, gop = 60, 60
clip = list(synth.s2_frames(w, h, 16) if args.clip == "s2" else synth.s4_frames(w, h, 16))
for delay in [int(d) for d in args.delay.split(",")]:
    rc = E.RateControl(fps, gop, steps[0])
    oe = O.Encoder(w, h, fps=fps, gop=gop, threads=8, scenecut=False)
    sizes, pend, rows = [], [], []
    n = len(steps) * 2 * gop
    for i in range(n):
        if i % (2 * gop) == 0:
            rc.set_bitrate(steps[i // (2 * gop)])
        k = i % (2 * len(clip) - 2)
        y, uv = clip[k if k < len(clip) else 2 * len(clip) - 2 - k]
        idr = i % gop == 0
        qp, drop = rc.pick(idr)
        au, key = oe.encode(y, uv, qp, drop=drop, force_idr=idr)
        assert key == idr
        pend.append((idr, qp, drop, len(au)))
        sizes.append(len(au)); rows.append((qp, drop, len(au)))
        if len(pend) > delay:
            rc.update(*pend.pop(0))
    sizes = np.array(sizes, float)
    out = []
    for k, bps in enumerate(steps):
        r = [sizes[(2 * k + g) * gop:(2 * k + g + 1) * gop].sum() * 8 * fps / gop / bps for g in range(2)]
        out.append("%d: %.3f %.3f" % (bps // 1000, r[0], r[1]))
    print("%s %s delay %d | " % (args.size, args.clip, delay) + " | ".join(out), flush=True)
    if args.trace >= 0:
        for g in range(2):
            seg = rows[(2 * args.trace + g) * gop:(2 * args.trace + g + 1) * gop]
            print(" ".join("%d:%d/%d" % r for r in seg))
