#!/usr/bin/env python3
"""dev tool: rate-distortion points of the CPU oracle at fixed QPs (what the device computes bit for bit), per feature set of
the P-macroblock stage.  python tools/rd_oracle.py [--size 640x368] [--frames 30] [--clip s2|s4|s4pan] [--feat all|LIST]
Prints kbit/s at 60 fps and mean PSNR-Y/U/V over all pictures, and a Bjontegaard-style average rate difference against the
first feature set."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ceracoder_amd import synth
from oracle import oracle as O

ap = argparse.ArgumentParser()
ap.add_argument("--size", default="640x368")
ap.add_argument("--frames", type=int, default=30)
ap.add_argument("--clip", default="s2")
ap.add_argument("--qps", default="24,30,36,42,48")
ap.add_argument("--feat", default="0,1,3,7,15,31")
ap.add_argument("--iters", default="2", help="orc_me_select iterations (comma list: one table per value)")
ap.add_argument("--tune", default="", help="dev: which=value,... for orc_set_tuning")
ap.add_argument("--lib", default=None, help="another build of liboracle_h264.so (e.g. of an older commit) for the first row")
args = ap.parse_args()
w, h = (int(v) for v in args.size.split("x"))
gen = {"s2": lambda: synth.s2_frames(w, h, args.frames), "s4": lambda: synth.s4_frames(w, h, args.frames),
       "s4pan": lambda: synth.s4_frames(w, h, args.frames, pan_after=args.frames // 3)}[args.clip]
clip = list(gen())
import ctypes
O.lib().orc_set_tuning.argtypes = [ctypes.c_int, ctypes.c_int]
for kv in filter(None, args.tune.split(",")):
    k, v = kv.split("=")
    O.lib().orc_set_tuning(int(k), int(v))
qps = [int(q) for q in args.qps.split(",")]


def run(feat, iters):
    O.set_features(feat)
    pts = []
    for qp in qps:
        e = O.Encoder(w, h, fps=60, gop=args.frames, threads=8, me_iters=iters)
        bits, ps = 0, []
        for y, uv in clip:
            au, _ = e.encode(y, uv, qp)
            bits += 8 * len(au)
            ry, ruv = e.recon_y[:h, :w], e.recon_uv[:h // 2, :w]
            ps.append((synth.psnr(y, ry), synth.psnr(uv[:, 0::2], ruv[:, 0::2]), synth.psnr(uv[:, 1::2], ruv[:, 1::2])))
        e.close()
        pts.append((bits * 60.0 / args.frames / 1e3, *np.mean(ps, axis=0)))
    return pts


def bd_rate(a, b):
    """average log-rate difference of curve b against a over the common PSNR-Y range (piecewise-linear interpolation), in %"""
    la, pa = np.log([p[0] for p in a])[::-1], np.array([p[1] for p in a])[::-1]
    lb, pb = np.log([p[0] for p in b])[::-1], np.array([p[1] for p in b])[::-1]
    lo, hi = max(pa.min(), pb.min()), min(pa.max(), pb.max())
    if hi <= lo:
        return float("nan")
    xs = np.linspace(lo, hi, 50)
    return (np.exp(np.mean(np.interp(xs, pb, lb) - np.interp(xs, pa, la))) - 1) * 100


names = {0: "none", 1: "mvdcost", 2: "skip probe", 4: "decimate", 8: "satd", 16: "intra in P", 32: "Intra_4x4 in P"}
base = None
for f, it in [(int(v), int(i)) for i in args.iters.split(",") for v in args.feat.split(",")]:
    t0 = time.time()
    pts = run(f, it)
    label = ("+".join(n for b, n in names.items() if b and f & b) or "none") + ", %d iterations" % it
    if base is None:
        base = pts
    print("feat %2d (%s)  BD-rate vs first: %+.1f %%   [%.0f s]" % (f, label, bd_rate(base, pts), time.time() - t0))
    for qp, p in zip(qps, pts):
        print("   qp %2d  %9.1f kbit/s  PSNR Y %.2f U %.2f V %.2f" % (qp, *p))
