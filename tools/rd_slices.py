#!/usr/bin/env python3
"""dev tool: what slices cost in bits (CPU oracle = the device bit for bit): fixed-QP rate-distortion points of a clip coded with one slice per picture, with P
slices and the deblocking filter across the seams (idc 0), and with slice-local deblocking (idc 2); Bjontegaard-style rate difference against the one-slice
stream, and the PSNR of the seam rows (the 16 luma lines either side of every slice boundary) beside the whole picture's.
    python tools/rd_slices.py [--size 1920x1080] [--frames 24] [--clip s2|s4pan] [--slices 4,5,8] [--qps 24,30,36,42]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ceracoder_amd import synth
from oracle import oracle as O

ap = argparse.ArgumentParser()
ap.add_argument("--size", default="1920x1080")
ap.add_argument("--frames", type=int, default=24)
ap.add_argument("--clip", default="s2")
ap.add_argument("--qps", default="24,30,36,42")
ap.add_argument("--slices", default="5")
ap.add_argument("--threads", type=int, default=8)
args = ap.parse_args()
w, h = (int(v) for v in args.size.split("x"))
gen = {"s2": lambda: synth.s2_frames(w, h, args.frames), "s4": lambda: synth.s4_frames(w, h, args.frames),
       "s4pan": lambda: synth.s4_frames(w, h, args.frames, pan_after=args.frames // 3)}[args.clip]
clip = list(gen())
qps = [int(q) for q in args.qps.split(",")]
mbh = (h + 15) // 16


def seam_mask(rows):
    m = np.zeros(h, bool)
    if rows:
        for s in range(rows * 16, h, rows * 16):
            m[max(0, s - 16):min(h, s + 16)] = True
    return m


def run(ns, local, mask):
    pts = []
    for qp in qps:
        e = O.Encoder(w, h, fps=60, gop=args.frames, threads=args.threads, intra_slices=ns if ns > 1 else 1, p_slices=ns, slice_deblock_local=local)
        bits, ps, seam = 0, [], []
        for y, uv in clip:
            au, _ = e.encode(y, uv, qp)
            bits += 8 * len(au)
            ry = e.recon_y[:h, :w]
            ps.append(synth.psnr(y, ry))
            if mask.any():
                seam.append(synth.psnr(y[mask], ry[mask]))
        e.close()
        pts.append((bits * 60.0 / args.frames / 1e3, float(np.mean(ps)), float(np.mean(seam)) if seam else float("nan")))
    return pts


def bd_rate(a, b):
    la, pa = np.log([p[0] for p in a])[::-1], np.array([p[1] for p in a])[::-1]
    lb, pb = np.log([p[0] for p in b])[::-1], np.array([p[1] for p in b])[::-1]
    lo, hi = max(pa.min(), pb.min()), min(pa.max(), pb.max())
    xs = np.linspace(lo, hi, 50)
    return (np.exp(np.mean(np.interp(xs, pb, lb) - np.interp(xs, pa, la))) - 1) * 100


print("%dx%d %s, %d pictures (1 I + P), QP %s" % (w, h, args.clip, args.frames, args.qps))
for ns in [int(v) for v in args.slices.split(",")]:
    rows = O.slice_rows_for(mbh, ns, True)
    mask = seam_mask(rows)
    t0 = time.time()
    base = run(1, False, mask)  # one slice; the same rows measured
    for name, n_, loc in (("%d slices (rows %d), idc 0" % (ns, O.slice_rows_for(mbh, ns, False)), ns, False), ("%d slices (rows %d), idc 2" % (ns, rows), ns, True)):
        pts = run(n_, loc, seam_mask(O.slice_rows_for(mbh, n_, loc)) if not loc else mask)
        print("%s: BD-rate vs one slice %+.2f %%   [%.0f s]" % (name, bd_rate(base, pts), time.time() - t0))
        for qp, b, p in zip(qps, base, pts):
            print("   qp %2d  one slice %9.1f kbit/s PSNR-Y %.2f (seam rows %.2f) | sliced %9.1f kbit/s (%+.2f %%) PSNR-Y %.2f (%+.3f) seam rows %.2f (%+.3f)" %
                  (qp, b[0], b[1], b[2], p[0], 100 * (p[0] / b[0] - 1), p[1], p[1] - b[1], p[2], p[2] - b[2]))
