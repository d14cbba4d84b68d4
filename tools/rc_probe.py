#!/usr/bin/env python3
"""dev tool (GPU): per-GOP rates, QPs and drop levels of a run that steps the setpoint.  python tools/rc_probe.py 1920x1080 s2 [gop]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ceracoder_amd import enc as E, synth
w, h = (int(v) for v in sys.argv[1].split("x"))
kind = sys.argv[2] if len(sys.argv) > 2 else "s2"
gop = int(sys.argv[3]) if len(sys.argv) > 3 else 60
fps = 60
steps = [6_000_000, 300_000, 1_000_000, 1_500_000, 20_000_000, 30_000_000, 6_000_000] if gop > 1 else [6_000_000]
clip = list(synth.s2_frames(w, h, 16) if kind == "s2" else synth.s4_frames(w, h, 16))
e = E.Encoder(w, h, fps=fps, gop=gop, bitrate_bps=steps[0], pipeline_depth=1)
G = 60
rows = []
n = len(steps) * 2 * G
for i in range(n):
    if i % (2 * G) == 0:
        e.set_bitrate(steps[i // (2 * G)])
    k = i % (2 * len(clip) - 2)
    y, uv = clip[k if k < len(clip) else 2 * len(clip) - 2 - k]
    e.submit(y, uv, pts=i)
    if e.pending > 1:
        au, key, pts, qp = e.collect(copy=False); rows.append((au, key, qp, e.last_drop))
while e.pending:
    au, key, pts, qp = e.collect(copy=False); rows.append((au, key, qp, e.last_drop))
a = np.array([(r[0], r[1], r[2], r[3]) for r in rows], float)
if len(sys.argv) > 4:
    g = int(sys.argv[4])
    for i, r in enumerate(rows[g * G:(g + 1) * G]):
        print(i, "IDR" if r[1] else "P", "qp", r[2], "drop", r[3], "bytes", r[0])
for g in range(len(rows) // G):
    s = a[g * G:(g + 1) * G]
    idr = s[s[:, 1] == 1]
    print("gop %2d  target %9d  rate %9.0f  (%+5.1f %%)  idr bytes %s  qp mean %.1f max %d  drop mean %.1f max %d  skip pictures %d  P bytes median %d" % (
        g, steps[g // 2], s[:, 0].sum() * 8 * fps / G, (s[:, 0].sum() * 8 * fps / G / steps[g // 2] - 1) * 100, [int(v) for v in idr[:, 0]][:3],
        s[:, 2].mean(), s[:, 2].max(), s[s[:, 3] < 255][:, 3].mean(), s[s[:, 3] < 255][:, 3].max(), (s[:, 3] == 255).sum(), np.median(s[s[:, 1] == 0][:, 0]) if (s[:, 1] == 0).any() else 0))
