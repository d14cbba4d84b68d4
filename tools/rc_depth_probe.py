#!/usr/bin/env python3
"""dev tool: rate-control accuracy after setpoint steps at pipeline_depth 1 and 2 (tests/test_ratecontrol_gpu.py's clip and steps);
with arguments `depth step_index`, the per-picture trace (QP, drop level, bytes) of the two GOPs after that step."""
import sys; sys.path.insert(0, '.')
import numpy as np
from ceracoder_amd import enc as E, synth
from tests.test_ratecontrol_gpu import STEPS
w, h, fps, gop = 1920, 1080, 60, 60
clip = list(synth.s2_frames(w, h, 16))
def run(depth, steps, gops_per_step=2):
    e = E.Encoder(w, h, fps=fps, gop=gop, bitrate_bps=steps[0], pipeline_depth=depth, exclusive=depth == 2)
    out = []
    n = len(steps) * gops_per_step * gop
    for i in range(n):
        if i % (gops_per_step * gop) == 0:
            e.set_bitrate(steps[i // (gops_per_step * gop)])
        k = i % (2 * len(clip) - 2)
        y, uv = clip[k if k < len(clip) else 2 * len(clip) - 2 - k]
        e.submit(y, uv, pts=i)
        if e.pending > depth:
            au, key, pts, qp = e.collect(copy=False); out.append((au, qp, e.last_drop, key))
    while e.pending:
        au, key, pts, qp = e.collect(copy=False); out.append((au, qp, e.last_drop, key))
    e.close()
    return out
if len(sys.argv) > 2:
    depth, k = int(sys.argv[1]), int(sys.argv[2])
    out = run(depth, STEPS)
    for g in range(2):
        seg = out[(2 * k + g) * gop:(2 * k + g + 1) * gop]
        print("GOP %d: %.3f of the setpoint" % (g, sum(s[0] for s in seg) * 8 * fps / gop / STEPS[k]))
        print(" ".join("%d:%d/%d" % (s[1], s[2], s[0] // 1000) for s in seg))
else:
    for depth in (1, 2):
        out = run(depth, STEPS)
        sizes = np.array([o[0] for o in out], float)
        for k, bps in enumerate(STEPS):
            r = [sizes[(2 * k + g) * gop:(2 * k + g + 1) * gop].sum() * 8 * fps / gop / bps for g in range(2)]
            print(depth, bps, "%.3f %.3f" % tuple(r))
