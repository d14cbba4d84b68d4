import sys; sys.path.insert(0, '.')
import numpy as np
from ceracoder_amd import enc as E, synth
w, h, fps, gop = 3840, 2160, 60, 60
clip = list(synth.s2_frames(w, h, 8))
depth = int(sys.argv[1])
e = E.Encoder(w, h, fps=fps, gop=gop, bitrate_bps=20_000_000, pipeline_depth=depth, exclusive=depth == 2)
out = []
for i in range(4 * gop):
    k = i % (2 * len(clip) - 2)
    y, uv = clip[k if k < len(clip) else 2 * len(clip) - 2 - k]
    e.submit(y, uv, pts=i)
    if e.pending > depth:
        au, key, pts, qp = e.collect(copy=False); out.append((au, qp, e.last_drop, key))
while e.pending:
    au, key, pts, qp = e.collect(copy=False); out.append((au, qp, e.last_drop, key))
for g in (2, 3):
    seg = out[g * gop:(g + 1) * gop]
    print("depth %d GOP %d: %.3f of the setpoint" % (depth, g, sum(s[0] for s in seg) * 8 * fps / gop / 20e6))
    print(" ".join("%d:%d/%d" % (s[1], s[2], s[0] // 1000) for s in seg))
