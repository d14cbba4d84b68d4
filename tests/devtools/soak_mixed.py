#!/usr/bin/env python3
"""dev tool: a 1080p stream that walks through moving, still, half-still and noise scenes (hard cuts between them) under CBR with
pipeline_depth 1 (or argv[1]); every access unit goes through the independent decoder and must equal the encoder's reconstruction at
every scene end.  Exercises scene-cut recovery, idle deblocking bands and the rate control together."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from ceracoder_amd import enc as E, synth
from oracle import oracle as O
w, h = 1920, 1080
s2 = list(synth.s2_frames(w, h, 24))
s3 = list(synth.s3_frames(w, h, 6))
half = [(np.concatenate([s2[0][0][:544], f[0][544:]]), np.concatenate([s2[0][1][:272], f[1][272:]])) for f in s2]
scenes = [("moving", [s2[i % 24] for i in range(70)]), ("still", [s2[5]] * 45), ("half still", [half[i % 24] for i in range(58)]),
          ("noise", [s3[i % 6] for i in range(22)]), ("moving again", [s2[(3 * i) % 24] for i in range(70)])]
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 1
e = E.Encoder(w, h, fps=60, gop=60, bitrate_bps=8_000_000, pipeline_depth=depth, exclusive=depth == 2)
dec = O.Decoder()
t0, n, keys, pend = time.time(), 0, [], []
def take():
    au, key, pts, qp = e.collect()
    if key:
        keys.append(pts)
    return dec.decode(au)
for name, clip in scenes:
    for y, uv in clip:
        e.submit(y, uv, pts=n)
        n += 1
        if e.pending > depth:
            y_d, uv_d = take()
    while e.pending:
        y_d, uv_d = take()
    assert np.array_equal(y_d, e.fetch(E.FETCH_RECON_Y)) and np.array_equal(uv_d, e.fetch(E.FETCH_RECON_UV)), "drift at the end of scene '%s'" % name
    print("scene '%s' ok (%d pictures so far)" % (name, n), flush=True)
print("ok: %d pictures in %.1f s, IDR pictures at %s" % (n, time.time() - t0, keys))
