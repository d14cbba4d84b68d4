#!/usr/bin/env python3
"""dev tool (CPU, oracle): how often could a deblocking band be cut in two at a macroblock column where the filter does nothing?  For the P pictures of the S2 clip under a
fixed quantiser: per band of MI355_BAND_ROWS macroblock rows (inside a slice), is there a column x within +-W of the middle where the vertical macroblock edge x-1 | x has
bS = 0 in every row of the band (both macroblocks inter, no luma / chroma coefficients in the blocks that touch the edge, vectors less than a sample apart) -- there the
left and the right part of the band share no filtered sample, so two workgroups could walk them side by side.
    python tools/split_probe.py [W H [qp [pictures]]]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ceracoder_amd import synth
from oracle import oracle as O
w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
qp = int(sys.argv[3]) if len(sys.argv) > 3 else 31
n = int(sys.argv[4]) if len(sys.argv) > 4 else 8
BAND = 4
mbw, mbh = (w + 15) // 16, (h + 15) // 16
ns = O.auto_slices(mbh)
rows = O.slice_rows_for(mbh, ns, True)
oe = O.Encoder(w, h, gop=600, threads=8, intra_slices=0, p_slices=ns, slice_deblock_local=True, scenecut=False)
clip = list(synth.s2_frames(w, h, n + 1))
LEFT = sum(1 << b for b in (0, 2, 8, 10))    # luma blocks in the macroblock's first column (blkIdx)
RIGHT = sum(1 << b for b in (5, 7, 13, 15))  # ... and last column
CH = (0xFF << 16) | (3 << 24)               # any chroma AC / DC (coarse: a chroma coefficient anywhere keeps the edge)
tot = {8: [0, 0], 16: [0, 0], 30: [0, 0]}
for i, (y, uv) in enumerate(clip):
    oe.encode(y, uv, qp)
    if i == 0:
        continue
    mb = oe.mbinfo.reshape(mbh, mbw)
    inter = mb["mb_type"] == 1
    quiet_l = inter & ((mb["nzmask"] & (LEFT | CH)) == 0)   # macroblock x: nothing on its left edge
    quiet_r = inter & ((mb["nzmask"] & (RIGHT | CH)) == 0)  # macroblock x-1: nothing on its right edge
    dmx = np.abs(mb["mvx"][:, 1:].astype(int) - mb["mvx"][:, :-1]) < 4
    dmy = np.abs(mb["mvy"][:, 1:].astype(int) - mb["mvy"][:, :-1]) < 4
    free = np.zeros((mbh, mbw), bool)
    free[:, 1:] = quiet_l[:, 1:] & quiet_r[:, :-1] & dmx & dmy & (mb["i16_mode"][:, 1:] == 0) & (mb["i16_mode"][:, :-1] == 0)  # (partitioned macroblocks: left alone)
    for r0 in range(0, mbh, rows):
        for b0 in range(r0, min(r0 + rows, mbh), BAND):
            band = free[b0:min(b0 + BAND, r0 + rows, mbh)].all(axis=0)
            for W in tot:
                lo, hi = max(1, mbw // 2 - W), min(mbw - 1, mbw // 2 + W)
                tot[W][0] += int(band[lo:hi].any()); tot[W][1] += 1
print("%dx%d, QP %d, %d P pictures, %d-row slices, bands of %d rows:" % (w, h, qp, n, rows, BAND))
for W in sorted(tot):
    print("  a column with bS = 0 in every row of the band within +-%d macroblocks of the middle: %d of %d bands (%.0f %%)" % (W, tot[W][0], tot[W][1], 100.0 * tot[W][0] / tot[W][1]))
