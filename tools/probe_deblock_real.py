#!/usr/bin/env python3
"""Deblocking time on real pictures of the S2 clip (one I, then P pictures) at several QPs, for A/B runs of kernel variants
(MI355ENC_LIB=... selects the library)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ceracoder_amd import enc as E, synth
E.LIB_PATH = os.environ.get("MI355ENC_LIB", E.LIB_PATH)
w, h = 1920, 1080
fr = list(synth.s2_frames(w, h, 4))
out = []
for qp in (24, 32, 41, 48):
    e = E.Encoder(w, h, gop=60, fixed_qp=qp)
    e.encode(*fr[0])
    t_i = e.time_stage(E.STAGE_DEBLOCK, 20)
    for f in fr[1:]:
        e.encode(*f)
    t_p = e.time_stage(E.STAGE_DEBLOCK, 20)
    out.append("qp %d: I %.3f ms  P %.3f ms" % (qp, t_i, t_p))
    e.close()
print(" | ".join(out))
