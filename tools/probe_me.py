#!/usr/bin/env python3
"""dev tool: time of the three P-picture kernels (MI355ENC_LIB selects an alternative build).  python tools/probe_me.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ceracoder_amd import enc as E, synth
if os.environ.get("MI355ENC_LIB"):
    E.LIB_PATH = os.environ["MI355ENC_LIB"]
for (w, h) in ((1920, 1080), (3840, 2160)):
    e = E.Encoder(w, h, gop=60, fixed_qp=40)
    fr = list(synth.s2_frames(w, h, 3))
    for f in fr:
        e.encode(*f)
    print("%dx%d: me %.1f us, subpel %.1f us, inter %.1f us" % (w, h, e.time_stage(E.STAGE_ME, 50) * 1e3, e.time_stage(E.STAGE_SUBPEL, 50) * 1e3, e.time_stage(E.STAGE_INTER, 50) * 1e3), flush=True)
    e.close()
