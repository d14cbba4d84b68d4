#!/usr/bin/env python3
"""inter_kernel time against the kind of motion vectors it is given (whole-sample: fast path; fractional: 9x9 gather + 6-tap)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ceracoder_amd import enc as E, synth
for (w, h) in ((1920, 1080), (3840, 2160)):
    for sub in (False, True):
        e = E.Encoder(w, h, gop=60, fixed_qp=40, subpel=sub)
        fr = list(synth.s2_frames(w, h, 3))
        for f in fr:
            e.encode(*f)
        print("%dx%d subpel=%s: me %.1f us, subpel %.1f us, inter %.1f us" % (w, h, sub, e.time_stage(E.STAGE_ME, 20) * 1e3,
              e.time_stage(E.STAGE_SUBPEL, 20) * 1e3 if sub else 0.0, e.time_stage(E.STAGE_INTER, 20) * 1e3), flush=True)
        e.close()
