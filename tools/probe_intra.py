#!/usr/bin/env python3
"""dev tool: intra wavefront time per picture for both forms, with and without Intra_4x4.  python tools/probe_intra.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ceracoder_amd import enc as E, synth
E.LIB_PATH = os.environ.get('MI355ENC_LIB', E.LIB_PATH)
MODES = tuple(int(m) for m in os.environ.get('IMODES', '0,2,1').split(','))
for (w, h) in ((1920, 64), (1920, 1080)):
    for imode in MODES:
        for i4, qp in ((True, 24), (True, 40), (False, 40)):
            e = E.Encoder(w, h, gop=60, fixed_qp=qp, i4x4=i4, intra_mode=imode)
            fr = list(synth.s2_frames(w, h, 1))
            e.encode(*fr[0])
            t = e.time_stage(E.STAGE_INTRA, 5)
            mbi = e.fetch(E.FETCH_MBINFO)
            steps = e.mbw + e.mbh - 1
            print("%dx%d intra_mode %d i4x4=%s qp=%d: %.3f ms (%.2f us per x+y step), I4 macroblocks %.0f%%"
                  % (w, h, imode, i4, qp, t, t * 1e3 / steps, 100.0 * (mbi["mb_type"] == 2).mean()), flush=True)
            e.close()
