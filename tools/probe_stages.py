#!/usr/bin/env python3
"""Stage timings on the GPU for a few geometries (dev tool): python tools/probe_stages.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ceracoder_amd import enc as E, synth

for (w, h) in ((1920, 64), (1920, 128), (1920, 256), (1920, 1088), (3840, 2160)):
    for mode in (0, 1):
        e = E.Encoder(w, h, gop=60, fixed_qp=int(os.environ.get("QP", "40")), deblock_mode=mode)
        fr = list(synth.s2_frames(w, h, 3))
        e.encode(*fr[0])
        t_i = e.time_stage(E.STAGE_DEBLOCK, 10)
        t_intra = e.time_stage(E.STAGE_INTRA, 5)
        e.encode(*fr[1]); e.encode(*fr[2])
        t_p = e.time_stage(E.STAGE_DEBLOCK, 10)
        t_me = e.time_stage(E.STAGE_ME, 20)
        t_inter = e.time_stage(E.STAGE_INTER, 20)
        steps = e.mbw + 2 * (e.mbh - 1) if mode else e.mbw + e.mbh - 1
        print("%4dx%-4d mode %d: deblock I %.3f ms  P %.3f ms (%d wavefront steps, %.2f / %.2f us per step) | intra %.3f ms | me %.1f us inter %.1f us"
              % (w, h, mode, t_i, t_p, steps, t_i * 1e3 / steps, t_p * 1e3 / steps, t_intra, t_me * 1e3, t_inter * 1e3), flush=True)
        e.close()

for (w, h) in ((1920, 1080), (3840, 2160)):
    e = E.Encoder(w, h, fixed_qp=30)
    P = e.mbw * e.mbh * 256
    for name, stg, bpp in (("I420", E.STAGE_CSC_I420, 3.0), ("YUY2", E.STAGE_CSC_YUY2, 3.5), ("UYVY", E.STAGE_CSC_UYVY, 3.5)):
        t = e.time_stage(stg, 200)
        print("%dx%d csc %s -> NV12: %.2f us, %.0f GB/s algorithmic (%.1f B/pixel)" % (w, h, name, t * 1e3, bpp * P / (t * 1e-3) / 1e9, bpp), flush=True)
    e.close()
