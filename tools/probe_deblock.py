#!/usr/bin/env python3
"""Per-step cost of the band deblocker on crafted records: all idle (bS = 0 everywhere) vs all intra."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ceracoder_amd import enc as E
E.LIB_PATH = os.environ.get("MI355ENC_LIB", E.LIB_PATH)  # e.g. an experiment build

for (w, h) in ((1920, 64), (1920, 1088)):
    e = E.Encoder(w, h, fixed_qp=30)
    n = e.mbw * e.mbh
    y = np.random.default_rng(0).integers(0, 256, (h, w), dtype=np.uint8)
    uv = np.random.default_rng(1).integers(0, 256, (h // 2, w), dtype=np.uint8)
    steps = e.mbw + e.mbh - 1  # x + y order of the band kernel (plus ~3 per band boundary)
    for name, typ, nz in (("idle", 1, 0), ("coded", 1, 0xFFFF), ("intra", 0, 0)):
        mbi = np.zeros(n, E.MBINFO_DTYPE)
        mbi["mb_type"], mbi["qp"], mbi["nzmask"] = typ, 30, nz
        e.stage_deblock(y, uv, mbi)
        t = e.time_stage(E.STAGE_DEBLOCK, 10)
        print("%dx%d %-6s %.3f ms  %.2f us/step" % (w, h, name, t, t * 1e3 / steps), flush=True)
    e.close()
