#!/usr/bin/env python3
"""dev tool: per-phase cycle sums of deblock_rows_kernel (library built with -DDBR_PROF: tools/build_variant.sh PROF k_deblock.hip -DDBR_PROF).
    MI355ENC_LIB=.../libmi355enc_PROF.so python tests/devtools/dbrprof.py"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from ceracoder_amd import enc as E, synth
E.LIB_PATH = os.environ.get("MI355ENC_LIB", E.LIB_PATH)
w, h = 1920, 1080
names = ["A prefetch issue", "records -> SGPR", "B vertical", "C barrier", "D horizontal", "E ring/finals", "vmcnt wait", "landing", "F stores"]
for qp, npic in ((32, 1), (32, 3)):
    e = E.Encoder(w, h, gop=60, fixed_qp=qp)
    for f in list(synth.s2_frames(w, h, npic)):
        e.encode(*f)
    e.time_stage(E.STAGE_DEBLOCK, 1)
    buf = np.zeros((e.mbw * e.mbh, 16), np.uint32)
    assert e.L.mi355enc_fetch(e.h, 100, buf.ctypes.data_as(C.c_void_p), buf.nbytes) == 0
    flat = buf.reshape(-1)
    for plane, off in (("luma", 0), ("chroma", 16)):
        v = flat[off:off + 10].astype(np.int64)
        ns = max(1, int(v[9]))
        print("%s picture, %s wave 1 of band 1: %d steps, %.0f cycles/step: " % ("I" if npic == 1 else "P", plane, ns, v[:9].sum() / ns) +
              ", ".join("%s %.0f" % (n, c / ns) for n, c in zip(names, v[:9])))
    e.close()
