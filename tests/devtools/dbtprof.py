#!/usr/bin/env python3
"""dev tool: cycles before / in / after the step barrier per role (filter, mover, storer) of deblock_rows3_kernel; library built with
-DDBT_PROF (tools/build_variant.sh PROF3 k_deblock.hip -DDBT_PROF):  MI355ENC_LIB=.../libmi355enc_PROF3.so python tests/devtools/dbtprof.py"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from ceracoder_amd import enc as E, synth
E.LIB_PATH = os.environ.get("MI355ENC_LIB", E.LIB_PATH)
w, h = 1920, 1080
for qp, npic in ((32, 1), (32, 3)):
    e = E.Encoder(w, h, gop=60, fixed_qp=qp)
    for f in list(synth.s2_frames(w, h, npic)):
        e.encode(*f)
    e.time_stage(E.STAGE_DEBLOCK, 1)
    buf = np.zeros((e.mbw * e.mbh, 16), np.uint32)
    assert e.L.mi355enc_fetch(e.h, 100, buf.ctypes.data_as(C.c_void_p), buf.nbytes) == 0
    flat = buf.reshape(-1).astype(np.int64)
    for plane, off in (("luma", 0), ("chroma", 128)):
        for band in range(2):
            for r in range(4):
                out = []
                for role, name in enumerate(("F", "M", "S")):
                    v = flat[off + 64 * band + 16 * r + 4 * role: off + 64 * band + 16 * r + 4 * role + 4]
                    ns = max(1, int(v[3]) & 0xFFFF)
                    out.append("%s %4.0f /%4.0f /%4.0f" % (name, v[0] / ns, v[1] / ns, v[2] / ns) + ((" miss %d/%d" % (int(v[3]) >> 16, ns)) if role == 1 and r == 0 else ""))
                print("%s %-6s band %d row %d  before/barrier/after: " % ("I" if npic == 1 else "P", plane, band, r) + " | ".join(out))
    e.close()
