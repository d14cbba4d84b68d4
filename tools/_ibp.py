import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, ctypes as C
from ceracoder_amd import enc as E, synth
E.LIB_PATH = os.environ["MI355ENC_LIB"]
for i4, qp in ((False, 40),):
    e = E.Encoder(1920, 1080, gop=60, fixed_qp=qp, i4x4=i4)
    fr = list(synth.s2_frames(1920, 1080, 1))
    e.stage_intra(np.pad(fr[0][0], ((0, 8), (0, 0)), mode="edge"), np.pad(fr[0][1], ((0, 4), (0, 0)), mode="edge"), qp)
    raw = np.zeros(256, np.uint32)
    assert e.L.mi355enc_fetch(e.h, 101, raw.ctypes.data_as(C.c_void_p), raw.nbytes) == 0
    for band in range(2):
        for w in range(4):
            c, n, loop, ns = raw[(band * 8 + w) * 4:(band * 8 + w) * 4 + 4]
            c0, c1, c2, c4 = raw[64 + (band * 8 + w) * 4:64 + (band * 8 + w) * 4 + 4]
            print("i4=%s band %d wave %d: per step: total %d = barrier %d + land/prefetch %d + mover %d + compute %d + tail %d" % (
                i4, band, w, loop / ns, c0 / ns, c1 / ns, c2 / ns, c / ns, c4 / ns))
    e.close()
