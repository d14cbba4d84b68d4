import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from ceracoder_amd import enc as E, synth
E.LIB_PATH = os.environ.get("MI355ENC_LIB", E.LIB_PATH)  # a library built with -DIB_PROF
for i4, qp in ((False, 40), (True, 24)):
    e = E.Encoder(1920, 1080, gop=60, fixed_qp=qp, i4x4=i4)
    fr = list(synth.s2_frames(1920, 1080, 1))
    e.stage_intra(np.pad(fr[0][0], ((0, 8), (0, 0)), mode="edge"), np.pad(fr[0][1], ((0, 4), (0, 0)), mode="edge"), qp)
    import ctypes as C
    raw = np.zeros(256, np.uint32)  # fetch 101 hands out 1024 bytes
    assert e.L.mi355enc_fetch(e.h, 101, raw.ctypes.data_as(C.c_void_p), raw.nbytes) == 0
    for band in range(2):
        for w in range(8):
            c, n, loop, ns = raw[(band * 8 + w) * 4:(band * 8 + w) * 4 + 4]
            print("i4=%s band %d wave %d (%s row %d): compute %d cycles over %d MBs = %.0f / MB; loop %d cycles / %d steps = %.0f / step"
                  % (i4, band, w, "chroma" if w & 1 else "luma", w >> 1, c, n, c / max(1, n), loop, ns, loop / ns))
    e.close()
