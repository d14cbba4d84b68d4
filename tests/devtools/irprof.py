import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ctypes as C
import numpy as np
from ceracoder_amd import enc as E, synth
E.LIB_PATH = os.environ.get("MI355ENC_LIB", E.LIB_PATH)  # a library built with -DIR_PROF (tools/build_variant.sh IRPROF k_intra.hip -DIR_PROF)
for i4, qp in ((True, 24),):
    e = E.Encoder(1920, 1080, gop=60, fixed_qp=qp, i4x4=i4)
    fr = list(synth.s2_frames(1920, 1080, 1))
    e.stage_intra(np.pad(fr[0][0], ((0, 8), (0, 0)), mode="edge"), np.pad(fr[0][1], ((0, 4), (0, 0)), mode="edge"), qp)
    raw = np.zeros(256, np.uint32)  # fetch 101 hands out 1024 bytes
    assert e.L.mi355enc_fetch(e.h, 101, raw.ctypes.data_as(C.c_void_p), raw.nbytes) == 0
    LW = int(os.environ.get('IR_LW', '3'))
    for row in range(2):
        for w in list(range(LW)) + [3]:
            o = raw[(row * 4 + w) * 16:(row * 4 + w) * 16 + 16]
            if w < 3 and w < LW:
                print("i4=%s qp %d row %d luma wave %d: total %d cycles; I4 %d MBs %.0f cycles each; I16 %d MBs %.0f each; waits: top %d left %d source %d record slot %d; publishing %d"
                      % (i4, qp, row, w, o[0], o[2], o[1] / max(1, o[2]), o[4], o[3] / max(1, o[4]), o[5], o[6], o[8], o[9], o[7]))
            else:
                print("i4=%s qp %d row %d chroma wave: total %d cycles; %d MBs %.0f cycles each; waits: top %d source %d record slot %d" % (i4, qp, row, o[0], o[2], o[1] / max(1, o[2]), o[5], o[8], o[9]))
    e.close()
