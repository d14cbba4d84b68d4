#!/usr/bin/env python3
"""dev tool: dbtprof.py's counters (cycles before / in / after the step barrier per role of deblock_rows3_kernel, bands 0 and 1) for the LAST launch of a free-running
stream (three pictures in flight, exclusive device) and, beside them, for the same launch run alone -- what stretches a step when the other stages' kernels run beside it.
Library built with tools/build_variant.sh PROF3 k_deblock.hip -DDBT_PROF:  MI355ENC_LIB=.../libmi355enc_PROF3.so python tests/devtools/dbtprof_stream.py"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ceracoder_amd import enc as E, synth
E.LIB_PATH = os.environ.get("MI355ENC_LIB", E.LIB_PATH)
w, h = 1920, 1080
clip = list(synth.s2_frames(w, h, 16))
bufs = [torch.from_numpy(np.concatenate([y.reshape(-1), uv.reshape(-1)])).cuda() for y, uv in clip]
torch.cuda.synchronize()
def dump(e, title):
    buf = np.zeros((e.mbw * e.mbh, 16), np.uint32)
    assert e.L.mi355enc_fetch(e.h, 100, buf.ctypes.data_as(C.c_void_p), buf.nbytes) == 0
    flat = buf.reshape(-1).astype(np.int64)
    print(title)
    for plane, off in (("luma", 0), ("chroma", 128)):
        for band in range(2):
            for r in range(4):
                out = []
                for role, name in enumerate(("F", "M", "S")):
                    v = flat[off + 64 * band + 16 * r + 4 * role: off + 64 * band + 16 * r + 4 * role + 4]
                    ns = max(1, int(v[3]) & 0xFFFF)
                    out.append("%s %4.0f /%4.0f /%4.0f = %4.0f" % (name, v[0] / ns, v[1] / ns, v[2] / ns, (v[0] + v[1] + v[2]) / ns) + ((" miss %d/%d" % (int(v[3]) >> 16, ns)) if role == 1 and r == 0 else ""))
                print("  %-6s band %d row %d  before / barrier / after = cycles a step: " % (plane, band, r) + " | ".join(out))
e = E.Encoder(w, h, fps=60, gop=600, fixed_qp=32, pipeline_depth=2, exclusive=True)
for i in range(150):
    k = i % 30
    p = bufs[k if k < 16 else 30 - k].data_ptr()
    e.submit_device(p, w, p + w * h, w, pts=i)
    if e.pending > 2:
        e.collect(copy=False)
while e.pending:
    e.collect(copy=False)
dump(e, "last launch of the free-running stream:")
e.time_stage(E.STAGE_DEBLOCK, 1)
dump(e, "the same picture's launch alone (time_stage):")
e.close()
