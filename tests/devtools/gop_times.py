#!/usr/bin/env python3
"""dev tool: wall time of each GOP of a free-running stream, for two encoders opened one after the other in one process (is the first stream with IDR
pictures slower than the second, and where?).  python tests/devtools/gop_times.py [qp]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ceracoder_amd import enc as E, synth
E.LIB_PATH = os.environ.get("MI355ENC_LIB", E.LIB_PATH)
qp = int(sys.argv[1]) if len(sys.argv) > 1 else 32
w, h, gop, ngop = 1920, 1080, 60, 11
clip = list(synth.s2_frames(w, h, 16))
bufs = [torch.from_numpy(np.concatenate([y.reshape(-1), uv.reshape(-1)])).cuda() for y, uv in clip]
torch.cuda.synchronize()
for run in range(3):
    e = E.Encoder(w, h, fps=60, gop=gop, fixed_qp=qp, pipeline_depth=2, exclusive=True)
    marks = []
    for i in range(gop * ngop):
        k = i % 30
        p = bufs[k if k < 16 else 30 - k].data_ptr()
        e.submit_device(p, w, p + w * h, w, pts=i)
        if e.pending > 2:
            e.collect(copy=False)
        if i % gop == gop - 1:
            marks.append(time.perf_counter())
    while e.pending:
        e.collect(copy=False)
    st = e.stats()
    print("encoder %d: ms per GOP of 60:" % run, " ".join("%.2f" % (1e3 * (b - a)) for a, b in zip(marks, marks[1:])), "| recoveries", st.recoveries, flush=True)
    e.close()
