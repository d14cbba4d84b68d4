#!/usr/bin/env python3
"""dev tool: host-side timestamps around an IDR picture in a free-running stream (pipeline_depth 2, fixed QP): when each submit() and collect()
returned, relative to the IDR picture's submit, averaged over the IDR pictures of the run.
    python tests/devtools/idr_gaps.py [qp]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ceracoder_amd import enc as E, synth
E.LIB_PATH = os.environ.get("MI355ENC_LIB", E.LIB_PATH)
qp = int(sys.argv[1]) if len(sys.argv) > 1 else 32
w, h, n, gop = 1920, 1080, 600, 60
clip = list(synth.s2_frames(w, h, 16))
bufs = [torch.from_numpy(np.concatenate([y.reshape(-1), uv.reshape(-1)])).cuda() for y, uv in clip]
torch.cuda.synchronize()
e = E.Encoder(w, h, fps=60, gop=gop, fixed_qp=qp, pipeline_depth=2, exclusive=True)
ts, tc = {}, {}
nc = 0
for i in range(n):
    k = i % 30
    p = bufs[k if k < 16 else 30 - k].data_ptr()
    e.submit_device(p, w, p + w * h, w, pts=i)
    ts[i] = time.perf_counter()
    if e.pending > 2:
        e.collect(copy=False); tc[nc] = time.perf_counter(); nc += 1
while e.pending:
    e.collect(copy=False); tc[nc] = time.perf_counter(); nc += 1
e.close()
idrs = [i for i in range(2 * gop, n - gop, gop)]
print("picture (0 = IDR): submit returned / collect returned, us after the IDR picture's submit returned; median over %d IDR pictures" % len(idrs))
for d in range(-4, 9):
    print("%3d  submit %8.0f   collect %8.0f" % (d, 1e6 * np.median([ts[i + d] - ts[i] for i in idrs]), 1e6 * np.median([tc[i + d] - ts[i] for i in idrs])))
