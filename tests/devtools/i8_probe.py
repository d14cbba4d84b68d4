#!/usr/bin/env python3
"""dev tool: key-int 60 at QP 22, i8x8 off / on alternating: pictures/s, host entropy time, host wait time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ceracoder_amd import enc as E, synth
w, h, n = 1920, 1080, 600
qp = int(sys.argv[1]) if len(sys.argv) > 1 else 22
clip = list(synth.s2_frames(w, h, 16))
bufs = [torch.from_numpy(np.concatenate([y.reshape(-1), uv.reshape(-1)])).cuda() for y, uv in clip]
torch.cuda.synchronize()
for i8 in (False, True, False, True):
    e = E.Encoder(w, h, fps=60, gop=60, fixed_qp=qp, pipeline_depth=2, exclusive=True, transform8x8=True, i8x8=i8)
    tk = []
    def run(cnt, base):
        for i in range(cnt):
            k = (base + i) % 30
            p = bufs[k if k < 16 else 30 - k].data_ptr()
            e.submit_device(p, w, p + w * h, w, pts=base + i)
            if e.pending > 2:
                t = time.perf_counter(); au, key, pts, q = e.collect(copy=False); tk.append((key, au, time.perf_counter() - t, time.perf_counter()))
        while e.pending:
            t = time.perf_counter(); au, key, pts, q = e.collect(copy=False); tk.append((key, au, time.perf_counter() - t, time.perf_counter()))
    run(60, 0)
    tk.clear(); e.reset_stats()
    t0 = time.perf_counter(); run(n, 60); t = time.perf_counter() - t0
    st = e.stats()
    ends = [x[3] for x in tk]
    gaps = np.diff(ends) * 1e6
    keys = [i for i, x in enumerate(tk) if x[0]]
    around = [gaps[max(0, k - 3):k + 4].round() for k in keys[1:3]]
    print("i8x8 %d: %.0f pictures/s; entropy %.1f ms/GOP, wait %.1f ms/GOP; recoveries %d; collect-to-collect gaps around two IDR pictures (us): %s" % (i8, n / t, st.ms_entropy / 10, st.ms_wait / 10, st.recoveries, around), flush=True)
    e.close()
