#!/usr/bin/env python3
"""dev tool: slices per I picture against the stream's speed and the IDR pictures' bytes (1080p, key-int 60, rate control at 6 Mbit/s and fixed QP 30).
    python tests/devtools/slices_cost.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ceracoder_amd import enc as E, synth
w, h, n = 1920, 1080, 600
clip = list(synth.s2_frames(w, h, 16))
bufs = [torch.from_numpy(np.concatenate([y.reshape(-1), uv.reshape(-1)])).cuda() for y, uv in clip]
torch.cuda.synchronize()
for kw in ({"fixed_qp": 30}, {"bitrate_bps": 6_000_000}):
    for rep in range(2):
        for ns in (1, 4, 6, 8, 12):
            e = E.Encoder(w, h, fps=60, gop=60, pipeline_depth=2, exclusive=True, intra_slices=ns, **kw)
            kb = []
            def run(cnt, base):
                for i in range(cnt):
                    k = (base + i) % 30
                    p = bufs[k if k < 16 else 30 - k].data_ptr()
                    e.submit_device(p, w, p + w * h, w, pts=base + i)
                    if e.pending > 2:
                        au, key, pts, q = e.collect(copy=False); kb.append((key, au))
                while e.pending:
                    au, key, pts, q = e.collect(copy=False); kb.append((key, au))
            run(60, 0)
            kb.clear()
            t0 = time.perf_counter(); run(n, 60); t = time.perf_counter() - t0
            st = e.stats()
            print("%s slices %2d (rows %d): %.0f pictures/s, IDR %.1f KB, all %.2f KB; recoveries %d" % (kw, ns, e.slice_rows, n / t, np.mean([b for k, b in kb if k]) / 1e3, np.mean([b for k, b in kb]) / 1e3, st.recoveries), flush=True)
            e.close()
