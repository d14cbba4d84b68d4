#!/usr/bin/env python3
"""dev tool: what an IDR picture costs the free-running stream, bubbles included: 600 pictures at fixed QP with key-int 60 and with key-int 600,
same clip, pipeline_depth 2; (t60 - t600) / 9 is the wall time one IDR picture adds over a P picture.
    python tests/devtools/gop_cost.py [qp [entropy-coding threads [aq] [partitions]]]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ceracoder_amd import enc as E, synth
E.LIB_PATH = os.environ.get("MI355ENC_LIB", E.LIB_PATH)
qp = int(sys.argv[1]) if len(sys.argv) > 1 else 32
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 0  # entropy-coding threads (0: the default)
aq = len(sys.argv) > 3 and "aq" in sys.argv[3:]  # adaptive quantisation
parts = "partitions" in sys.argv[3:]  # inter partitions
w, h, n = 1920, 1080, 600
clip = list(synth.s2_frames(w, h, 16))
bufs = [torch.from_numpy(np.concatenate([y.reshape(-1), uv.reshape(-1)])).cuda() for y, uv in clip]
torch.cuda.synchronize()
res = {}
for gop in (600, 60, 600, 60, 600, 60):
    e = E.Encoder(w, h, fps=60, gop=gop, fixed_qp=qp, pipeline_depth=2, exclusive=True, cavlc_threads=threads, aq=aq, partitions=parts)
    ent = {True: [], False: []}
    def col():
        a = e.stats().ms_entropy
        au, key, pts, q = e.collect(copy=False)
        ent[bool(key)].append((e.stats().ms_entropy - a, au))
    def run(cnt, base):
        for i in range(cnt):
            k = (base + i) % 30
            p = bufs[k if k < 16 else 30 - k].data_ptr()
            e.submit_device(p, w, p + w * h, w, pts=base + i)
            if e.pending > 2:
                col()
        while e.pending:
            col()
    run(60, 0)
    t0 = time.perf_counter(); run(n, 60); t = time.perf_counter() - t0
    res.setdefault(gop, []).append(t)
    st = e.stats()
    print("key-int %d: %.1f us per picture (%.0f pictures/s); recoveries %d, safe level %d; entropy coding: IDR %.0f us / %.0f KB, P %.0f us / %.1f KB" % (gop, t / n * 1e6, n / t, st.recoveries, st.safe_level,
          1e3 * np.mean([x[0] for x in ent[True]]), np.mean([x[1] for x in ent[True]]) / 1e3, 1e3 * np.mean([x[0] for x in ent[False]]), np.mean([x[1] for x in ent[False]]) / 1e3), flush=True)
    e.close()
d = (min(res[60]) - min(res[600])) / (n // 60 - 1)
print("an IDR picture adds %.0f us over a P picture at QP %d" % (d * 1e6, qp))
