#!/usr/bin/env python3
"""dev tool: many pictures on the fully overlapped schedule (pipeline_depth 2, exclusive_device) in several configurations; every device-side wait is bounded and a
bound that runs out is recovered from, so the thing to look at is `recoveries` (must stay 0) and the rate.  python tests/devtools/soak_waits.py [pictures]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ceracoder_amd import enc as E, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
for (w, h, kw) in ((1920, 1080, {}), (1920, 1080, {"aq": True}), (1920, 1080, {"partitions": True}), (1920, 1080, {"aq": True, "partitions": True, "intra_slices": 8}),
                   (1280, 720, {}), (1280, 720, {"partitions": True}), (640, 368, {}), (1920, 1080, {"transform8x8": True, "i8x8": True}), (3840, 2160, {}), (3840, 2160, {"aq": True, "partitions": True})):  # (720p and below: the intra rows ride in the deblocking launch)
    cnt = n if w < 3000 else n // 4
    clip = list(synth.s2_frames(w, h, 16))
    bufs = [torch.from_numpy(np.concatenate([y.reshape(-1), uv.reshape(-1)])).cuda() for y, uv in clip]
    torch.cuda.synchronize()
    e = E.Encoder(w, h, fps=60, gop=60, bitrate_bps=6_000_000 if w < 3000 else 20_000_000, pipeline_depth=2, exclusive=True, **kw)
    t0 = time.perf_counter()
    for i in range(cnt):
        k = i % 30
        p = bufs[k if k < 16 else 30 - k].data_ptr()
        e.submit_device(p, w, p + w * h, w, pts=i)
        if e.pending > 2:
            e.collect(copy=False)
    while e.pending:
        e.collect(copy=False)
    dt = time.perf_counter() - t0
    st = e.stats()
    print("%dx%d %s: %d pictures, %.0f pictures/s, recoveries %d, safe level %d, last error word %d" % (w, h, kw or "default", cnt, cnt / dt, st.recoveries, st.safe_level, st.last_error_word), flush=True)
    e.close()
    del bufs
