#!/usr/bin/env python3
"""Per-picture latency of the synchronous path (pipeline_depth 0, host NV12 in -> access unit out) against the number of
entropy-coding threads, back to back and with the 16.7 ms gaps of a live 60 fps source (sleeping workers must be woken)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ceracoder_amd import enc as E, synth
E.LIB_PATH = os.environ.get("MI355ENC_LIB", E.LIB_PATH)  # A/B against another build
w, h = 1920, 1080
fr = [np.concatenate([y, uv]) for y, uv in synth.s2_frames(w, h, 24)]
for gap_ms in (0.0, 16.7):
    for thr in ((0,) if os.environ.get("QUICK") else (1, 2, 4, 0)):
        e = E.Encoder(w, h, fps=60, gop=60, bitrate_bps=6_000_000, pipeline_depth=0, cavlc_threads=thr)
        lat = []
        n = 150 if gap_ms == 0 else 60
        for i in range(n):
            f = fr[i % len(fr)]
            if gap_ms:
                time.sleep(gap_ms / 1e3)
            t0 = time.perf_counter()
            e.encode(f[:h], f[h:], pts=i)
            lat.append((time.perf_counter() - t0) * 1e3)
        st = e.stats()
        lat = np.sort(np.array(lat[20:]))
        print("gap %4.1f ms threads %d (resolved %d): p50 %.3f ms p95 %.3f ms; host cavlc %.3f ms/picture, wait %.3f" % (
            gap_ms, thr, st.cavlc_threads, lat[len(lat) // 2], lat[int(len(lat) * 0.95)], st.ms_entropy / st.frames, st.ms_wait / st.frames))
        e.close()
