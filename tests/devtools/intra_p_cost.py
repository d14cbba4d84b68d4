#!/usr/bin/env python3
"""dev tool: what the intra macroblocks of P pictures cost the free-running 1080p stream: alternating runs with intra_in_p on / off (600 pictures, CBR, three in flight,
key-int 600 so that only P pictures count), pictures/s and bytes.
    python tests/devtools/intra_p_cost.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ceracoder_amd import enc as E, synth
w, h, n = 1920, 1080, 600
clip = list(synth.s2_frames(w, h, 16))
bufs = [torch.from_numpy(np.concatenate([y.reshape(-1), uv.reshape(-1)])).cuda() for y, uv in clip]
torch.cuda.synchronize()
res = {True: [], False: []}
for rnd in range(4):
    for ip in (True, False):
        e = E.Encoder(w, h, fps=60, gop=600, bitrate_bps=6_000_000, pipeline_depth=2, exclusive=True, intra_in_p=ip)
        nb = [0]
        def run(cnt, base):
            for i in range(cnt):
                k = (base + i) % 30
                p = bufs[k if k < 16 else 30 - k].data_ptr()
                e.submit_device(p, w, p + w * h, w, pts=base + i)
                if e.pending > 2: nb[0] += e.collect(copy=False)[0]
            while e.pending: nb[0] += e.collect(copy=False)[0]
        run(60, 0); nb[0] = 0
        t0 = time.perf_counter(); run(n, 60); t = time.perf_counter() - t0
        res[ip].append(n / t)
        print("intra_in_p %s: %.0f pictures/s (%.1f us a picture), %d bytes, recoveries %d" % (ip, n / t, 1e6 * t / n, nb[0], e.stats().recoveries), flush=True)
        e.close()
print("median with %.0f, without %.0f pictures/s: %.1f -> %.1f us a picture" % (np.median(res[True]), np.median(res[False]), 1e6 / np.median(res[True]), 1e6 / np.median(res[False])))
