#!/usr/bin/env python3
"""dev tool: A/B of a schedule switch that the library reads from the environment when an encoder is opened (e.g. MI355ENC_NO_FIP; a new encoder per configuration), alternating runs inside ONE process so that
box-to-box and minute-to-minute noise cancels: N rounds of (unset, set), 600 pictures each, pipeline_depth 2, exclusive, CBR.
    python tests/devtools/ab_env.py VAR [W H [rounds]]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ceracoder_amd import enc as E, synth
var = sys.argv[1]
w, h = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080)
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 5
n = 600
clip = list(synth.s2_frames(w, h, 16))
bufs = [torch.from_numpy(np.concatenate([y.reshape(-1), uv.reshape(-1)])).cuda() for y, uv in clip]
torch.cuda.synchronize()
bps = 6_000_000 * (w * h) // (1920 * 1080)
res = {0: [], 1: []}
for rnd in range(rounds):
    for on in (0, 1):
        if on: os.environ[var] = "1"
        else: os.environ.pop(var, None)
        e = E.Encoder(w, h, fps=60, gop=60, bitrate_bps=bps, pipeline_depth=2, exclusive=True)
        nb = [0]
        def run(cnt, base):
            for i in range(cnt):
                k = (base + i) % 30
                p = bufs[k if k < 16 else 30 - k].data_ptr()
                e.submit_device(p, w, p + w * h, w, pts=base + i)
                if e.pending > 2:
                    nb[0] += e.collect(copy=False)[0]
            while e.pending:
                nb[0] += e.collect(copy=False)[0]
        run(60, 0)
        nb[0] = 0
        t0 = time.perf_counter(); run(n, 60); t = time.perf_counter() - t0
        st = e.stats()
        res[on].append(n / t)
        print("%s=%s: %.0f pictures/s, %d bytes, recoveries %d, safe level %d" % (var, "1" if on else "unset", n / t, nb[0], st.recoveries, st.safe_level), flush=True)
        e.close()
os.environ.pop(var, None)
a, b = np.array(res[0]), np.array(res[1])
print("%dx%d median unset %.0f, set %.0f: set / unset = %.3f; per-round ratios %s" % (w, h, np.median(a), np.median(b), np.median(b) / np.median(a), np.round(b / a, 3)))
