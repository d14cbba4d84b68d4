#!/usr/bin/env python3
"""dev tool: A/B of two builds of the library (tools/build_variant.sh), alternating child processes so that minute-to-minute noise cancels: N rounds of (A, B),
600 pictures each after a warm-up GOP, pipeline_depth 2, exclusive, CBR.  One child = one library (a process loads one build).
    python tests/devtools/ab_lib.py LIB_A LIB_B [W H [rounds]]        (LIB_x may also be VAR=value: the default build with that environment variable)"""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import os, sys, time
sys.path.insert(0, %r)
import numpy as np, torch
from ceracoder_amd import enc as E, synth
w, h, n = int(sys.argv[1]), int(sys.argv[2]), 600
clip = list(synth.s2_frames(w, h, 16))
bufs = [torch.from_numpy(np.concatenate([y.reshape(-1), uv.reshape(-1)])).cuda() for y, uv in clip]
torch.cuda.synchronize()
out = []
for rep in range(int(sys.argv[3])):
    e = E.Encoder(w, h, fps=60, gop=60, bitrate_bps=6_000_000 * (w * h) // (1920 * 1080), pipeline_depth=2, exclusive=True)
    nb = [0]
    def run(cnt, base):
        for i in range(cnt):
            k = (base + i) %% 30
            p = bufs[k if k < 16 else 30 - k].data_ptr()
            e.submit_device(p, w, p + w * h, w, pts=base + i)
            if e.pending > 2: nb[0] += e.collect(copy=False)[0]
        while e.pending: nb[0] += e.collect(copy=False)[0]
    run(60, 0); nb[0] = 0
    t0 = time.perf_counter(); run(n, 60); t = time.perf_counter() - t0
    st = e.stats()
    out.append((n / t, nb[0], st.recoveries))
    e.close()
print("RESULT", " ".join("%%.1f:%%d:%%d" %% x for x in out))
''' % ROOT
la, lb = sys.argv[1], sys.argv[2]
w, h = (sys.argv[3], sys.argv[4]) if len(sys.argv) > 4 else ("1920", "1080")
rounds = int(sys.argv[5]) if len(sys.argv) > 5 else 3
res = {la: [], lb: []}
for rnd in range(rounds):
    for lib in (la, lb):
        env = dict(os.environ)
        if "=" in lib:  # "VAR=value" instead of a library: the default build with that variable set ("NONE=0": nothing set) -- for switches read when the encoder is opened
            k, v = lib.split("=", 1)
            env[k] = v
        else:
            env["MI355ENC_LIB"] = os.path.join(ROOT, lib)
        r = subprocess.run([sys.executable, "-c", CHILD, w, h, "2"], env=env, capture_output=True, text=True, timeout=300)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
        if not line:
            print("child failed:", r.stderr[-2000:]); sys.exit(1)
        vals = [tuple(float(v) for v in x.split(":")) for x in line[0].split()[1:]]
        res[lib] += [v[0] for v in vals]
        print(lib, ["%.0f pictures/s, %d bytes, %d recoveries" % v for v in vals], flush=True)
a, b = np.array(res[la]), np.array(res[lb])
print("%sx%s median A %.0f, B %.0f: B / A = %.3f" % (w, h, np.median(a), np.median(b), np.median(b) / np.median(a)))
