#!/usr/bin/env python3
"""dev tool: the same 1080p CBR stream, pipeline_depth 2, encoded several times -- with and without exclusive_device (the gated P stage) -- must be the
same bytes every time (rate control sees the same sizes in the same order).  Prints the digest of each run and the first differing access unit."""
import sys, hashlib; sys.path.insert(0, '.')
import numpy as np, torch
from ceracoder_amd import enc as E, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 600
w, h = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080)
clip = list(synth.s2_frames(w, h, 16 if w < 2000 else 8))
NC = len(clip)
bufs = [torch.from_numpy(np.concatenate([y.reshape(-1), uv.reshape(-1)])).cuda() for y, uv in clip]
torch.cuda.synchronize()
def run(exclusive, sample):
    e = E.Encoder(w, h, fps=60, gop=60, bitrate_bps=6_000_000 * (4 if w > 2000 else 1), pipeline_depth=2, exclusive=exclusive, profile_events=sample)
    out = []
    for i in range(n):
        k = i % (2 * NC - 2)
        p = bufs[k if k < NC else 2 * NC - 2 - k].data_ptr()
        e.submit_device(p, w, p + w * h, w, pts=i)
        if e.pending > 2:
            out.append(bytes(e.collect()[0]))
    while e.pending:
        out.append(bytes(e.collect()[0]))
    e.close()
    return out
runs = [("exclusive", True, 0), ("exclusive again", True, 0), ("exclusive, sampled 29", True, 29), ("shared", False, 0), ("shared again", False, 0)]
ref = None
for name, ex, sample in runs:
    out = run(ex, sample)
    d = hashlib.sha256(b"".join(out)).hexdigest()[:16]
    if ref is None:
        ref = out
    diff = next((i for i, (a, b) in enumerate(zip(ref, out)) if a != b), None)
    print("%-24s %s  bytes %d  first differing access unit vs the first run: %s" % (name, d, sum(len(x) for x in out), diff), flush=True)
