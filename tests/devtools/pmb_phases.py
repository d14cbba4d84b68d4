#!/usr/bin/env python3
"""dev tool: where a wave of the fused P stage (pmb_kernel) spends its cycles -- library built with `tools/build_variant.sh PMBPROF k_motion.hip -DPMB_PROF`
(cycle counts between the PMB_MARKs of pmb_mb, per macroblock, in ctx->dbrec):
    MI355ENC_LIB=ceracoder_amd/variants/libmi355enc_PMBPROF.so python tests/devtools/pmb_phases.py [W H [depth]]"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ceracoder_amd import enc as E, synth
E.LIB_PATH = os.environ.get("MI355ENC_LIB", E.LIB_PATH)
w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
depth = int(sys.argv[3]) if len(sys.argv) > 3 else 2
names = ["field + predictors + skip SAD arrive", "skip probe (prediction, transform, decision)", "refinement window -> LDS", "winner's SAD against the reference", "half-sample planes",
         "half-sample round", "quarter-sample round (SATD)", "final cost, intra test", "chroma prediction", "luma residual", "chroma residual + record"]
clip = list(synth.s2_frames(w, h, 16))
bufs = [torch.from_numpy(np.concatenate([y.reshape(-1), uv.reshape(-1)])).cuda() for y, uv in clip]
torch.cuda.synchronize()
e = E.Encoder(w, h, fps=60, gop=600, bitrate_bps=6_000_000 * (w * h) // (1920 * 1080), pipeline_depth=depth, exclusive=True, slices=None, slice_deblock=None)
def snap():
    buf = np.zeros((e.mbw * e.mbh, 16), np.uint32)
    assert e.L.mi355enc_fetch(e.h, 100, buf.ctypes.data_as(C.c_void_p), buf.nbytes) == 0
    return buf[:, :len(names)].astype(np.int64)
def run(n, base):
    for i in range(n):
        k = (base + i) % 30
        p = bufs[k if k < 16 else 30 - k].data_ptr()
        e.submit_device(p, w, p + w * h, w, pts=base + i)
        if e.pending > depth:
            e.collect(copy=False)
    while e.pending:
        e.collect(copy=False)
acc = np.zeros((len(names), 2), np.int64); path = []
for rep in range(6):
    run(20 + rep, 100 * rep)
    b = snap()  # the last picture's macroblocks
    acc[:, 0] += b.sum(axis=0); acc[:, 1] += (b > 0).sum(axis=0)
    path.append(b.sum(axis=1))
mbs = e.mbw * e.mbh * 6
print("%dx%d, %d P pictures sampled, depth %d; per phase: waves that reach the mark (share of all), mean cycles since the previous mark" % (w, h, 6, depth))
tot = 0.0
for k, nm in enumerate(names):
    cyc, cnt = int(acc[k][0]), int(acc[k][1])
    if cnt:
        print("  %-48s %5.1f %% of the macroblocks, %7.0f cycles each, %6.0f per macroblock of the picture" % (nm, 100.0 * cnt / mbs, cyc / cnt, cyc / mbs))
        tot += cyc / mbs
pp = np.concatenate(path)
print("  a wave's whole path: mean %.0f cycles, median %.0f, 90 %% %.0f, longest %.0f" % (pp.mean(), np.median(pp), np.percentile(pp, 90), pp.max()))
e.close()
