#!/usr/bin/env python3
"""dev tool: device-side timeline of the P-picture stages in a free-running stream (no profiler in the way of the host).  Library built
with tools/build_tlprof.sh (wall-clock marks per picture epoch in ctx->dbrec):
    MI355ENC_LIB=ceracoder_amd/variants/libmi355enc_TL.so python tests/devtools/timeline.py [depth] [pictures]"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ceracoder_amd import enc as E, synth
E.LIB_PATH = os.environ.get("MI355ENC_LIB", E.LIB_PATH)
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n = int(sys.argv[2]) if len(sys.argv) > 2 else 150
w, h = (int(os.environ.get("TL_W", "1920")), int(os.environ.get("TL_H", "1080")))  # TL_W / TL_H: another picture size
clip = list(synth.s2_frames(w, h, 16))
bufs = [torch.from_numpy(np.concatenate([y.reshape(-1), uv.reshape(-1)])).cuda() for y, uv in clip]
torch.cuda.synchronize()
slices, sdb = int(os.environ.get("TL_SLICES", "-1")), int(os.environ.get("TL_SDB", "-1"))  # TL_SLICES / TL_SDB: cfg.slices / cfg.slice_deblock (-1: the library's default)
e = E.Encoder(w, h, fps=60, gop=600, bitrate_bps=6_000_000 * (w * h) // (1920 * 1080), pipeline_depth=depth, exclusive=True, slices=None if slices < 0 else slices,
              slice_deblock=None if sdb < 0 else bool(sdb), intra_in_p=int(os.environ.get("TL_IP", "1")))
for i in range(n):
    k = i % 30
    p = bufs[k if k < 16 else 30 - k].data_ptr()
    e.submit_device(p, w, p + w * h, w, pts=i)
    if e.pending > depth:
        e.collect(copy=False)
while e.pending:
    e.collect(copy=False)
buf = np.zeros((e.mbw * e.mbh, 8), np.uint64)
assert e.L.mi355enc_fetch(e.h, 100, buf.ctypes.data_as(C.c_void_p), buf.nbytes) == 0
t = buf.reshape(-1)[:64 * 16].reshape(64, 16).astype(np.int64)
rows = sorted((r for r in t if r[7] > 0 and r[0] > 0), key=lambda r: r[7])[-40:]
names = ["me0", "me1", "pmb0", "gate", "pmb1", "ip0", "ip1", "db0", "db1", "b0go", "b0end", "bMgo", "bMend", "bLgo"]  # b0 / bM / bL: first, middle, last luma band: past its row wait / done
print("per picture, us relative to the start of its deblocking (100 MHz clock): " + " ".join("%7s" % x for x in names) + "   db0 - previous db1 | period")
prev = None
for r in rows:
    us = [(int(r[k]) - int(r[7])) / 100.0 if r[k] > 0 else float('nan') for k in range(14)]
    gap = (int(r[7]) - int(prev[8])) / 100.0 if prev is not None else float("nan")
    per = (int(r[7]) - int(prev[7])) / 100.0 if prev is not None else float("nan")
    pro = [(int(r[k]) - int(r[9])) / 100.0 for k in (14, 15)]  # band 0's prologue: its own records computed (stores under way), step loop about to start -- us after its row wait
    print(" " * 73 + " ".join("%7.1f" % x for x in us) + "   %7.1f | %7.1f | band 0 prologue: records +%.1f, loop starts +%.1f, loop %.1f us" % (gap, per, pro[0], pro[1], (int(r[10]) - int(r[15])) / 100.0))
    prev = r
last = rows[-1]
e.time_stage(E.STAGE_DEBLOCK, 1)  # the last picture's launch once more, alone: its marks land in the same slots
assert e.L.mi355enc_fetch(e.h, 100, buf.ctypes.data_as(C.c_void_p), buf.nbytes) == 0
t2 = buf.reshape(-1)[:64 * 16].reshape(64, 16).astype(np.int64)
for r in t2:
    if r[7] > last[7]:
        print("the same launch alone: band 0 records +%.1f, loop starts +%.1f, loop %.1f us; launch %.1f us" % ((int(r[14]) - int(r[9])) / 100.0, (int(r[15]) - int(r[9])) / 100.0, (int(r[10]) - int(r[15])) / 100.0, (int(r[8]) - int(r[7])) / 100.0))
e.close()
