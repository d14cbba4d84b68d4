#!/usr/bin/env python3
"""dev tool: the columns at which the last deblocking launch of a free-running stream cut its bands in two (mi355enc_fetch 102: per band and plane the parts' counter, then
{cut column, epoch}); a cut equal to the picture width means the band was walked whole.    python tests/devtools/cut_probe.py [W H]"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ceracoder_amd import enc as E, synth
w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
clip = list(synth.s2_frames(w, h, 16))
bufs = [torch.from_numpy(np.concatenate([y.reshape(-1), uv.reshape(-1)])).cuda() for y, uv in clip]
torch.cuda.synchronize()
e = E.Encoder(w, h, fps=60, gop=600, bitrate_bps=6_000_000 * (w * h) // (1920 * 1080), pipeline_depth=2, exclusive=True, slices=None, slice_deblock=None)
nb = (e.mbh + 3) // 4
for rep in range(4):
    for i in range(20 + rep):
        k = (31 * rep + i) % 30
        p = bufs[k if k < 16 else 30 - k].data_ptr()
        e.submit_device(p, w, p + w * h, w, pts=i)
        if e.pending > 2:
            e.collect(copy=False)
    while e.pending:
        e.collect(copy=False)
    buf = np.zeros(6 * nb, np.uint32)
    assert e.L.mi355enc_fetch(e.h, 102, buf.ctypes.data_as(C.c_void_p), buf.nbytes) == 0
    cuts = buf[2 * nb:].reshape(nb, 2, 2)  # [band][plane][{cut, epoch}]
    print("picture %d: luma cuts per band %s | chroma %s | epochs %s" % (rep, list(cuts[:, 0, 0]), list(cuts[:, 1, 0]), sorted(set(cuts[:, :, 1].reshape(-1)))))
e.close()
