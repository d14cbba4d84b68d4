import os, sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from ceracoder_amd import enc as E, synth
w, h = 1920, 1080
clip = list(synth.s2_frames(w, h, 16))
bufs = [torch.from_numpy(np.concatenate([y.reshape(-1), uv.reshape(-1)])).cuda() for y, uv in clip]
pin = E.PinnedBuffer(16 * (w * h * 3 // 2))
pf = pin.array.reshape(16, h * 3 // 2, w)
for i, (y, uv) in enumerate(clip):
    pf[i, :h] = y; pf[i, h:] = uv
torch.cuda.synchronize()
for mode in ("device", "pinned"):
    e = E.Encoder(w, h, fps=60, gop=600, fixed_qp=32, pipeline_depth=2, exclusive=True)
    ts, tc = [], []
    for i in range(400):
        k = i % 30; k = k if k < 16 else 30 - k
        t0 = time.perf_counter()
        if mode == "device":
            p = bufs[k].data_ptr(); e.submit_device(p, w, p + w * h, w, pts=i)
        else:
            e.submit(pf[k, :h], pf[k, h:], pts=i)
        t1 = time.perf_counter()
        if e.pending > 2:
            e.collect(copy=False)
        t2 = time.perf_counter()
        if i > 100: ts.append(t1 - t0); tc.append(t2 - t1)
    while e.pending: e.collect(copy=False)
    e.close()
    print(mode, "submit %.1f us (p50 %.1f), collect %.1f us (p50 %.1f)" % (1e6 * np.mean(ts), 1e6 * np.median(ts), 1e6 * np.mean(tc), 1e6 * np.median(tc)))
