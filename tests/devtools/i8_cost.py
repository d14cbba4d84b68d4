#!/usr/bin/env python3
"""dev tool: what Intra_8x8 buys and costs (High profile, 1080p): bytes and luma PSNR of the IDR pictures and wall time per picture, all-intra and key-int 60,
with cfg.i8x8 on and off, at a few QPs.
    python tests/devtools/i8_cost.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ceracoder_amd import enc as E, synth
w, h = 1920, 1080
clip = list(synth.s2_frames(w, h, 16))
bufs = [torch.from_numpy(np.concatenate([y.reshape(-1), uv.reshape(-1)])).cuda() for y, uv in clip]
torch.cuda.synchronize()
def psnr(a, b):
    d = a.astype(np.float64) - b.astype(np.float64)
    return 10 * np.log10(255.0 ** 2 / max(1e-9, np.mean(d * d)))
for qp in (22, 28, 34, 40):
    for gop, n in ((1, 240), (60, 600)):
        row = []
        for i8 in (True, False):
            e = E.Encoder(w, h, fps=60, gop=gop, fixed_qp=qp, pipeline_depth=2, exclusive=True, transform8x8=True, i8x8=i8)
            kb = []
            def run(cnt, base):
                for i in range(cnt):
                    k = (base + i) % 30
                    p = bufs[k if k < 16 else 30 - k].data_ptr()
                    e.submit_device(p, w, p + w * h, w, pts=base + i)
                    if e.pending > 2:
                        au, key, pts, q = e.collect(copy=False); kb.append((key, au))
                while e.pending:
                    au, key, pts, q = e.collect(copy=False); kb.append((key, au))
            run(60, 0)
            kb.clear()
            t0 = time.perf_counter(); run(n, 60); t = time.perf_counter() - t0
            # quality of one IDR picture (synchronous)
            y, uv = clip[0]
            au, key = e.encode(y, uv, pts=10**6, force_idr=True)
            ry = e.fetch(E.FETCH_RECON_Y)[:h, :w]
            st = e.stats()
            row.append((n / t, np.mean([b for k, b in kb if k]) / 1e3, np.mean([b for k, b in kb]) / 1e3, psnr(ry, y), key, st.recoveries))
            e.close()
        a, b = row
        print("QP %d key-int %3d: i8x8 on %.0f pictures/s, IDR %.1f KB, all %.1f KB, PSNR-Y %.2f (key %d) | off %.0f pictures/s, IDR %.1f KB, all %.1f KB, PSNR-Y %.2f (key %d) | IDR bytes %+.2f %%, PSNR %+.3f dB, speed %+.1f %%; recoveries %d %d"
              % (qp, gop, a[0], a[1], a[2], a[3], a[4], b[0], b[1], b[2], b[3], b[4], 100 * (a[1] / b[1] - 1), a[3] - b[3], 100 * (a[0] / b[0] - 1), a[5], b[5]), flush=True)
