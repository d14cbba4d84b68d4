#!/usr/bin/env python3
"""GPU: the rate-PSNR yardstick -- 1080p60, GOP 60, CBR at 2 / 4 / 6 / 9 / 12 Mbit/s on S1 (videotestsrc-like
colour bars) and S2 (ME stress); three GOPs coded, the last two measured: produced bitrate, mean QP, mean PSNR of Y / Cb / Cr over
ALL measured pictures (encoder reconstruction vs source; the reconstruction equals what a decoder outputs -- tests/).
    python tools/rd_table.py [out.json]      -> JSON on stdout / in out.json, a markdown table on stderr"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ceracoder_amd import enc as E, synth

w, h, fps, gop = 1920, 1080, 60, 60
out = {"geometry": "%dx%d@%d, GOP %d, CBR" % (w, h, fps, gop), "points": []}
for kind, gen in (("S1", lambda: synth.s1_frames(w, h, 16)), ("S2", lambda: synth.s2_frames(w, h, 24)), ("S4", lambda: synth.s4_frames(w, h, 24))):
    clip = list(gen())
    for bps in (2_000_000, 4_000_000, 6_000_000, 9_000_000, 12_000_000):
        e = E.Encoder(w, h, fps=fps, gop=gop, bitrate_bps=bps, pipeline_depth=0)
        ps, nbytes, qps, skips = [], 0, [], 0
        for i in range(3 * gop):
            k = i % (2 * len(clip) - 2)
            y, uv = clip[k if k < len(clip) else 2 * len(clip) - 2 - k]
            e.submit(y, uv, pts=i)
            au, key, pts, qp = e.collect(copy=False)
            if i >= gop:
                nbytes += au; qps.append(qp); skips += e.last_drop == 255
                ry, ruv = e.fetch(E.FETCH_RECON_Y)[:h, :w], e.fetch(E.FETCH_RECON_UV)[:h // 2, :w]
                ps.append((synth.psnr(y, ry), synth.psnr(uv[:, 0::2], ruv[:, 0::2]), synth.psnr(uv[:, 1::2], ruv[:, 1::2])))
        e.close()
        pm = np.mean(np.array(ps), axis=0)
        out["points"].append({"clip": kind, "setpoint_bps": bps, "bitrate_bps": round(nbytes * 8 * fps / (2 * gop)), "mean_qp": round(float(np.mean(qps)), 2),
                              "psnr_y": round(float(pm[0]), 2), "psnr_u": round(float(pm[1]), 2), "psnr_v": round(float(pm[2]), 2),
                              "psnr_y_min": round(float(np.min(np.array(ps)[:, 0])), 2), "skip_pictures": int(skips), "pictures": 2 * gop})
js = json.dumps(out, indent=1)
print(js)
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write(js + "\n")
sys.stderr.write("| clip | setpoint Mbit/s | produced Mbit/s | mean QP | PSNR-Y | PSNR-U | PSNR-V | min PSNR-Y |\n|---|---|---|---|---|---|---|---|\n")
for p in out["points"]:
    sys.stderr.write("| %s | %.0f | %.2f | %.1f | %.2f | %.2f | %.2f | %.2f |\n" % (p["clip"], p["setpoint_bps"] / 1e6, p["bitrate_bps"] / 1e6, p["mean_qp"], p["psnr_y"], p["psnr_u"], p["psnr_v"], p["psnr_y_min"]))
