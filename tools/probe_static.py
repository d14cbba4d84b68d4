#!/usr/bin/env python3
"""Device time per stage on a still scene and on a half-still scene (top half never changes) at 1080p: what the per-band
"has work" flags of the deblocker buy.  MI355ENC_LIB selects another build for A/B."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ceracoder_amd import enc as E, synth
E.LIB_PATH = os.environ.get("MI355ENC_LIB", E.LIB_PATH)
w, h = 1920, 1080
fr = list(synth.s2_frames(w, h, 12))
for name, clip in (("still", [fr[0]] * 40), ("top half still", [(np.concatenate([fr[0][0][:544], f[0][544:]]), np.concatenate([fr[0][1][:272], f[1][272:]])) for f in fr] * 4)):
    e = E.Encoder(w, h, gop=300, fixed_qp=38, profile_events=1)
    for y, uv in clip:
        e.encode(y, uv)
    e.reset_stats()
    for y, uv in clip:
        e.encode(y, uv)
    st = e.stats()
    print("%s: deblock %.3f ms, me %.3f, fused P %.3f, total %.3f ms per picture" % (name, st.ms_deblock / st.n_deblock, st.ms_me / max(1, st.n_me),
          st.ms_inter / max(1, st.n_inter), st.ms_total_gpu / st.n_total_gpu))
    e.close()
