#!/usr/bin/env python3
"""dev tool: long CBR run with bitrate changes; every access unit is decoded by the independent decoder and the
decoder's picture must equal the encoder's reconstruction every `check` pictures.  python tests/devtools/soak.py W H N [t8 [i8]] [depth2]
(t8: High profile; i8: Intra_8x8 as well, key-int 15 so that I pictures are a good part of the run; depth2: three pictures in flight on an exclusive device --
the schedule with every device-side wait -- instead of two on a shared one)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from ceracoder_amd import enc as E, synth
from oracle import oracle as O

w, h, n = (int(v) for v in sys.argv[1:4])
t8 = len(sys.argv) > 4 and sys.argv[4] == "t8"
i8 = t8 and len(sys.argv) > 5 and sys.argv[5] == "i8"
d2 = "depth2" in sys.argv[4:]
e = E.Encoder(w, h, fps=60, gop=15 if i8 else 60, bitrate_bps=6_000_000, pipeline_depth=2 if d2 else 1, exclusive=d2, transform8x8=t8, i8x8=i8)
dec = O.Decoder()
clip = list(synth.s2_frames(w, h, 24))
sizes, t0, pend = [], time.time(), []
def take():
    au, key, pts, qp = e.collect()
    sizes.append(len(au))
    y, uv = dec.decode(au)
    i = pend.pop(0)
    if e.pending == 0 and (i % 50 == 0 or i == n - 1):
        assert np.array_equal(y, e.fetch(E.FETCH_RECON_Y)) and np.array_equal(uv, e.fetch(E.FETCH_RECON_UV)), "drift at picture %d" % i
for i in range(n):
    if i == n // 3: e.set_bitrate(1_500_000)
    if i == 2 * n // 3: e.set_bitrate(12_000_000)
    k = i % 46
    y, uv = clip[k if k < 24 else 46 - k]
    e.submit(y, uv, pts=i)
    pend.append(i)
    if e.pending > (2 if d2 else 1): take()
    if i % 50 == 49:
        while e.pending: take()
while e.pending: take()
s = np.array(sizes, float)
third = n // 3
st = e.stats()
assert st.recoveries == 0 and st.safe_level == 0, (st.recoveries, st.safe_level)
print("ok %dx%d %d pictures in %.1f s, no device-wait recovery; Mbit/s per third: %.2f %.2f %.2f (targets 6, 1.5, 12)" % (
    w, h, n, time.time() - t0, *(s[a:b].sum() * 8 * 60 / (b - a) / 1e6 for a, b in ((60, third), (third + 60, 2 * third), (2 * third + 60, n)))))
