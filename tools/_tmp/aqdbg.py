import sys; sys.path.insert(0, '.')
import numpy as np
from ceracoder_amd import enc as E, synth
from oracle import oracle as O
from tests.util import half_static_clip
w, h = 1920, 1080
clip = half_static_clip(w, h, 2, (h // 3) & ~15)
for depth in (0,):
    e = E.Encoder(w, h, gop=4, fixed_qp=30, aq=True, pipeline_depth=depth, exclusive=True)
    oe = O.Encoder(w, h, gop=4, threads=8, aq=True)
    y, uv = clip[0]
    au, _ = e.encode(y, uv, pts=0)
    rau, _ = oe.encode(y, uv, 30)
    m = e.fetch(E.FETCH_MBINFO); om = oe.mbinfo
    lv = e.fetch(E.FETCH_LEVELS); olv = oe.levels
    print("au equal", au == rau, len(au), len(rau))
    for f in ("mb_type", "qp", "nzmask", "i16_mode", "chroma_mode"):
        d = np.nonzero(m[f] != om[f])[0]
        print(f, len(d), d[:10], [(int(m[f][i]), int(om[f][i])) for i in d[:5]])
    dl = np.nonzero((lv != olv).any(axis=1))[0]
    print("levels differ in MBs", len(dl), dl[:10])
    print("recon equal", np.array_equal(e.fetch(E.FETCH_RECON_Y), oe.recon_y))
    e.close()
