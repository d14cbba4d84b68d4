#!/usr/bin/env python3
"""dev tool: where does deblock mode N differ from the oracle?  python tests/devtools/dbg_deblock.py W H QP MODE"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from ceracoder_amd import enc as E
from oracle import oracle as O
if os.environ.get('MI355ENC_LIB'): E.LIB_PATH = os.environ['MI355ENC_LIB']
from tests.util import frames
w, h, qp, mode = (int(v) for v in sys.argv[1:5])
oe = O.Encoder(w, h, gop=60, threads=8)
e = E.Encoder((w + 15) // 16 * 16, (h + 15) // 16 * 16, fixed_qp=qp, deblock_mode=mode)
for n, (_, _, y, uv) in enumerate(frames(w, h, 2)):
    oe.encode(y, uv, qp)
    d_y, d_uv = e.stage_deblock(oe.prefilter_y, oe.prefilter_uv, oe.mbinfo)
    for name, d, o in (("Y", d_y, oe.recon_y), ("UV", d_uv, oe.recon_uv)):
        idx = np.argwhere(d != o)
        print("frame", n, name, "diffs", len(idx), [(int(a), int(b), int(d[a, b]), int(o[a, b])) for a, b in idx[:24]])
    if n == 0 and os.environ.get("DUMP"):
        r0, r1, c0, c1 = (int(v) for v in os.environ["DUMP"].split(","))
        print("qp", set(oe.mbinfo["qp"].tolist()), "types", set(oe.mbinfo["mb_type"].tolist()))
        for name, a in (("pre", oe.prefilter_uv), ("exp", oe.recon_uv), ("got", d_uv)):
            print(name); print(a[r0:r1, c0:c1])
