"""GPU: adversarial schedules for the kernels that wait on the device for each other.  Debug builds of the library (built by
__graft_entry__.build(): tools/build_variant.sh) stretch exactly the windows the waits exist for:
  ADVBAND  one deblocking band (-DDBG_DELAY_BAND=3) keeps its lines in its XCD's L2 for ~0.3 ms before the release + band-done word, long after the
           band below it has published -- pmb_kernel<GATED> of the next picture must wait for EVERY band its reference window touches, not only
           the lowest (the round-2 fix; before it the stream differed from the in-order one in a macroblock every few hundred pictures);
  ADVIP    intra_p_kernel (-DDBG_DELAY_IP) publishes every progress word ~20 us late -- the band deblocker beside it must stop at every intra
           macroblock (and the one left of it) until the word says it is final.
Each runs once, in a child process (the library is chosen at load time), and its access units must equal the in-order path's bit for bit."""
import hashlib
import json
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = textwrap.dedent("""
    import hashlib, json, sys
    sys.path.insert(0, %r)
    from ceracoder_amd import enc as E, synth
    import numpy as np
    w, h, n, depth, exclusive = 1920, 1080, 14, int(sys.argv[1]), bool(int(sys.argv[2]))
    # a window sliding 8 lines per picture down and back over a taller S2 clip: vertical vectors of +-8 samples, so that a macroblock row's
    # prediction reaches well into the row above it (the lines another deblocking band wrote)
    big = list(synth.s2_frames(w, h + 64, n))
    offs = [8 * (i if i < 8 else 14 - i) for i in range(n)]
    clip = [(np.ascontiguousarray(y[o:o + h]), np.ascontiguousarray(uv[o // 2:o // 2 + h // 2])) for (y, uv), o in zip(big, offs)]
    e = E.Encoder(w, h, gop=30, fixed_qp=30, pipeline_depth=depth, exclusive=exclusive, scenecut=False)
    m, out = hashlib.sha256(), []
    for i, (y, uv) in enumerate(clip):
        e.submit(y, uv, pts=i)
        if e.pending > depth:
            out.append(e.collect()[0])
    while e.pending:
        out.append(e.collect()[0])
    for au in out:
        m.update(au)
    st = e.stats()
    print(json.dumps({"digest": m.hexdigest(), "recoveries": int(st.recoveries), "bytes": sum(len(a) for a in out)}))
    e.close()
""") % ROOT


def run_child(tmp_path, lib, depth, exclusive):
    script = tmp_path / "adv.py"
    script.write_text(CHILD)
    env = dict(os.environ)
    if lib:
        path = os.path.join(ROOT, "ceracoder_amd", "variants", "libmi355enc_%s.so" % lib)
        if not os.path.exists(path):
            pytest.fail("%s not built: run __graft_entry__.build()" % path)
        env["MI355ENC_LIB"] = path
    else:
        env.pop("MI355ENC_LIB", None)
    r = subprocess.run([sys.executable, str(script), str(depth), str(int(exclusive))], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


@pytest.mark.parametrize("lib,depth", [("ADVBAND", 2), ("ADVIP", 0), ("ADVIP", 2)])
def test_adversarial_schedule_gives_the_in_order_stream(tmp_path, lib, depth):
    ref = run_child(tmp_path, None, 0, False)         # the shipped library, every kernel in stream order
    adv = run_child(tmp_path, lib, depth, True)       # the stretched window, with the kernels waiting for each other on the device
    assert adv["recoveries"] == 0                      # (a wait that ran into its bound would have been recovered from: not what is tested here)
    assert adv["digest"] == ref["digest"], (adv, ref)
