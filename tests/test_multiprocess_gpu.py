"""GPU: the reference's process model -- one process per stream -- on the one GPU of the test box: two fresh
child processes, one encoder each, bracketed by the gloo barrier pair of ceracoder_amd/multistream.py (what bench.py --gpus N
does with one GPU per rank).  Streams are independent: both must equal the single-process result bit for bit."""
import hashlib
import re
import json
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import hashlib, json, sys
    sys.path.insert(0, %r)
    from ceracoder_amd import enc as E, synth
    from ceracoder_amd.multistream import Ranks
    r = Ranks()
    w, h, n = 640, 368, 12
    clip = list(synth.s2_frames(w, h, n))
    e = E.Encoder(w, h, gop=6, fixed_qp=30, device_id=0)   # both ranks share the box's only GPU
    def run():
        m = hashlib.sha256()
        for i, (y, uv) in enumerate(clip):
            m.update(e.encode(y, uv, pts=i)[0])
        return m.hexdigest()
    dt, digest = r.timed(run)
    frames = r.sum_over_ranks(n)
    print(json.dumps({"rank": r.rank, "digest": digest, "dt": dt, "frames": frames, "open_ms": e.stats().ms_open}), flush=True)
    e.close()
    r.close()
""") % ROOT


def test_two_processes_one_stream_each_equal_the_single_process_stream(tmp_path, E):
    from ceracoder_amd import synth
    w, h, n = 640, 368, 12
    e = E.Encoder(w, h, gop=6, fixed_qp=30)
    m = hashlib.sha256()
    for i, (y, uv) in enumerate(synth.s2_frames(w, h, n)):
        m.update(e.encode(y, uv, pts=i)[0])
    e.close()
    script = tmp_path / "w.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29547", str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    res = [json.loads(j) for j in re.findall(r"\{[^{}]*\}", out.stdout)]  # the two ranks share one pipe: their lines may arrive glued together
    assert sorted(x["rank"] for x in res) == [0, 1]
    assert all(x["digest"] == m.hexdigest() for x in res), res
    assert all(x["frames"] == 2 * n and x["dt"] > 0 for x in res)
    assert all(0 < x["open_ms"] < 500 for x in res), res


@pytest.mark.gpu
def test_two_encoders_in_one_process_equal_the_single_stream(E):
    """Several encoders in ONE process share its hardware queues, so the band deblocker must not wait there for an intra_p_kernel of
    its own picture that may be queued behind another encoder's waiting kernel: with more than one encoder open every stage runs in
    stream order.  Same bits either way."""
    from ceracoder_amd import synth
    w, h, n = 640, 368, 14
    frames = list(synth.s2_frames(w, h, n))
    single = E.Encoder(w, h, gop=7, fixed_qp=28)
    ref = [single.encode(y, uv, pts=i)[0] for i, (y, uv) in enumerate(frames)]
    single.close()
    a, b = E.Encoder(w, h, gop=7, fixed_qp=28, pipeline_depth=1), E.Encoder(w, h, gop=7, fixed_qp=28, pipeline_depth=1)
    out = {id(a): [], id(b): []}
    for i, (y, uv) in enumerate(frames):
        for e in (a, b):
            e.submit(y, uv, pts=i)
            if e.pending > 1:
                out[id(e)].append(bytes(e.collect()[0]))
    for e in (a, b):
        while e.pending:
            out[id(e)].append(bytes(e.collect()[0]))
        e.close()
    assert out[id(a)] == ref and out[id(b)] == ref


@pytest.mark.gpu
def test_single_stream_encoders_share_a_gpu_and_equal_the_single_stream(E):
    """cfg.single_stream: every stage of an encoder in order on ONE HIP stream (one hardware queue each) -- how many encoders share a GPU
    without the driver time-slicing a few dozen streams.  Four of them, interleaved, three pictures in flight each: same bits as one alone."""
    from ceracoder_amd import synth
    w, h, n = 640, 368, 12
    frames = list(synth.s2_frames(w, h, n))
    single = E.Encoder(w, h, gop=6, fixed_qp=28)
    ref = [single.encode(y, uv, pts=i)[0] for i, (y, uv) in enumerate(frames)]
    single.close()
    encs = [E.Encoder(w, h, gop=6, fixed_qp=28, pipeline_depth=2, single_stream=True) for _ in range(4)]
    out = {id(e): [] for e in encs}
    for i, (y, uv) in enumerate(frames):
        for e in encs:
            e.submit(y, uv, pts=i)
            if e.pending > 2:
                out[id(e)].append(bytes(e.collect()[0]))
    for e in encs:
        while e.pending:
            out[id(e)].append(bytes(e.collect()[0]))
        e.close()
    assert all(out[id(e)] == ref for e in encs)
