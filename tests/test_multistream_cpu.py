"""N>1 path on CPU: two gloo ranks run the same barrier / max-over-ranks bracket bench.py uses
(no data-path collective exists: streams are independent)."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import json, sys, time
    sys.path.insert(0, %r)
    from ceracoder_amd.multistream import Ranks
    r = Ranks()
    assert r.world == 2
    # rank 1 is the slow stream: the whole-job time is the MAX over ranks
    dt, frames = r.timed(lambda: (time.sleep(0.05 + 0.25 * r.rank), 10 + r.rank)[1])
    total = r.sum_over_ranks(frames)
    if r.rank == 0:
        print(json.dumps({"dt": dt, "total_frames": total}))
    r.close()
""") % ROOT


def test_two_rank_bracket_takes_max_and_sums_work(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29533", str(script)], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert res["total_frames"] == 21
    assert 0.29 <= res["dt"] < 2.0, res


def test_single_rank_is_a_noop_bracket():
    sys.path.insert(0, ROOT)
    from ceracoder_amd.multistream import Ranks
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        os.environ.pop(k, None)
    r = Ranks()
    dt, v = r.timed(lambda: 7)
    assert v == 7 and dt >= 0 and r.sum_over_ranks(3) == 3.0
