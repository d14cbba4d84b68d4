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


# ---- the launcher behind `bench.py --gpus N` (one process per stream, as /root/reference/bindings/typescript/src/process.ts:129-170 starts them)
LAUNCHED = textwrap.dedent("""
    import json, os, sys
    sys.path.insert(0, %r)
    from ceracoder_amd import multistream as M
    cpus = M.apply_affinity()
    r = M.Ranks()
    total = r.sum_over_ranks(r.rank + 1)
    if os.environ.get("FAIL_RANK") == str(r.rank):
        sys.exit(7)
    r.barrier()
    print(json.dumps({"rank": r.rank, "local": r.local_rank, "world": r.world, "total": total, "cpus": cpus, "device": os.environ["MI355_BENCH_DEVICE"],
                      "shared": os.environ.get("MI355_BENCH_SHARED_DEVICE", "0")}), flush=True)
    r.close()
""") % ROOT


def test_plan_ranks_bookkeeping():
    sys.path.insert(0, ROOT)
    from ceracoder_amd import multistream as M
    # eight GPUs on two sockets, 8 ranks: device i, CPUs from the GPU's own node, disjoint between ranks
    nodes = [0, 0, 0, 0, 1, 1, 1, 1]
    cpus = {0: list(range(0, 32)), 1: list(range(32, 64))}
    plans = M.plan_ranks(8, 8, nodes, cpus, list(range(64)), master_port=1234)
    assert [p["device"] for p in plans] == list(range(8))
    assert all(p["env"]["RANK"] == str(i) and p["env"]["LOCAL_RANK"] == str(i) and p["env"]["WORLD_SIZE"] == "8" for i, p in enumerate(plans))
    assert all(p["env"]["MASTER_ADDR"] == "127.0.0.1" and p["env"]["MASTER_PORT"] == "1234" for p in plans)
    assert all(set(p["cpus"]) <= set(cpus[nodes[i]]) and len(p["cpus"]) == 8 for i, p in enumerate(plans))
    assert len(set(c for p in plans for c in p["cpus"])) == 64 and not any(p["shares_device"] for p in plans)
    # two ranks on a one-GPU box share device 0 (and say so); no NUMA information: the allowed CPUs are split evenly
    plans = M.plan_ranks(2, 1, [], None, [2, 3, 4, 5, 6, 7])
    assert [p["device"] for p in plans] == [0, 0] and all(p["shares_device"] and p["env"]["MI355_BENCH_SHARED_DEVICE"] == "1" for p in plans)
    assert [p["cpus"] for p in plans] == [[2, 3, 4], [5, 6, 7]]
    # a cpuset narrower than the GPU's node: only CPUs this process may use are handed out
    plans = M.plan_ranks(2, 2, [0, 1], {0: [0, 1, 2, 3], 1: [4, 5, 6, 7]}, [0, 1, 4])
    assert plans[0]["cpus"] == [0, 1] and plans[1]["cpus"] == [4]
    assert M.parse_cpulist("0-3,8,10-11") == [0, 1, 2, 3, 8, 10, 11]


def test_launcher_starts_n_ranks_and_collects_rank0(tmp_path):
    sys.path.insert(0, ROOT)
    from ceracoder_amd import multistream as M
    script = tmp_path / "child.py"
    script.write_text(LAUNCHED)
    os.environ.pop("FAIL_RANK", None)
    code, out0, errs, plans = M.launch(2, [str(script)], n_devices=1, timeout=300)
    assert code == 0, errs
    import json
    res = json.loads([l for l in out0.splitlines() if l.startswith("{")][-1])
    assert res["rank"] == 0 and res["world"] == 2 and res["total"] == 3.0 and res["device"] == "0" and res["shared"] == "1"
    assert res["cpus"] == plans[0]["cpus"] and set(plans[0]["cpus"]).isdisjoint(plans[1]["cpus"])


def test_launcher_reports_a_failing_rank(tmp_path):
    sys.path.insert(0, ROOT)
    from ceracoder_amd import multistream as M
    script = tmp_path / "child.py"
    script.write_text(LAUNCHED)
    os.environ["FAIL_RANK"] = "1"
    try:
        code, out0, errs, _ = M.launch(2, [str(script)], n_devices=2, timeout=300)
    finally:
        os.environ.pop("FAIL_RANK", None)
    assert code != 0


def test_bench_gpus_2_goes_through_the_launcher_and_fails_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("CPU-box test")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"], capture_output=True, text=True, timeout=600)
    assert out.returncode != 0
    assert "rank 0 stderr" in out.stderr and "needs a HIP device" in out.stderr
