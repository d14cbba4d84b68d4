"""N independent streams, one per GPU/rank (the path does not shard, so there is
no data-path collective).  The only cross-rank traffic is the barrier pair that brackets a timed
region and the MAX of the per-rank wall times -- host tensors over gloo."""
import os
import time


class Ranks:
    def __init__(self):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.dist = None
        if self.world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            if not dist.is_initialized():
                dist.init_process_group("gloo", rank=self.rank, world_size=self.world)
            self.dist = dist

    def barrier(self):
        if self.dist:
            self.dist.barrier()

    def max_over_ranks(self, value):
        if not self.dist:
            return float(value)
        import torch
        t = torch.tensor([float(value)], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t[0])

    def sum_over_ranks(self, value):
        if not self.dist:
            return float(value)
        import torch
        t = torch.tensor([float(value)], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t[0])

    def timed(self, fn, sync=None):
        """barrier + device sync, run fn(), device sync + barrier; returns (max wall seconds over ranks, fn's result)."""
        self.barrier()
        if sync:
            sync()
        t0 = time.perf_counter()
        out = fn()
        if sync:
            sync()
        self.barrier()
        return self.max_over_ranks(time.perf_counter() - t0), out

    def close(self):
        if self.dist:
            self.dist.barrier()
            self.dist.destroy_process_group()
            self.dist = None


# ---------------------------------------------------------------------------------------------------------------------
# Launcher: one PROCESS per stream, as the reference runs them (one ceracoder process per stream, spawned by
# /root/reference/bindings/typescript/src/process.ts:129-170).  `bench.py --gpus N` without RANK in the environment calls
# launch(): N fresh children, child i with RANK = LOCAL_RANK = i, WORLD_SIZE = N, encoder on device i, pinned to CPUs of the
# NUMA node its GPU hangs off.  Nothing here imports torch or touches HIP: the children are started before any
# process has initialised a GPU, and each child sets its own affinity before ITS first GPU call (apply_affinity()).

def _read(path, default=None):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return default


def parse_cpulist(text):
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11]"""
    out = []
    for part in (text or "").split(","):
        part = part.strip()
        if not part:
            continue
        if "-" in part:
            a, b = part.split("-")
            out.extend(range(int(a), int(b) + 1))
        else:
            out.append(int(part))
    return out


def gpu_numa_nodes(kfd_root="/sys/class/kfd/kfd/topology/nodes", pci_root="/sys/bus/pci/devices"):
    """NUMA node of every GPU in KFD topology order (the order HIP enumerates devices in when HIP_VISIBLE_DEVICES is unset),
    from sysfs alone: the KFD node's PCI address -> /sys/bus/pci/devices/<bdf>/numa_node.  -1 where the platform does not say."""
    nodes = []
    try:
        names = sorted(os.listdir(kfd_root), key=lambda s: int(s) if s.isdigit() else 1 << 30)
    except OSError:
        return nodes
    for n in names:
        props = {}
        for line in (_read(os.path.join(kfd_root, n, "properties"), "") or "").splitlines():
            kv = line.split()
            if len(kv) == 2:
                props[kv[0]] = kv[1]
        if int(props.get("simd_count", "0") or 0) <= 0:  # a CPU node
            continue
        loc, dom = int(props.get("location_id", "0") or 0), int(props.get("domain", "0") or 0)
        bdf = "%04x:%02x:%02x.%x" % (dom, (loc >> 8) & 0xFF, (loc >> 3) & 0x1F, loc & 7)
        nodes.append(int(_read(os.path.join(pci_root, bdf, "numa_node"), "-1") or -1))
    return nodes


def plan_ranks(n, n_devices, gpu_nodes=None, node_cpus=None, all_cpus=None, master_port=29511):
    """The bookkeeping of launch(), as a pure function: for rank i its environment additions, its device and its CPU set.
    n_devices: GPUs the box has (ranks beyond it share devices: rank i -> device i % n_devices);
    gpu_nodes: NUMA node per device (or None / -1: unknown); node_cpus: {node: [cpu, ...]}; all_cpus: CPUs this process may use.
    Ranks whose GPUs sit on the same NUMA node split that node's CPUs evenly (contiguous slices), so that their entropy-coding
    threads do not sit on each other; with no NUMA information the allowed CPUs are split evenly among all ranks."""
    if n < 1:
        raise ValueError("need at least one rank")
    n_devices = max(1, int(n_devices))
    all_cpus = sorted(all_cpus) if all_cpus else []
    plans = []
    by_node = {}
    for i in range(n):
        dev = i % n_devices
        node = gpu_nodes[dev] if gpu_nodes and dev < len(gpu_nodes) else -1
        if node_cpus is None or node not in node_cpus or not set(node_cpus[node]) & set(all_cpus or node_cpus[node]):
            node = -1
        by_node.setdefault(node, []).append(i)
        plans.append({"rank": i, "device": dev, "numa_node": node, "shares_device": n > n_devices, "cpus": [],
                      "env": {"RANK": str(i), "LOCAL_RANK": str(i), "WORLD_SIZE": str(n), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(master_port),
                              "LOCAL_WORLD_SIZE": str(n), "HSA_ENABLE_IPC_MODE_LEGACY": "0"}})
    for node, members in by_node.items():
        pool = all_cpus if node < 0 else sorted(set(node_cpus[node]) & set(all_cpus)) if all_cpus else sorted(node_cpus[node])
        if not pool:
            continue
        k = len(members)
        for j, i in enumerate(members):
            lo, hi = j * len(pool) // k, (j + 1) * len(pool) // k
            plans[i]["cpus"] = pool[lo:hi] if hi > lo else [pool[j % len(pool)]]
    for p in plans:
        if p["cpus"]:
            p["env"]["MI355_BENCH_CPUS"] = ",".join(str(c) for c in p["cpus"])
        p["env"]["MI355_BENCH_DEVICE"] = str(p["device"])
        p["env"]["MI355_BENCH_NUMA_NODE"] = str(p["numa_node"])
        if p["shares_device"]:
            p["env"]["MI355_BENCH_SHARED_DEVICE"] = "1"
    return plans


def apply_affinity():
    """Child side: pin this process (and every thread it starts later: the entropy-coding pool, HIP's helper threads) to the CPUs
    the launcher planned.  Call before the first GPU call.  Returns the CPU list in force."""
    want = parse_cpulist(os.environ.get("MI355_BENCH_CPUS", ""))
    if want and hasattr(os, "sched_setaffinity"):
        try:
            os.sched_setaffinity(0, want)
        except OSError:
            pass
    return sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else []


def _node_cpus():
    node_cpus = {}
    try:
        for d in os.listdir("/sys/devices/system/node"):
            if d.startswith("node") and d[4:].isdigit():
                node_cpus[int(d[4:])] = parse_cpulist(_read("/sys/devices/system/node/%s/cpulist" % d, ""))
    except OSError:
        pass
    return node_cpus or None


def self_plan():
    """A rank started by someone else's launcher (torch.distributed.run sets RANK / LOCAL_RANK / LOCAL_WORLD_SIZE but knows nothing
    about GPUs): work out the same plan launch() would have made and adopt this rank's part of it.  Returns the plan entry."""
    if "MI355_BENCH_CPUS" in os.environ or "MI355_BENCH_DEVICE" in os.environ:
        return None
    local_n = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if local_n <= 1:
        return None
    allowed = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    plans = plan_ranks(local_n, count_gpus() or 1, gpu_numa_nodes(), _node_cpus(), allowed)
    me = plans[local_rank % local_n]
    for k in ("MI355_BENCH_CPUS", "MI355_BENCH_DEVICE", "MI355_BENCH_NUMA_NODE", "MI355_BENCH_SHARED_DEVICE"):
        if k in me["env"]:
            os.environ[k] = me["env"][k]
    return me


def count_gpus():
    """GPUs of this box without initialising one: KFD topology nodes that have SIMDs (honours HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES counts)."""
    n = len(gpu_numa_nodes())
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None and v.strip() != "":
            n = min(n, len([x for x in v.split(",") if x.strip() != ""])) if n else len([x for x in v.split(",") if x.strip() != ""])
    return n


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch(n, argv, n_devices=None, python=None, timeout=None):
    """Start n children `python argv...`, child i with plan_ranks()'s environment; wait for all of them.
    Returns (exit code, rank 0's stdout, {rank: stderr tail}, the plans).  The exit code is 0 only if every child exited 0; when one fails
    the others are terminated (their own process groups, never by pattern)."""
    import subprocess
    import sys
    devs = n_devices if n_devices is not None else count_gpus()
    gnodes = gpu_numa_nodes()
    node_cpus = _node_cpus()
    allowed = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    plans = plan_ranks(n, devs or 1, gnodes, node_cpus, allowed, master_port=free_port())
    procs = []
    for p in plans:
        env = dict(os.environ)
        env.update(p["env"])
        procs.append(subprocess.Popen([python or sys.executable] + list(argv), env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                                      start_new_session=True))
    import threading
    outs = [None] * n

    def reap(i):
        try:
            outs[i] = procs[i].communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            procs[i].kill()
            outs[i] = procs[i].communicate()
    th = [threading.Thread(target=reap, args=(i,)) for i in range(n)]
    for t in th:
        t.start()
    failed = None
    while any(t.is_alive() for t in th):
        for i, pr in enumerate(procs):
            rc = pr.poll()
            if rc not in (None, 0) and failed is None:
                failed = i
                for j, other in enumerate(procs):  # a rank died: the others would sit in the barrier for ever
                    if j != i and other.poll() is None:
                        try:
                            os.killpg(other.pid, 15)
                        except OSError:
                            pass
        time.sleep(0.05)
    for t in th:
        t.join()
    code = 0
    for pr in procs:
        if pr.returncode != 0:
            code = pr.returncode if pr.returncode and pr.returncode > 0 else 1
    return code, outs[0][0] if outs[0] else "", {i: (o[1] or "")[-2000:] for i, o in enumerate(outs) if o}, plans
