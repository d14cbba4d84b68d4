#!/usr/bin/env python3
"""Combine the rocprofv3 passes of a round into profiles/: kernel stats (rocpd db) and the separate --pmc FETCH_SIZE /
--pmc WRITE_SIZE passes (csv) -> profiles/rNN_kernel_stats_<workload>.csv, profiles/rNN_pmc_hbm_traffic_<workload>.json.
    python tools/pmc_summary.py ROUND WORKLOAD stats.db fetch_dir write_dir"""
import csv, glob, json, os, re, subprocess, sys
rnd, workload, db, fdir, wdir = sys.argv[1:6]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
stats_csv = os.path.join(root, "profiles", "r%02d_kernel_stats_%s.csv" % (int(rnd), workload))
subprocess.check_call([sys.executable, os.path.join(root, "tools", "rocpd_summary.py"), db, stats_csv], stdout=subprocess.DEVNULL)
trace = {}
for line in csv.DictReader(open(stats_csv)):
    trace[re.sub(r"^_Z\d+", "", line["Name"]).split("PK")[0].split("7db_args")[0].replace(".kd", "")] = line


def short(name):
    m = re.match(r"^_Z(\d+)", name)  # Itanium mangling: _Z<len><identifier>...
    if m:
        n = int(m.group(1))
        return name[m.end():m.end() + n]
    name = re.sub(r"^void\s+", "", name)          # demangled template instantiations: "void kernel<true>(args)"
    return re.sub(r"(<.*|\(.*|\.kd)$", "", name)


def load(d, counter):
    f = glob.glob(os.path.join(d, "*counter_collection.csv"))[0]
    out = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        out.setdefault(short(r["Kernel_Name"]), []).append(float(r["Counter_Value"]))
    return out


F, W = load(fdir, "FETCH_SIZE"), load(wdir, "WRITE_SIZE")
kern = {}
for k in sorted(set(F) | set(W)):
    f, w = F.get(k, []), W.get(k, [])
    e = {}
    if f:
        e["FETCH_SIZE"] = {"launches": len(f), "avg_kb": round(sum(f) / len(f), 3), "min_kb": min(f), "max_kb": max(f)}
    if w:
        e["WRITE_SIZE"] = {"launches": len(w), "avg_kb": round(sum(w) / len(w), 3), "min_kb": min(w), "max_kb": max(w)}
    if f and w:
        e["hbm_bytes_per_launch_corrected"] = int((2 * sum(f) / len(f) + sum(w) / len(w)) * 1024)
    calls = tot = 0
    for name, line in trace.items():  # template instantiations of one kernel are merged (weighted by calls)
        if short(line["Name"]) == k:
            if "pmb_kernelILb1E" in line["Name"]:  # ... except the one whose workgroups wait for another kernel's flags: its duration is not work
                e["gated_instantiation"] = {"calls": int(line["Calls"]), "avg_us": round(float(line["TotalDurationNs"]) / int(line["Calls"]) / 1e3, 3),
                                            "note": "pmb_kernel<GATED>: runs beside the reference picture's deblocking launch and waits per workgroup for the band it reads; the launch duration includes that wait"}
                continue
            calls += int(line["Calls"]); tot += float(line["TotalDurationNs"])
    if calls:
        e["kernel_trace_avg_us"] = round(tot / calls / 1e3, 3)
        e["kernel_trace_calls"] = calls
    kern[k] = e
doc = {"round": int(rnd), "workload": workload,
       "commands": {"kernel_stats": "cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats -d … -- python3 bench.py --workload W --no-cpu-baseline --no-gst-latency --no-extras  (rocpd database, summarised by tools/rocpd_summary.py)",
                    "pmc": "rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python3 bench.py --workload W --steps 120 --warmup 20 --no-cpu-baseline --no-gst-latency --no-extras ; the same with --pmc WRITE_SIZE (two separate passes, no other trace domains)"},
       "units": "counter values are KB; bytes = value*1024. On gfx950 FETCH_SIZE reports half of the bytes of a wide coalesced read (MI355X_MICROARCH.md, HBM), so hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024; Infinity-Cache hits are counted, so this is memory-side traffic, not strictly HBM",
       "kernels": kern}
out = os.path.join(root, "profiles", "r%02d_pmc_hbm_traffic_%s.json" % (int(rnd), workload))
json.dump(doc, open(out, "w"), indent=1)
print(open(stats_csv).read())
for k, e in kern.items():
    print(k, e.get("hbm_bytes_per_launch_corrected"), e.get("kernel_trace_avg_us"))
