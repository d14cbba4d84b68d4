#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (kernel trace): per-kernel stats and, for one steady-state
P picture, the dispatch timeline with the idle gaps between kernels.
    python tools/rocpd_summary.py gpurun_out/prof/x_results.db [out.csv]"""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1]); c = db.cursor()
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
cols = [r[1] for r in c.execute("pragma table_info(%s)" % ks)]
namecol = "kernel_name" if "kernel_name" in cols else [x for x in cols if "name" in x][0]
names = {r[0]: re.sub(r"\(.*", "", r[1]) for r in c.execute("select id, %s from %s" % (namecol, ks))}
rows = c.execute("select kernel_id, start, end, queue_id, stream_id from %s order by start" % kd).fetchall()
st = {}
for k, s, e, _q, _s in rows:
    st.setdefault(names.get(k, str(k)), []).append(e - s)
tot = sum(sum(v) for v in st.values())
lines = ["Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage"]
for n, v in sorted(st.items(), key=lambda kv: -sum(kv[1])):
    lines.append("%s,%d,%d,%.1f,%d,%d,%.2f" % (n, len(v), sum(v), sum(v) / len(v), min(v), max(v), 100.0 * sum(v) / tot))
print("\n".join(lines))
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write("\n".join(lines) + "\n")
# timeline of the last full picture that starts with me_kernel
idx = [i for i, (k, s, e, _q, _s) in enumerate(rows) if "me_kernel" in names.get(k, "")]
if len(idx) > 3:
    a, b = idx[-7], idx[-3]
    t0 = rows[a][1]
    print("\n# a few P pictures (us from the first me_kernel start): name queue stream start end dur")
    for k, s, e, q, sid in rows[a:b + 1]:
        print("%-28s q%-3d s%-3d %9.1f %9.1f %8.1f" % (names.get(k, str(k))[:28], q, sid, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))
