#!/usr/bin/env python3
"""Dispatch timeline of a window of a rocprofv3 kernel trace (rocpd database): name, queue, stream, grid, start, end, duration.
    python tools/rocpd_timeline.py gpurun_out/prof/x_results.db [index of the me_kernel dispatch to start at] [rows]"""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1]); c = db.cursor()
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
cols = [r[1] for r in c.execute("pragma table_info(%s)" % ks)]
namecol = "kernel_name" if "kernel_name" in cols else [x for x in cols if "name" in x][0]
names = {r[0]: re.sub(r"\(.*", "", r[1]) for r in c.execute("select id, %s from %s" % (namecol, ks))}
rows = c.execute("select kernel_id, start, end, queue_id, stream_id, grid_size_x from %s order by start" % kd).fetchall()
idx = [i for i, r in enumerate(rows) if "me_kernel" in names[r[0]]]
a = idx[int(sys.argv[2]) if len(sys.argv) > 2 else len(idx) // 2]
t0 = rows[a][1]
for k, s, e, q, sid, g in rows[a:a + (int(sys.argv[3]) if len(sys.argv) > 3 else 70)]:
    n = re.sub(r"^_Z\d+", "", names[k])[:16]
    print("%-16s q%-2d s%-2d g%-6d %8.1f %8.1f %7.1f" % (n, q, sid, g, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))
