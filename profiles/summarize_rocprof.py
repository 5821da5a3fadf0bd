#!/usr/bin/env python3
"""Turn a rocprofv3 rocpd database (…_results.db) into the text summary committed under profiles/.

usage: python profiles/summarize_rocprof.py gpurun_out/prof_x/x_results.db [kernel-substring] > profiles/rNN_x.txt
Prints the `--stats`-style table (calls, total/avg/min/max duration) and, for kernels matching the
substring, every dispatch with its launch geometry and register/LDS usage.
"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
sub = sys.argv[2] if len(sys.argv) > 2 else "hnsw"
cur = db.cursor()
print("# kernel stats (durations in microseconds)")
print(f"{'calls':>6} {'total_us':>12} {'avg_us':>11} {'min_us':>11} {'max_us':>11} {'pct':>6}  name")
rows = list(cur.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                        "from kernels group by name order by sum(duration) desc"))
tot = sum(r[2] for r in rows) or 1
for name, n, s, a, mn, mx in rows[:25]:
    print(f"{n:6d} {s / 1e3:12.2f} {a / 1e3:11.2f} {mn / 1e3:11.2f} {mx / 1e3:11.2f} {100 * s / tot:6.2f}  {name[:110]}")
print()
print(f"# dispatches of kernels matching '{sub}'")
print(f"{'dur_us':>10} {'grid':>8} {'wg':>4} {'lds':>6} {'vgpr':>5} {'agpr':>5} {'sgpr':>5}  name")
for r in cur.execute("select duration, grid_x, workgroup_x, lds_size, vgpr_count, accum_vgpr_count, sgpr_count, name "
                     "from kernels where name like ? order by start", (f"%{sub}%",)):
    print(f"{r[0] / 1e3:10.2f} {r[1]:8d} {r[2]:4d} {r[3]:6d} {r[4]:5d} {r[5]:5d} {r[6]:5d}  {r[7][:90]}")
