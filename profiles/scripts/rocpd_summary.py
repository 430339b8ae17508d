#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (--kernel-trace --stats) as a
per-kernel table: calls, total / average / min / max duration."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
rows = db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows) or 1
print(f"{'kernel':70s} {'calls':>6s} {'total_ms':>10s} {'avg_ms':>10s} {'min_ms':>10s} {'max_ms':>10s} {'pct':>6s}")
for name, n, s, a, mn, mx in rows:
    print(f"{name[:70]:70s} {n:6d} {s/1e6:10.3f} {a/1e6:10.3f} {mn/1e6:10.3f} {mx/1e6:10.3f} {100*s/tot:6.2f}")
