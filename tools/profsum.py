#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd .db (kernel trace) as per-kernel stats: python tools_profsum.py x.db"""
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
cur = con.cursor()
rows = cur.execute("select name, grid_x, count(*), avg(duration)/1000.0, min(duration)/1000.0, max(duration)/1000.0, sum(duration)/1000.0, vgpr_count, lds_size from kernels group by name, grid_x order by sum(duration) desc").fetchall()
tot = sum(r[6] for r in rows)
print(f"{'kernel':72s} {'grid':>8s} {'calls':>6s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s} vgpr lds")
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    print(f"{r[0][:72]:72s} {r[1]:8d} {r[2]:6d} {r[3]:9.2f} {r[4]:9.2f} {r[5]:9.2f} {100*r[6]/tot:6.1f} {r[7]} {r[8]}")
