#!/usr/bin/env python3
"""Per-kernel duration statistics from a rocprofv3 --kernel-trace results database (rocpd sqlite) -> CSV on stdout."""
import collections
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
cur = con.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
t_sym = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol_"))
t_disp = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch_"))
syms = {r[0]: r[1] for r in cur.execute(f"select id, kernel_name from {t_sym}")}
d = collections.defaultdict(list)
for kid, s, e in cur.execute(f"select kernel_id, start, end from {t_disp}"):
    d[syms[kid].split("(")[0]].append(e - s)
total = sum(sum(v) for v in d.values())
print("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs")
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    print(f"\"{k}\",{len(v)},{sum(v)},{sum(v) / len(v):.1f},{100.0 * sum(v) / total:.2f},{min(v)},{max(v)}")
