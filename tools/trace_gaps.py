"""Idle time between consecutive kernels of one HIP queue in a rocprofv3 --kernel-trace database: where the stream waits
(event waits, host-bound enqueue, tiny-kernel dispatch).  usage: python tools/trace_gaps.py <dir with *_results.db> [marker]
`marker` = substring of the kernel that closes one unit of work (default: the Gaussians' Adam launch; image_to_rgb8 for the render loop)."""
import glob
import sqlite3
import sys

import numpy as np

db = glob.glob(sys.argv[1].rstrip("/") + "/*results.db")[0]
marker = sys.argv[2] if len(sys.argv) > 2 else None
con = sqlite3.connect(db)
tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
kt = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
names = dict(con.execute(f"select id, kernel_name from {ks}").fetchall())
rows = con.execute(f"select kernel_id, queue_id, start, end from {kt} order by start").fetchall()
if marker is None:      # the Gaussians' Adam launch: adam_sh_rest_kernel on one GPU, adam_kernel in data-parallel runs
    marker = "adam_sh_rest_kernel" if any("adam_sh_rest_kernel" in v for v in names.values()) else "adam_kernel"
idx = [i for i, r in enumerate(rows) if marker in names[r[0]]]
gaps, tot, span, busy = {}, [], [], []
for a, b in zip(idx[:-1], idx[1:]):
    seg = rows[a + 1:b + 1]
    if len(seg) > 30 or not all(("omfs" in names[r[0]]) or ("rocclr" in names[r[0]]) for r in seg):
        continue                      # units interleaved with other work (set-up, torch kernels) are skipped
    q = rows[b][1]
    t_prev, g = rows[a][3], 0
    for r in (r for r in seg if r[1] == q):
        gap = max(0, r[2] - t_prev); g += gap
        gaps.setdefault(names[r[0]].split("omfs")[-1][:34], []).append(gap)
        t_prev = max(t_prev, r[3])
    tot.append(g); span.append(rows[b][3] - rows[a][3]); busy.append(sum(r[3] - r[2] for r in seg if r[1] == q))
print(f"{len(tot)} clean units closed by *{marker}*: {np.mean(span) / 1e3:.1f} us each, kernels {np.mean(busy) / 1e3:.1f} us, idle {np.mean(tot) / 1e3:.1f} us")
for k, v in sorted(gaps.items(), key=lambda kv: -np.mean(kv[1])):
    print(f"  idle before {k:36s} {np.mean(v) / 1e3:7.2f} us  (n={len(v)})")
