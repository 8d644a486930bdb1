"""One training iteration on the GPU's clock, from a rocprofv3 --kernel-trace database: for every kernel of the iteration its
queue, start (relative to the end of the previous iteration's Gaussian Adam kernel) and duration, averaged over the clean
iterations -- shows what runs beside what when the FLAME chain is on the side stream.
usage: python tools/iter_timeline.py <dir with *_results.db> [n_last_iterations]"""
import glob
import sqlite3
import sys

import numpy as np

db = (glob.glob(sys.argv[1].rstrip("/") + "/*results.db") + glob.glob(sys.argv[1].rstrip("/") + "/*/*results.db"))[0]
n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 40
con = sqlite3.connect(db)
tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
kt = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
import re


def short(v):
    m = re.match(r"_ZN4omfs(\d+)", v)          # mangled: _ZN4omfs<len><name>...
    if m:
        n = int(m.group(1))
        return v[m.end():m.end() + n] + ("<" + re.search(r"ILi(\d+)E", v).group(1) + ">" if "ILi" in v else "")
    return v.split("(")[0].replace("omfs::", "").replace("void ", "")


names = {k: short(v) for k, v in con.execute(f"select id, kernel_name from {ks}")}
rows = con.execute(f"select kernel_id, queue_id, start, end from {kt} order by start").fetchall()
marks = [i for i, r in enumerate(rows) if names[r[0]].startswith(("adam_kernel", "adam_sh_rest_kernel"))]   # the Gaussians' Adam launch closes an iteration
units = []
for a, b in zip(marks[:-1], marks[1:]):
    seg = rows[a + 1:b + 1]
    if len(seg) <= 40:
        units.append((rows[a][3], seg))
units = units[-n_last:]
sig = {}
for t0, seg in units:
    key = tuple((names[r[0]], r[1]) for r in seg)
    sig.setdefault(key, []).append([(r[2] - t0, r[3] - r[2]) for r in seg])
key, runs = max(sig.items(), key=lambda kv: len(kv[1]))
arr = np.array(runs, np.float64) / 1e3
queues = sorted({q for _, q in key})
print(f"{len(runs)} iterations with the most common kernel sequence ({len(key)} kernels, queues {queues}); "
      f"iteration span {np.mean([u[1][-1][3] - u[0] for u in units]) / 1e3:.1f} us")
for i, (name, q) in enumerate(key):
    print(f"  q{queues.index(q)}  start {arr[:, i, 0].mean():8.1f} us  dur {arr[:, i, 1].mean():7.1f} us  {name[:60]}")
