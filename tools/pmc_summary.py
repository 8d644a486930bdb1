#!/usr/bin/env python3
"""Per-kernel averages of the counters in a rocprofv3 --pmc results database (rocpd sqlite)."""
import collections
import json
import sqlite3
import sys


def summarise(path, last=16):
    con = sqlite3.connect(path)
    cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    def tab(prefix):
        return next(t for t in tabs if t.startswith(prefix))
    t_pmc, t_sym, t_ev, t_disp = tab("rocpd_info_pmc_"), tab("rocpd_info_kernel_symbol_"), tab("rocpd_pmc_event_"), tab("rocpd_kernel_dispatch_")
    names = {r[0]: r[1] for r in cur.execute(f"select id, name from {t_pmc}")}
    syms = {r[0]: r[1] for r in cur.execute(f"select id, kernel_name from {t_sym}")}
    disp = {r[0]: (syms[r[1]], r[2], r[3]) for r in cur.execute(f"select id, kernel_id, start, end from {t_disp}")}
    cols = [r[1] for r in cur.execute(f"pragma table_info({t_ev})")]
    per = collections.defaultdict(lambda: collections.defaultdict(float))   # dispatch -> counter -> value (summed over instances)
    for ev_id, pmc_id, value in cur.execute(f"select event_id, pmc_id, value from {t_ev}"):
        per[ev_id][names[pmc_id]] += value
    # event_id -> dispatch id: rocpd_event / dispatch share ids through event table; fall back to order
    t_event_link = None
    try:
        link = {r[0]: r[1] for r in cur.execute(f"select event_id, id from {t_disp}")}
    except sqlite3.OperationalError:
        link = None
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    for ev_id, ctrs in per.items():
        d = link.get(ev_id) if link else ev_id
        if d is None or d not in disp:
            continue
        k = disp[d][0].split("(")[0]
        for c, v in ctrs.items():
            out[k][c].append(v)
    res = {}
    for k, ctrs in out.items():
        res[k] = {c: sum(v[-last:]) / len(v[-last:]) for c, v in ctrs.items()}
        res[k]["launches"] = len(next(iter(ctrs.values())))
    return res


if __name__ == "__main__":
    r = summarise(sys.argv[1])
    for k, v in sorted(r.items()):
        if "omfs" in k:
            print(k[:60].ljust(60), {c: (round(x) if x > 100 else round(x, 2)) for c, x in v.items()})
    if len(sys.argv) > 2:
        json.dump(r, open(sys.argv[2], "w"), indent=1)
