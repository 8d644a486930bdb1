#!/usr/bin/env python3
"""Counter bytes vs known bytes for tools/micro/fetch_calib.hip: python fetch_calib_report.py <dir FETCH_SIZE> <dir WRITE_SIZE> <log with KNOWN lines>"""
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pmc_summary import summarise  # noqa: E402


def db(d):
    return (glob.glob(d.rstrip("/") + "/**/*results.db", recursive=True) or [None])[0]


f, w = summarise(db(sys.argv[1]), 3), summarise(db(sys.argv[2]), 3)
known = {}
for line in open(sys.argv[3]):
    if line.startswith("KNOWN"):
        _, k, _, r, _, wr = line.split()
        known[k] = (int(r), int(wr))
out = {"_note": "rocprofv3 FETCH_SIZE / WRITE_SIZE (KiB) x 1024 against the bytes the kernels of tools/micro/fetch_calib.hip really ask for "
                "(2 GiB tables, uniformly random records, every record touched once: HBM traffic); ratio = known / counter"}
for k, (kr, kw) in known.items():
    fk = next((v for n, v in f.items() if k + "(" in n or n.startswith(k) or ("_" + k) in n or k in n.split("(")[0]), None)
    wk = next((v for n, v in w.items() if k in n.split("(")[0]), None)
    fb = fk["FETCH_SIZE"] * 1024 if fk else None
    wb = wk["WRITE_SIZE"] * 1024 if wk else None
    out[k] = {"known_read_bytes": kr, "FETCH_SIZE_bytes": fb, "read_ratio": round(kr / fb, 3) if fb else None,
              "known_write_bytes": kw, "WRITE_SIZE_bytes": wb, "write_ratio": round(kw / wb, 3) if wb and kw else None}
    print(k, out[k])
if len(sys.argv) > 4:
    json.dump(out, open(sys.argv[4], "w"), indent=1)
