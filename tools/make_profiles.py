#!/usr/bin/env python3
"""Turn the merged gpurun_out/ results of a profiling call into the tracked files under profiles/:
  make_profiles.py <tag> <suffix>      e.g.  make_profiles.py r01_h h
make_profiles.py <tag> <suffix> <out_dir> on the GPU box itself (then only the summaries are merged back).
reads gpurun_out/{bench_<s>.json, bench_prof_<s>.json, prof_<s>/, pmc_fetch_<s>/, pmc_write_<s>/, pmc_inst_<s>/, pmc_busy_<s>/}."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from pmc_summary import summarise  # noqa: E402

tag, suf = sys.argv[1], sys.argv[2]
go = os.path.join(ROOT, "gpurun_out")
pr = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "profiles")     # on the GPU box: a directory under gpurun_out/ (the
os.makedirs(pr, exist_ok=True)                                               # databases themselves are too big to travel back)
with open(os.path.join(pr, f"{tag}_kernel_stats.csv"), "w") as f:
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_stats.py"), os.path.join(go, f"prof_{suf}", "prof_results.db")], stdout=f, check=True)
for src, dst in ((f"bench_{suf}.json", f"{tag}_bench_default.json"), (f"bench_prof_{suf}.json", f"{tag}_bench_under_rocprof.json")):
    line = open(os.path.join(go, src)).read().strip().splitlines()[-1]
    open(os.path.join(pr, dst), "w").write(line + "\n")
os.environ["OMFS_PROFILE_TAG"] = tag
subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_traffic.py"), os.path.join(go, f"pmc_fetch_{suf}", "pmc_results.db"),
                os.path.join(go, f"pmc_write_{suf}", "pmc_results.db"), "1920x1080x300000", os.path.join(pr, "traffic.json")], check=True,
               stdout=subprocess.DEVNULL)
a = summarise(os.path.join(go, f"pmc_inst_{suf}", "pmc_results.db"))
b = summarise(os.path.join(go, f"pmc_busy_{suf}", "pmc_results.db"))
out = {}
for k in a:
    if "omfs" not in k:
        continue
    d = dict(a[k])
    d.update(b.get(k, {}))
    if d.get("GRBM_GUI_ACTIVE"):      # summed over the 8 XCDs
        cyc = d["GRBM_GUI_ACTIVE"] / 8.0
        d["valu_issue_util"] = round(d.get("SQ_ACTIVE_INST_VALU", 0) * 4 / (1024 * cyc), 3)
    out[k[:70]] = {kk: (round(v, 3) if isinstance(v, float) else v) for kk, v in d.items()}
json.dump(out, open(os.path.join(pr, f"{tag}_pmc_instruction_mix.json"), "w"), indent=1)
for k, v in out.items():
    print(k[9:44].ljust(36), v.get("valu_issue_util"), int(v.get("SQ_INSTS_VALU", 0)))
