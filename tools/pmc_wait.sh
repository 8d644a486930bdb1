#!/usr/bin/env bash
# Where do the waves of each kernel wait?  Five separate --pmc passes (wave-cycle accounting, active cycles per unit, LDS, fetch,
# totals) over the training loop of the bench scene; per-kernel averages land in gpurun_out/pmc_wait_<suffix>.json, the databases
# are deleted on the box (they are too big to travel).   usage (GPU box): bash tools/pmc_wait.sh <suffix>
S=${1:-x}
PY=$(python -c 'import sys; print(sys.executable)')
export TMPDIR=/tmp
R=$PWD
cd /tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" \
           "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU" \
           "SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -d /tmp/pmcw_$i -o pmc -- "$PY" $R/tools/run_scene.py --train --finetune --pretrain 20 --iters 16 > $R/gpurun_out/pmcw_$i.log 2>&1 || echo "pass $i failed"
done
"$PY" - <<PYEOF
import json, sys, glob, os
sys.path.insert(0, "$R/tools")
from pmc_summary import summarise
out = {}
for i in range(1, 6):
    f = glob.glob(f"/tmp/pmcw_{i}/**/pmc_results.db", recursive=True)
    if not f:
        print("no database for pass", i); continue
    for k, v in summarise(f[0]).items():
        if "omfs" in k:
            out.setdefault(k[:64], {}).update({c: round(x, 1) for c, x in v.items()})
json.dump(out, open("$R/gpurun_out/pmc_wait_$S.json", "w"), indent=1)
for k, v in out.items():
    if "composite" in k or "bin_" in k or "tile_sort" in k:
        print(k[9:40], {c: int(x) for c, x in v.items()})
PYEOF
