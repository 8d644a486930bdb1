#!/usr/bin/env bash
# Counters of the two composite_bwd implementations on the bench scene (tools/bwd_time.py launches both on one fixed state):
# instruction mix, active / wait cycles, LDS, MFMA co-execution.  Separate --pmc passes with --kernel-trace only.
# usage (GPU box): bash tools/pmc_bwd.sh <out.json>
OUT=${1:-gpurun_out/pmc_bwd.json}
PY=$(python -c 'import sys; print(sys.executable)')
export TMPDIR=/tmp
R=$PWD
cd /tmp
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" \
           "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH"; do
  i=$((i+1))
  rm -rf /tmp/pmcb_$i
  rocprofv3 --pmc $set --kernel-trace -d /tmp/pmcb_$i -o pmc -- "$PY" $R/tools/bwd_time.py --pretrain 100 --reps 16 > $R/gpurun_out/pmcb_$i.log 2>&1 || echo "pass $i failed: $(tail -2 $R/gpurun_out/pmcb_$i.log)"
done
"$PY" - <<PYEOF
import json, sys, glob
sys.path.insert(0, "$R/tools")
from pmc_summary import summarise
out = {}
for i in range(1, 6):
    f = glob.glob(f"/tmp/pmcb_{i}/**/pmc_results.db", recursive=True)
    if not f:
        print("no database for pass", i); continue
    for k, v in summarise(f[0]).items():
        if "composite_bwd" in k:
            out.setdefault(k[9:45], {}).update({c: round(x, 1) for c, x in v.items()})
json.dump(out, open("$R/$OUT", "w"), indent=1)
for k, v in out.items():
    print(k, {c: int(x) for c, x in v.items()})
PYEOF
rm -rf /tmp/pmcb_*
