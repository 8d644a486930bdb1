#!/usr/bin/env bash
# kernel trace of a short headline run -> per-kernel averages (composite / binning kernels first).  usage (GPU box): bash tools/quick_trace.sh [steps]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/qt && rocprofv3 --kernel-trace -d /tmp/qt -o qt -- python3 $R/bench.py --steps ${1:-100} --warmup 120 --no_cpu_baseline --no_aux --profile_steps 0 > /tmp/qt.log 2>&1
db=$(find /tmp/qt -name "*.db" | head -1)
python3 $R/tools/kernel_stats.py $db | awk -F, 'NR==1 || $2 >= 50' | cut -c1-150 | head -${2:-30}
