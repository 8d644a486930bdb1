#!/usr/bin/env bash
# One gpurun call: default bench, kernel trace and the four PMC passes; results land in gpurun_out/*_<suffix>.
# usage (on the GPU box): bash tools/profile_all.sh <suffix>      then locally: python tools/make_profiles.py <tag> <suffix>
S=$1
# the interpreter itself goes after `--` (a PATH shim would be an exec hop under the profiler's preloaded library);
# a failing pass does not abort the others
PY=$(python -c 'import sys; print(sys.executable)')
export TMPDIR=/tmp
# the traced run leaves the aux workloads out (--no_aux): every launch of a kernel then belongs to the headline workload and
# the per-kernel averages of the trace are comparable with the stage times bench.py measures itself
R=$PWD
"$PY" bench.py > gpurun_out/bench_$S.json 2> gpurun_out/bench_$S.err
tail -1 gpurun_out/bench_$S.json | cut -c1-160
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$S -o prof -- "$PY" $R/bench.py --no_cpu_baseline --no_aux > $R/gpurun_out/bench_prof_$S.json 2> $R/gpurun_out/bench_prof_$S.err
echo trace done
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/pmc_fetch_$S -o pmc -- "$PY" $R/tools/run_scene.py --train --finetune --pretrain 20 --iters 16 > $R/gpurun_out/pmc_fetch_$S.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/pmc_write_$S -o pmc -- "$PY" $R/tools/run_scene.py --train --finetune --pretrain 20 --iters 16 > $R/gpurun_out/pmc_write_$S.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS --kernel-trace -d $R/gpurun_out/pmc_inst_$S -o pmc -- "$PY" $R/tools/run_scene.py --train --finetune --pretrain 20 --iters 16 > $R/gpurun_out/pmc_inst_$S.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $R/gpurun_out/pmc_busy_$S -o pmc -- "$PY" $R/tools/run_scene.py --train --finetune --pretrain 20 --iters 16 > $R/gpurun_out/pmc_busy_$S.log 2>&1
echo pmc done
