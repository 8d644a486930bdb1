#!/usr/bin/env bash
# One gpurun call that produces everything under profiles/ for a round: default bench line, kernel trace + stats of the same command
# (aux workloads left out), the four PMC passes of profile_all.sh, the wave-wait passes, the iteration timeline of the eager
# path.  Only summaries travel back (gpurun_out/profiles_<suffix>/): the databases are deleted on the box.
# usage (GPU box):  bash tools/profile_round.sh <tag> <suffix>        e.g.  bash tools/profile_round.sh r02_b b
TAG=$1; S=$2
R=$PWD
PY=$(python -c 'import sys; print(sys.executable)')
OUT=$R/gpurun_out/profiles_$S
mkdir -p $OUT
bash tools/profile_all.sh $S
"$PY" tools/make_profiles.py $TAG $S $OUT > $OUT/${TAG}_make_profiles.log 2>&1 || echo "make_profiles failed"
"$PY" tools/iter_timeline.py $R/gpurun_out/prof_$S 60 > $OUT/${TAG}_iteration_timeline.txt 2>&1 || echo "iter_timeline failed"
rm -rf $R/gpurun_out/prof_$S $R/gpurun_out/pmc_fetch_$S $R/gpurun_out/pmc_write_$S $R/gpurun_out/pmc_inst_$S $R/gpurun_out/pmc_busy_$S
bash tools/pmc_wait.sh $S > $OUT/${TAG}_pmc_wait.log 2>&1
cp $R/gpurun_out/pmc_wait_$S.json $OUT/${TAG}_pmc_wait.json 2>/dev/null
rm -rf /tmp/pmcw_*
ls -la $OUT
