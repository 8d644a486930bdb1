#!/usr/bin/env bash
# Build-flag sweeps on the GPU box: rebuilds libomfs_splat.so with each set of -D flags and prints the per-stage times
# of the bench workload (HIP events between the C-ABI calls).  The tuning macros it is meant for:
#   OMFS_FWD_SEQ_SEGS, OMFS_DEEP_WAVES (composite forward hand-over), OMFS_BWD_PEND, OMFS_BWD_WAVES (backward residency),
#   OMFS_PBWD_WAVES (project_bwd residency), OMFS_BIN_THREADS, OMFS_SORT_SMALL_NT, OMFS_BUCKET_MAX (binning / sort).
# usage: bash tools/sweep_flags.sh "" "-DOMFS_BWD_PEND=8" "-DOMFS_BWD_PEND=8 -DOMFS_BWD_WAVES=7"
cd "$(dirname "$0")/.."
for v in "$@"; do
  export EXTRA_HIPCC_FLAGS="$v"
  bash omfs_4d_video_gen_amd/csrc/build.sh > /dev/null 2>&1 || { echo "== $v: build failed"; continue; }
  echo "== ${v:-<defaults>}"
  python bench.py --steps 60 --warmup 20 --no_cpu_baseline --no_aux --profile_steps 30 2>&1 >/dev/null | grep "stage timing" | sed 's/.*stage timing done: //'
done
export EXTRA_HIPCC_FLAGS=""
bash omfs_4d_video_gen_amd/csrc/build.sh > /dev/null
