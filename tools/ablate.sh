#!/usr/bin/env bash
# Timing-only ablation builds of the composite kernels (results are wrong by construction; only the stage times matter).
set -e
cd "$(dirname "$0")/.."
for v in NONE OMFS_ABL_NOATOMIC OMFS_ABL_NOREDUCE OMFS_ABL_NOMATH OMFS_ABL_NOMASK; do
  if [ "$v" = NONE ]; then export EXTRA_HIPCC_FLAGS=""; else export EXTRA_HIPCC_FLAGS="-D$v"; fi
  bash omfs_4d_video_gen_amd/csrc/build.sh > /dev/null
  echo "== $v"
  python bench.py --steps 40 --warmup 10 --no_cpu_baseline --no_aux --profile_steps 30 2>&1 >/dev/null | grep "stage timing" | sed 's/.*stage timing done: //'
done
export EXTRA_HIPCC_FLAGS=""
bash omfs_4d_video_gen_amd/csrc/build.sh > /dev/null
