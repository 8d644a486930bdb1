# usage: bash tools/ab/bq2.sh <lib tags...>  -- as bq.sh, other stages (candidate bound through OMFS_LIB_PATH)
for v in "$@"; do
  OMFS_LIB_PATH=$PWD/tools/_ab/so/$v.so python bench.py --no_aux --no_cpu_baseline --profile_steps 60 2>gpurun_out/bq_$v.err | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stages_ms']; print('$v', d['library']['sha256_16'], d['value'], d['ms_per_step_median'], 'pbwd', s['project_bwd'], 'flame_bwd', s['flame_bwd'], 'adam', s['adam'])"
done
