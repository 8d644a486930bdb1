# usage: bash tools/ab/bq.sh <lib tags...>  -- quick bench line per prebuilt library (no aux, no cpu baseline)
# the candidate is bound through OMFS_LIB_PATH: the in-tree library is never overwritten
for v in "$@"; do
  OMFS_LIB_PATH=$PWD/tools/_ab/so/$v.so python bench.py --no_aux --no_cpu_baseline --profile_steps 60 2>gpurun_out/bq_$v.err | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stages_ms']; print('$v', d['library']['sha256_16'], d['value'], d['ms_per_step_median'], 'fwd', s['composite_fwd'], 'bwd', s['composite_bwd'], 'loss', s['loss'], 'bin', round(s['bin_count']+s['bin_scan']+s['bin_scatter']+s['tile_sort'],4))"
done
