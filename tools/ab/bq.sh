# usage: bash tools/ab/bq.sh <lib tags...>  -- quick bench line per prebuilt library (no aux, no cpu baseline)
for v in "$@"; do
  cp tools/_ab/so/$v.so omfs_4d_video_gen_amd/libomfs_splat.so
  python bench.py --no_aux --no_cpu_baseline --profile_steps 60 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stages_ms']; print('$v', d['value'], d['ms_per_step_median'], 'fwd', s['composite_fwd'], 'bwd', s['composite_bwd'], 'loss', s['loss'])"
done
