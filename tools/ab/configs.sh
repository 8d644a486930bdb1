# The other BASELINE configurations on the current build (FLAME fine-tuning on / fixed sequence): one line each
run() { python bench.py --no_aux --no_cpu_baseline --profile_steps 4 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print(c['n_gaussians'], str(c['width'])+'x'+str(c['height']), 'views', c['views'], 'D', c['tile_pairs_D'], d['value'], 'it/s', d['ms_per_step'], 'ms', '$*')"; }
for ff in "" "--frozen_flame"; do
  run --n_gaussians 500000 $ff
  run --n_gaussians 100000 --width 512 --height 512 --views 1 $ff
  run --n_gaussians 5000 --width 256 --height 256 --views 1 $ff
  run --n_gaussians 300000 --width 3840 --height 2160 $ff
done
