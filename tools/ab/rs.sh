# usage: bash tools/ab/rs.sh <n_streams...>  -- render_surgery frames/s of the bench scene per stream count
for n in "$@"; do
  python bench.py --steps 5 --warmup 2 --profile_steps 2 --no_cpu_baseline --render_streams $n 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); a=d['aux']; print('streams $n', a['render_surgery_fps'], a.get('render_surgery_fps_with_png'))"
done
