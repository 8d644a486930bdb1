set -e
python -m pytest tests -m gpu -q -x 2>&1 | tail -2
python bench.py --steps 40 --warmup 10 --no_cpu_baseline --no_aux --profile_steps 30 2>&1 >/dev/null | grep "stage timing" | sed 's/.*stage timing done: //'
