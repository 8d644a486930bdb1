set -e
python -m pytest tests/test_gpu_backward.py -q -x 2>&1 | tail -3
python bench.py --steps 40 --warmup 10 --no_cpu_baseline --no_aux --profile_steps 30 2>&1 >/dev/null | grep "stage timing" | sed 's/.*stage timing done: //'
EXTRA_HIPCC_FLAGS="-DOMFS_SSIM_ROWS=56" bash omfs_4d_video_gen_amd/csrc/build.sh > /dev/null
echo "== rows 56"
python bench.py --steps 40 --warmup 10 --no_cpu_baseline --no_aux --profile_steps 30 2>&1 >/dev/null | grep "stage timing" | sed 's/.*stage timing done: //'
EXTRA_HIPCC_FLAGS="-DOMFS_SSIM_ROWS=23" bash omfs_4d_video_gen_amd/csrc/build.sh > /dev/null
echo "== rows 23"
python bench.py --steps 40 --warmup 10 --no_cpu_baseline --no_aux --profile_steps 30 2>&1 >/dev/null | grep "stage timing" | sed 's/.*stage timing done: //'
