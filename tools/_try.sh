set -e
for v in 1024 256; do
EXTRA_HIPCC_FLAGS="-DOMFS_BIN_THREADS=$v" bash omfs_4d_video_gen_amd/csrc/build.sh > /dev/null
echo "== threads $v"
python bench.py --steps 40 --warmup 10 --no_cpu_baseline --no_aux --profile_steps 30 2>&1 >/dev/null | grep "stage timing" | sed 's/.*stage timing done: //'
done
