# usage: [REPS=50] bash tools/ab/bt.sh <lib tags...>  -- omfs_composite_bwd alone (tools/bwd_time.py, DPP implementation) per prebuilt library
for v in "$@"; do
  cp tools/_ab/so/$v.so omfs_4d_video_gen_amd/libomfs_splat.so
  python tools/bwd_time.py --impls dpp --tag $v --reps ${REPS:-50} 2>/dev/null | tail -1
done
