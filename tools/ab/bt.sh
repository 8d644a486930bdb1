# usage: [REPS=50] bash tools/ab/bt.sh <lib tags...>  -- omfs_composite_bwd alone (tools/bwd_time.py) per prebuilt library
# (candidate bound through OMFS_LIB_PATH: the in-tree library is never overwritten)
for v in "$@"; do
  OMFS_LIB_PATH=$PWD/tools/_ab/so/$v.so python tools/bwd_time.py --impls ${IMPLS:-dpp} --tag $v --reps ${REPS:-50} 2>/dev/null | tail -1
done
