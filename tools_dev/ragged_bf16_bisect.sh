# which switch makes the ragged-batch bug of the fused bf16 step go away?  (B = 385: wrong results, no fault)
for env in "" "BLH_NO_ENCODE_FUSE=1" "BLH_NO_K9=1" "BLH_ONE_STREAM=1" "BLH_NO_SUMSQ_FOLD=1" "BLH_NO_DECODE_FUSE=1"; do
  echo "== $env"
  env $env RAGGED_BATCHES="385 1025" timeout -k 10 100 python3 tools_dev/ragged_bf16_debug.py 2>&1 | grep "^B "
done
