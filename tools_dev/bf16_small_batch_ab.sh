# bf16 storage at small batches: batched weight gradients allowed below 4096 rows per workgroup / 224 workgroups
# (api_layout.h: wgrad_batched_plan_h; BLH_WGRAD_BATCHED_MIN_ROWS / _MIN_WGS)
set -e
tag=${1:-bfsb}
for nb in 2 4; do
for b in 1024 2048 3072 4096 6144; do
  line="bf16s ${nb}x1024 batch $b:"
  for k in "4096 224" "2048 64" "1024 64" "512 64"; do
    set -- $k
    export BLH_WGRAD_BATCHED_MIN_ROWS=$1 BLH_WGRAD_BATCHED_MIN_WGS=$2
    python3 bench.py --blocks $nb --width 1024 --batch $b --dtype bf16s --no-configs --no-cpu-baseline --no-alt --no-strong-line --steps 500 --warmup 100 > gpurun_out/${tag}_tmp.json 2>/dev/null
    v=$(python3 -c "
import json; d=json.loads(open('gpurun_out/${tag}_tmp.json').read().strip().splitlines()[-1]); print('%.4f' % d['ms_per_step'])")
    line="$line  [$1/$2] $v"
  done
  echo "$line"
done
done
