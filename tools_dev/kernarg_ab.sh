# A/B of HIP_FORCE_DEV_KERNARG (kernel arguments in device memory) on three step shapes
set -e
O=gpurun_out/kernarg; mkdir -p $O
for v in 0 1 0 1; do
  for cfg in "--batch 64" "" "--batch 8192 --dtype bf16s --blocks 4"; do
    tag=$(echo "$cfg" | tr -d ' -'); tag=${tag:-headline}
    HIP_FORCE_DEV_KERNARG=$v python3 bench.py $cfg --no-configs --no-cpu-baseline --no-alt --steps 300 --warmup 50 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('devkernarg=$v', '$tag', round(d['ms_per_step'],4))" | tee -a $O/ab.txt
  done
done
