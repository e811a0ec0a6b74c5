# A/B of the fp32 K9 epilogue (default) against BLH_NO_K9=1, whole step
set -e
O=gpurun_out/k9f32; mkdir -p $O
for rep in 1 2; do
  for v in k9 no_k9; do
    for cfg in "" "--batch 2048" "--batch 8192" "--batch 16384" "--rehearse-rccl"; do
      tag=$(echo "$cfg" | tr -d ' -'); tag=${tag:-headline}
      if [ $v = no_k9 ]; then export BLH_NO_K9=1; else unset BLH_NO_K9; fi
      python3 bench.py $cfg --no-configs --no-cpu-baseline --no-alt --steps 300 --warmup 50 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', '$tag', round(d['ms_per_step'],4))" | tee -a $O/ab.txt
    done
  done
done
