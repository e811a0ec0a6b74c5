# fp32: single-stream backward (BLH_OPT_TWO_STREAM = 0) against the two-stream one, per batch
for rep in 1 2; do
for cfg in "--batch 2048" "--batch 4096" "--batch 8192" "--batch 16384 --blocks 4"; do
for v in "" "--one-stream"; do
python3 bench.py $cfg $v --no-configs --no-cpu-baseline --no-alt --steps 200 --warmup 50 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', '${v:-two-stream}', round(d['ms_per_step'],4))"
done; done; done
