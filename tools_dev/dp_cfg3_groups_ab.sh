set -e
for rep in a b; do
  for mr in 4096 2048; do
    BLH_WGRAD_BATCHED_MIN_ROWS=$mr python3 bench.py --config 3 --gpus 1 --rehearse-rccl --no-configs --no-cpu-baseline --no-alt --no-strong-line --steps 300 --warmup 100 > gpurun_out/r05z_c3_${mr}_$rep.json 2> gpurun_out/r05z_c3_${mr}_$rep.err
    python3 -c "
import json; d=json.loads(open('gpurun_out/r05z_c3_${mr}_$rep.json').read().strip().splitlines()[-1]); print('cfg3 DP min_rows $mr $rep', d['ms_per_step'])"
  done
done
