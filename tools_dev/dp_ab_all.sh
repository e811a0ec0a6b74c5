# all four shapes: data-parallel step before the wire against the fused step, interleaved (profiles/r05_dp_overhead.md)
set -e
tag=${1:-dpab}
for rep in a b; do
  for c in 1 2 3 4; do
    extra="--config $c"; [ $c = 1 ] && extra=""
    python3 bench.py $extra --no-configs --no-cpu-baseline --no-alt --no-strong-line --steps 300 --warmup 100 > gpurun_out/${tag}_fused_cfg${c}_$rep.json 2> gpurun_out/${tag}_fused_cfg${c}_$rep.err
    python3 bench.py $extra --gpus 1 --rehearse-rccl --no-configs --no-cpu-baseline --no-alt --no-strong-line --steps 300 --warmup 100 > gpurun_out/${tag}_rehearse_cfg${c}_$rep.json 2> gpurun_out/${tag}_rehearse_cfg${c}_$rep.err
  done
done
python3 - <<PY
import json
for c in (1,2,3,4):
    for rep in "ab":
        f=json.loads(open("gpurun_out/${tag}_fused_cfg%d_%s.json"%(c,rep)).read().strip().splitlines()[-1])["ms_per_step"]
        r=json.loads(open("gpurun_out/${tag}_rehearse_cfg%d_%s.json"%(c,rep)).read().strip().splitlines()[-1])["ms_per_step"]
        print("configs[%d] %s: fused %.4f ms, DP one rank every collective %.4f ms, +%.1f %%" % (c,rep,f,r,100*(r/f-1)))
PY
