# r06: the data-parallel step before the wire — fused step vs torch-driven collectives vs collectives issued by the
# library (blh_train_step_dp, both tails), same box, interleaved (profiles/r06_dp_overhead.md).
# usage: bash tools_dev/dp_native_ab.sh <tag> [configs...]   -> gpurun_out/<tag>_*.json + <tag>_table.txt
set -e
tag=${1:-dpn}; shift || true
cfgs=${@:-1 2 3 4}
common="--no-configs --no-cpu-baseline --no-alt --no-strong-line --steps 300 --warmup 100"
for rep in a b; do
  for c in $cfgs; do
    extra="--config $c"; [ $c = 1 ] && extra=""
    python3 bench.py $extra $common > gpurun_out/${tag}_fused_cfg${c}_$rep.json 2> gpurun_out/${tag}_fused_cfg${c}_$rep.err
    python3 bench.py $extra --gpus 1 --rehearse-rccl $common > gpurun_out/${tag}_torch_cfg${c}_$rep.json 2> gpurun_out/${tag}_torch_cfg${c}_$rep.err
    python3 bench.py $extra --gpus 1 --rehearse-rccl --native-rccl $common > gpurun_out/${tag}_native_cfg${c}_$rep.json 2> gpurun_out/${tag}_native_cfg${c}_$rep.err
    python3 bench.py $extra --gpus 1 --rehearse-rccl --native-rccl --native-tail comm $common > gpurun_out/${tag}_nativecs_cfg${c}_$rep.json 2> gpurun_out/${tag}_nativecs_cfg${c}_$rep.err
    echo "configs[$c] $rep done"
  done
done
python3 - "$tag" $cfgs > gpurun_out/${tag}_table.txt <<'PY'
import json, sys
tag, cfgs = sys.argv[1], [int(c) for c in sys.argv[2:]]
def ms(kind, c, rep):
    return json.loads(open("gpurun_out/%s_%s_cfg%d_%s.json" % (tag, kind, c, rep)).read().strip().splitlines()[-1])["ms_per_step"]
print("| config | run | fused ms | torch process group | library RCCL, tail on the producer stream | library RCCL, tail on the comm stream |")
print("|---|---|---|---|---|---|")
for c in cfgs:
    for rep in "ab":
        f = ms("fused", c, rep)
        row = ["%.4f (+%.1f %%)" % (ms(k, c, rep), 100 * (ms(k, c, rep) / f - 1)) for k in ("torch", "native", "nativecs")]
        print("| configs[%d] | %s | %.4f | %s |" % (c, rep, f, " | ".join(row)))
PY
cat gpurun_out/${tag}_table.txt
