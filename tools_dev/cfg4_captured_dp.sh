# r06: BASELINE configs[4] as it is named — "overlapped all-reduce + hipGraph-captured train step" — at its per-GPU shape
# on one rank with every collective issued: eager and captured, torch-driven and library-driven collectives.
set -e
tag=${1:-cfg4cap}
common="--config 4 --gpus 1 --rehearse-rccl --no-configs --no-cpu-baseline --no-alt --no-strong-line --steps 200 --warmup 50"
for rep in a b; do
  python3 bench.py --config 4 --no-configs --no-cpu-baseline --no-alt --no-strong-line --steps 200 --warmup 50 > gpurun_out/${tag}_fused_$rep.json 2> gpurun_out/${tag}_fused_$rep.err
  python3 bench.py $common > gpurun_out/${tag}_torch_eager_$rep.json 2> gpurun_out/${tag}_torch_eager_$rep.err
  python3 bench.py $common --graph > gpurun_out/${tag}_torch_graph_$rep.json 2> gpurun_out/${tag}_torch_graph_$rep.err
  python3 bench.py $common --native-rccl > gpurun_out/${tag}_native_eager_$rep.json 2> gpurun_out/${tag}_native_eager_$rep.err
  python3 bench.py $common --native-rccl --graph > gpurun_out/${tag}_native_graph_$rep.json 2> gpurun_out/${tag}_native_graph_$rep.err
done
python3 - "$tag" > gpurun_out/${tag}_table.txt <<'PY'
import json, sys
tag = sys.argv[1]
def rec(kind, rep):
    d = json.loads(open("gpurun_out/%s_%s_%s.json" % (tag, kind, rep)).read().strip().splitlines()[-1])
    return d["ms_per_step"], d["config"].get("launch", "")
for kind in ("fused", "torch_eager", "torch_graph", "native_eager", "native_graph"):
    a, la = rec(kind, "a"); b, _ = rec(kind, "b")
    print("%-13s %.4f / %.4f ms   (%s)" % (kind, a, b, la[:70]))
PY
cat gpurun_out/${tag}_table.txt
