# r06: gradient-norm partials from the batched slab sum (bf16 storage) against the separate sumsq pass, interleaved
set -e
tag=${1:-fold}
common="--no-configs --no-cpu-baseline --no-alt --no-strong-line --steps 300 --warmup 100"
for rep in a b; do
  for c in 2 3; do
    python3 bench.py --config $c $common > gpurun_out/${tag}_on_cfg${c}_$rep.json 2> gpurun_out/${tag}_on_cfg${c}_$rep.err
    BLH_NO_SUMSQ_FOLD=1 python3 bench.py --config $c $common > gpurun_out/${tag}_off_cfg${c}_$rep.json 2> gpurun_out/${tag}_off_cfg${c}_$rep.err
  done
done
python3 - "$tag" > gpurun_out/${tag}_table.txt <<'PY'
import json, sys
tag = sys.argv[1]
def ms(kind, c, rep):
    return json.loads(open("gpurun_out/%s_%s_cfg%d_%s.json" % (tag, kind, c, rep)).read().strip().splitlines()[-1])["ms_per_step"]
for c in (2, 3):
    for rep in "ab":
        print("configs[%d] %s: norm partials from the slab sum %.4f ms, separate sumsq pass %.4f ms (%+.2f %%)" % (
            c, rep, ms("on", c, rep), ms("off", c, rep), 100 * (ms("on", c, rep) / ms("off", c, rep) - 1)))
PY
cat gpurun_out/${tag}_table.txt
