# bf16 storage: one-pass decode against BLH_NO_DECODE_FUSE=1, interleaved (profiles/r05_decode_fused.md)
set -e
tag=${1:-dech}
for rep in a b; do
  for c in 2 3; do
    for k in on off; do
      if [ $k = off ]; then export BLH_NO_DECODE_FUSE=1; else unset BLH_NO_DECODE_FUSE; fi
      python3 bench.py --config $c --no-configs --no-cpu-baseline --no-alt --no-strong-line --steps 600 --warmup 100 > gpurun_out/${tag}_c${c}_${k}_$rep.json 2>/dev/null
    done
  done
done
unset BLH_NO_DECODE_FUSE
python3 - <<PY
import json
for c in (2,3):
    row=[]
    for k in ("on","off"):
        row.append([round(json.loads(open("gpurun_out/${tag}_c%d_%s_%s.json"%(c,k,r)).read().strip().splitlines()[-1])["ms_per_step"],4) for r in "ab"])
    print("configs[%d]: one-pass decode %s  separate %s" % (c,row[0],row[1]))
PY
