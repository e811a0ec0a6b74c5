# 385 .. 2048 rows, fp32 2 x 1024: step time with the pairwise-merged launches against BLH_NO_MID_PAIR=1, interleaved
set -e
tag=${1:-mid}
for rep in a b; do
  for b in 512 1024; do
    for k in on off; do
      if [ $k = off ]; then export BLH_NO_MID_PAIR=1; else unset BLH_NO_MID_PAIR; fi
      python3 bench.py --batch $b --no-configs --no-cpu-baseline --no-alt --no-strong-line --steps 1000 --warmup 200 > gpurun_out/${tag}_${b}_${k}_$rep.json 2>/dev/null
    done
  done
done
unset BLH_NO_MID_PAIR
python3 - <<PY
import json
for b in (512,1024):
    row=[]
    for k in ("on","off"):
        row.append([round(json.loads(open("gpurun_out/${tag}_%d_%s_%s.json"%(b,k,r)).read().strip().splitlines()[-1])["ms_per_step"],4) for r in "ab"])
    print("batch %5d: merged %s  multi-launch %s" % (b,row[0],row[1]))
PY
