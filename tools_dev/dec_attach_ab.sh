set -e
python3 -m pytest tests/test_gpu_timed_path.py -x -q -m gpu > gpurun_out/r05t_tests.log 2>&1 || { tail -20 gpurun_out/r05t_tests.log; exit 1; }
tail -2 gpurun_out/r05t_tests.log
for rep in 1 2 3; do
  python3 bench.py --no-configs --no-cpu-baseline --no-alt --no-strong-line --steps 1000 --warmup 200 > gpurun_out/r05t_on_$rep.json 2>/dev/null
  BLH_NO_DEC_ATTACH=1 python3 bench.py --no-configs --no-cpu-baseline --no-alt --no-strong-line --steps 1000 --warmup 200 > gpurun_out/r05t_off_$rep.json 2>/dev/null
done
python3 - <<PY
import json
for k in ("on","off"):
    print(k, [round(json.loads(open("gpurun_out/r05t_%s_%d.json"%(k,r)).read().strip().splitlines()[-1])["ms_per_step"],4) for r in (1,2,3)])
PY
