set -e
python3 -m pytest tests/test_gpu_timed_path.py tests/test_gpu_parity.py -x -q -m gpu -k "not compile" > gpurun_out/r06r_tests.log 2>&1 || { tail -20 gpurun_out/r06r_tests.log; exit 1; }
tail -2 gpurun_out/r06r_tests.log
for rep in 1 2 3; do
  python3 bench.py --no-configs --no-cpu-baseline --no-alt --no-strong-line --steps 1000 --warmup 200 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('headline', round(d['ms_per_step'],4))"
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r06r_prof -o f -- python3 $GRAFT_REPO_ROOT/bench.py --no-configs --no-cpu-baseline --no-alt --no-strong-line --steps 30 --warmup 10 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; python3 tools_dev/prof_summary.py gpurun_out/r06r_prof | grep -E "finalize|total"
