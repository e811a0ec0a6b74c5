set -e
B="python bench.py --no-configs --no-cpu-baseline --no-alt --no-strong-line --steps 300 --warmup 100"
run() { "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('   ', d['ms_per_step'])"; }
echo cfg3shape; run $B --config 2 --batch 8192
echo cfg3shape; run $B --config 2 --batch 8192
timeout -k 10 600 python -m pytest tests/test_gpu_bf16s.py tests/test_gpu_timed_path.py -q -m gpu -x -k "bf16s" > gpurun_out/r03_t.txt 2>&1 || true
tail -2 gpurun_out/r03_t.txt
