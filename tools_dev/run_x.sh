set -e
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests/test_gpu_timed_path.py tests/test_gpu_parity.py tests/test_gpu_dp.py -q -m gpu -x > gpurun_out/r03_t.txt 2>&1 || { tail -40 gpurun_out/r03_t.txt; exit 1; }
tail -2 gpurun_out/r03_t.txt
B="python bench.py --no-configs --no-cpu-baseline --no-alt --no-strong-line --steps 200 --warmup 60"
run() { "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('   ', d['ms_per_step'])"; }
echo "cfg4 (unchanged path)"; run $B --config 4
echo "2x1024 bf16s B=4096"; run $B --config 2 --blocks 2 --batch 4096
echo "2x1024 bf16s B=4096 per-stage"; BLH_NO_BATCHED_WGRAD=1 run $B --config 2 --blocks 2 --batch 4096
echo "cfg2 graph"; run $B --config 2 --graph
