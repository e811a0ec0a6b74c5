set -e
bash tools_dev/prof_cfg.sh r03_cfg2_add --config 2
bash tools_dev/prof_cfg.sh r03_cfg4_add --config 4
B="python bench.py --no-configs --no-cpu-baseline --no-alt --no-strong-line --steps 300 --warmup 100"
run() { "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('   ', d['ms_per_step'])"; }
echo cfg2; run $B --config 2
echo cfg4; run $B --config 4
timeout -k 10 300 python -m pytest tests/test_gpu_bf16s.py -q -m gpu -x -k "gemm or layouts or epilogue" 2>&1 | tail -2
