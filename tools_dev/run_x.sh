set -e
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests/test_gpu_timed_path.py tests/test_gpu_bf16s.py tests/test_gpu_parity.py -q -m gpu -x -k "bf16s or captured or config" > gpurun_out/r03_t.txt 2>&1 || { tail -40 gpurun_out/r03_t.txt; exit 1; }
tail -2 gpurun_out/r03_t.txt
bash tools_dev/final_measure.sh r03d > gpurun_out/final_r03d.log 2>&1; tail -2 gpurun_out/final_r03d.log
