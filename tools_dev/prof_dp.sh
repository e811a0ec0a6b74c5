set -e
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03w_dp_bf16 -o f -- python3 $GRAFT_REPO_ROOT/tools_dev/dp_overhead.py 4096 "bf16 buckets" > $GRAFT_REPO_ROOT/gpurun_out/r03w_dp_bf16.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03w_dp_forced -o f -- python3 $GRAFT_REPO_ROOT/tools_dev/dp_overhead.py 4096 "collectives forced" > $GRAFT_REPO_ROOT/gpurun_out/r03w_dp_forced.txt 2>&1
