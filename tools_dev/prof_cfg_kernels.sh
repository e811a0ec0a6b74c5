#!/bin/bash
# per-kernel table of one BASELINE config's step (rocprofv3 --kernel-trace): tools_dev/prof_cfg_kernels.sh <cfg> <out-prefix>
cfg=$1; out=$2; R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $R/${out}_prof -o t -- python3 $R/bench.py --config $cfg --no-configs --no-cpu-baseline --no-alt --steps 100 --warmup 20 --pre-ramp-ms 50 > $R/${out}.json 2> $R/${out}.log
python3 $R/tools_dev/prof_kernels.py $R/${out}_prof/t_results.db blh > $R/${out}_kernels.txt
