#!/bin/bash
# Rehearsal of bench.py's N > 1 control flow on a ONE-GPU box (gloo, ranks time-slice GPU 0).  The pool's process
# guard allows at most 6 processes with the card open (6 ranks + their launcher were killed by it, r06a), so the
# rehearsal runs 4 ranks (the 8-rank run is the driver's, on a real 8-GPU node); --strong-global-batch 2048 gives the 512 rows per rank an 8-way split of the
# headline batch has.  Usage: tools_dev/rehearse_ranks.sh <out-prefix>   (writes <out-prefix>_*.{json,log})
set -u
out=${1:-gpurun_out/r06_rehearse}
export BLH_BENCH_REHEARSE=1
echo "== 4 ranks, all healthy" | tee ${out}_ok.log
timeout -k 10 300 python bench.py --gpus 4 --steps 5 --warmup 2 --batch 256 --strong-global-batch 2048 \
    --no-cpu-baseline --pre-ramp-ms 20 > ${out}_ok.json 2>> ${out}_ok.log
echo "exit code $?" | tee -a ${out}_ok.log
echo "== 4 ranks, rank 3 raises after set-up" | tee ${out}_fail.log
BLH_BENCH_FAIL_RANK=3 timeout -k 10 300 python bench.py --gpus 4 --steps 5 --warmup 2 --batch 256 \
    --no-cpu-baseline --pre-ramp-ms 20 > ${out}_fail.json 2>> ${out}_fail.log
rc=$?
echo "exit code $rc (must be non-zero)" | tee -a ${out}_fail.log
sleep 1
left=$(ps -eo pid,args | grep -c "[b]ench.py --gpus 4")
echo "bench.py processes still alive after the parent returned: $left (must be 0)" | tee -a ${out}_fail.log
test "$rc" -ne 0 -a "$left" -eq 0
