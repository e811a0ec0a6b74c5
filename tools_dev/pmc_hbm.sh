# PMC passes for the HBM-bound kernels of the step (skinny projections, BatchNorm kernels, clip + Adam):
# FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 runs (--pmc with --kernel-trace only, the program directly
# behind `--`, as MI355X_MICROARCH.md prescribes), each over one short bench.py run of configs[1] (fp32) and
# one each of configs[2], [3], [4] (bf16 storage; PMC_CFGS selects); bench.py's stand-alone timing loops (roofline, roofline_hbm) are inside
# the profiled run, so the four blh_skinny_* entry points are covered in isolation as well as inside the step.
# usage (GPU box): bash tools_dev/pmc_hbm.sh [tag]  -> gpurun_out/<tag>_pmc_hbm/{summary.txt,hbm_traffic.json}
set -e
tag=${1:-r05}
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${tag}_pmc_hbm
mkdir -p $O
rm -f $O/summary.txt
for cfg in ${PMC_CFGS:-1 2 3 4}; do
  for C in FETCH_SIZE WRITE_SIZE; do
    D=$O/c${cfg}_$C
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -- python3 $R/bench.py --config $cfg --no-configs --no-alt --no-cpu-baseline --no-strong-line --steps 10 --warmup 3 --pre-ramp-ms 0 > $D.stdout 2> $D.stderr
    echo "== config $cfg pass $C" >> $O/summary.txt
    python3 $R/tools_dev/pmc_sum.py $D >> $O/summary.txt
    echo "config $cfg $C done"
  done
done
python3 $R/tools_dev/pmc_hbm_json.py $O/summary.txt > $O/hbm_traffic.json
