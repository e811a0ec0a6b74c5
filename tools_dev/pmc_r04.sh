# Round-4 PMC passes (separate --pmc runs with --kernel-trace only, as MI355X_MICROARCH.md prescribes):
# HBM-side traffic (FETCH_SIZE, WRITE_SIZE) and matrix-pipe busy of the shipped hidden-layer GEMMs,
# fp32 at B = 4096 and bf16 storage at the configs[2] / configs[3] / configs[4] per-GPU shapes.
# usage (GPU box): bash tools_dev/pmc_r04.sh   -> gpurun_out/r04_pmc/summary.txt, gpurun_out/r04_pmc/traffic.json
set -e
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_pmc
mkdir -p $O
rm -f $O/summary.txt
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS"
for shape in "4096 1024 all" "8192 1024 bf16s" "16384 1024 bf16s" "16384 2048 bf16s"; do
  set -- $shape
  i=0
  for C in "FETCH_SIZE" "WRITE_SIZE" "$P1"; do
    i=$((i+1))
    D=$O/m$1_w$2_p$i
    PYTHONPATH=$R rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -- python3 $R/bilinear_amd/csrc/tools/pmc_gemm.py $1 $2 $3 > /dev/null 2>&1
    echo "== M=$1 W=$2 pass $i ($C)" >> $O/summary.txt
    python3 $R/bilinear_amd/csrc/tools/pmc_gemm.py --sum $D >> $O/summary.txt
  done
done
python3 $R/tools_dev/pmc_r04_json.py $O/summary.txt > $O/traffic.json
tail -5 $O/traffic.json
