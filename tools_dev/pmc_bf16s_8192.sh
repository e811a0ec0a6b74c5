# FETCH_SIZE / WRITE_SIZE of the bf16-storage forward GEMM at M = 8192, W = 1024 (configs[3] per GPU: the 128x128 kernel)
set -e
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -f $R/gpurun_out/r03_pmc_8192_summary.txt
i=0
for C in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  D=$R/gpurun_out/r03_pmc_8192/p$i
  PYTHONPATH=$R rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -- python3 $R/bilinear_amd/csrc/tools/pmc_gemm.py 8192 1024 bf16s > /dev/null 2>&1
  echo "== M=8192 W=1024 pass $C" >> $R/gpurun_out/r03_pmc_8192_summary.txt
  python3 $R/bilinear_amd/csrc/tools/pmc_gemm.py --sum $D >> $R/gpurun_out/r03_pmc_8192_summary.txt
done
cat $R/gpurun_out/r03_pmc_8192_summary.txt
