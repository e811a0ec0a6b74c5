set -e
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM GRBM_GUI_ACTIVE"
for W in 1024 2048; do
  for tile in 256 128; do
    i=0
    for C in "$P1" "$P2" "FETCH_SIZE" "WRITE_SIZE"; do
      i=$((i+1))
      D=$R/gpurun_out/r03k_pmc/w${W}_t${tile}_p$i
      BLH_BF16S_TILE=$tile PYTHONPATH=$R rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -- python3 $R/bilinear_amd/csrc/tools/pmc_gemm.py 16384 $W bf16s > /dev/null 2>&1
      echo "== W=$W tile=$tile pass $i" >> $R/gpurun_out/r03k_pmc_summary.txt
      python3 $R/bilinear_amd/csrc/tools/pmc_gemm.py --sum $D >> $R/gpurun_out/r03k_pmc_summary.txt
    done
  done
done
