# PMC passes over one shape of the bf16 GEMM, both tile sizes -> gpurun_out/pmc_bf16.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
SHAPE=${1:-"nt,21760,9488,1000"}
: > $O/pmc_bf16.txt
for T in 128 256; do
 for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE"; do
  rm -rf $O/pmcb
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmcb -- python $R/tools/gemm_bf16_bench.py --shape "$SHAPE" --iters 3 --tile $T > /dev/null 2>&1
  f=$(find $O/pmcb -name "*counter_collection.csv" | head -1)
  echo "== tile $T" >> $O/pmc_bf16.txt
  python $R/tools/pmc_report.py $f gemm_bf16 >> $O/pmc_bf16.txt
 done
done
cat $O/pmc_bf16.txt
