# PMC passes over one shape of the fp32 GEMM -> gpurun_out/pmc_f32.txt   (usage: bash tools/pmc_f32.sh "nt,21760,4000,1000")
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
SHAPE=${1:-"nt,21760,4000,1000"}
: > $O/pmc_f32.txt
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
  rm -rf $O/pmcf
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmcf -- python $R/tools/gemm_bench.py --shape "$SHAPE" --iters 3 > /dev/null 2>&1
  f=$(find $O/pmcf -name "*counter_collection.csv" | head -1)
  python $R/tools/pmc_report.py $f gemm_f32 >> $O/pmc_f32.txt
done
cat $O/pmc_f32.txt
