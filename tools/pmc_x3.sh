cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE"; do
  rm -rf $R/gpurun_out/pmcx
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmcx -- python $R/tools/gemm_bench.py --only "logits" --iters 3 --mode bf16x3 > /dev/null 2>&1
  f=$(find $R/gpurun_out/pmcx -name "*counter_collection.csv" | head -1)
  python $R/tools/pmc_report.py $f gemm_f32_kernel | head -14
done
