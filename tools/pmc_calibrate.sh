# FETCH_SIZE calibration on THIS kernel's access pattern (MI355X_MICROARCH.md: "calibrate on a known byte count"):
# one 128x128 tile with a very long K reads each operand exactly once (2 x 128 x K x 4 bytes, no reuse possible).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for SH in "nt,128,128,262144" "tn,128,128,262144" "nn,128,128,262144"; do
  rm -rf $R/gpurun_out/pmc_cal
  SUBGC_SPLITK=0 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_cal -- python $R/tools/gemm_bench.py --shape "$SH" --iters 2 > /dev/null 2>&1
  f=$(find $R/gpurun_out/pmc_cal -name "*counter_collection.csv" | head -1)
  echo "== $SH  (known bytes: $((2*128*262144*4)))"; python $R/tools/pmc_report.py $f gemm_f32_kernel | head -4
done
