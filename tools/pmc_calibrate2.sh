# FETCH_SIZE / WRITE_SIZE calibration (two separate --pmc passes) -> gpurun_out/r02_pmc_calibration.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
: > $O/r02_pmc_calibration.txt
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmc_cal2_$C
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_cal2_$C -- python $R/tools/pmc_calibrate2.py > $O/pmc_cal2_$C.log 2>&1
  f=$(find $O/pmc_cal2_$C -name "*counter_collection.csv" | head -1)
  python $R/tools/pmc_calibrate2_report.py $f $C >> $O/r02_pmc_calibration.txt
done
cat $O/r02_pmc_calibration.txt
