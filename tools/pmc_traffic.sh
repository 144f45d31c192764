cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$C -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-decode --packed-only > $R/gpurun_out/pmc_$C.log 2>&1
  f=$(find $R/gpurun_out/pmc_$C -name "*counter_collection.csv" | head -1)
  echo "== $C $f"; python $R/tools/pmc_report.py $f gemm | head -40; python $R/tools/pmc_report.py $f splitk | head
  tail -2 $R/gpurun_out/pmc_$C.log | cut -c1-300
done
