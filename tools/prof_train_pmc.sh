cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
ROUND=r02
rm -rf $O/prof_train
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-decode --packed-only > $O/prof_train.log 2>&1
timeout 120 python $R/tools/rocprof_summary.py $O/prof_train $O/${ROUND}_train_kernel_stats.txt > /dev/null
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmc_$C
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_$C -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-decode --packed-only > $O/pmc_$C.log 2>&1
done
LPS=$(python -c "import json;print(json.load(open('$R/profiles/${ROUND}_bench_n1.json'))['roofline']['launches_per_step'])")
ALG=$(python -c "import json;print(json.load(open('$R/profiles/${ROUND}_bench_n1.json'))['roofline']['algorithmic_bytes_per_launch'])")
python $R/tools/pmc_traffic.py $(find $O/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find $O/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1) \
   --launches-per-step $LPS --alg-bytes-per-launch $ALG > $O/${ROUND}_pmc_traffic.json 2> $O/${ROUND}_pmc_traffic.err
cat $O/${ROUND}_pmc_traffic.json $O/${ROUND}_pmc_traffic.err | head -40; tail -1 $O/prof_train.log | cut -c1-300
