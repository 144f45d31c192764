# End-of-round evidence: the default bench line, rocprofv3 kernel stats of the train leg and of batched decode,
# and the two PMC passes behind roofline.traffic.  Everything lands in gpurun_out/; copy the summaries to profiles/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
python $R/bench.py > $O/bench_n1.log 2>&1; tail -1 $O/bench_n1.log > $O/bench_n1.json
rm -rf $O/prof_train $O/prof_decode $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-decode --packed-only > $O/prof_train.log 2>&1
timeout 120 python $R/tools/rocprof_summary.py $O/prof_train $O/train_kernel_stats.txt > /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_decode -- python $R/tools/decode_bench.py > $O/prof_decode.log 2>&1
timeout 120 python $R/tools/rocprof_summary.py $O/prof_decode $O/decode_kernel_stats.txt > /dev/null
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_$C -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-decode --packed-only > $O/pmc_$C.log 2>&1
done
cut -c1-600 $O/bench_n1.json; head -12 $O/train_kernel_stats.txt | cut -c1-150; tail -3 $O/prof_decode.log | cut -c1-300
