# End-of-round evidence (ROUND=r04 by default): the default bench line, the bf16 config lines, rocprofv3 kernel stats of the
# train legs and of the one-image decode, and the two PMC passes behind roofline.traffic.  Everything lands in gpurun_out/ as
# ${ROUND}_*; copy the summaries to profiles/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
ROUND=${ROUND:-r06}
python $R/bench.py > $O/${ROUND}_bench_n1.log 2>&1; tail -1 $O/${ROUND}_bench_n1.log > $R/profiles/${ROUND}_bench_n1.json
cp $R/profiles/${ROUND}_bench_n1.json $O/${ROUND}_bench_n1.json          # profiles/ on the box is not merged back, gpurun_out/ is
for C in full_gc_kar flickr kar_ss25; do
  python $R/bench.py --config $C --steps 10 --warmup 3 > $O/${ROUND}_bench_$C.log 2>&1; tail -1 $O/${ROUND}_bench_$C.log > $O/${ROUND}_bench_$C.json
done
prof() {  # name, command...
  local name=$1; shift
  rm -rf $O/prof_$name
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -- "$@" > $O/prof_$name.log 2>&1
  timeout 120 python $R/tools/rocprof_summary.py $O/prof_$name $O/${ROUND}_${name}_kernel_stats.txt > /dev/null
}
prof train python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-decode --packed-only --no-other-configs
prof full_gc_kar python $R/bench.py --config full_gc_kar --steps 8 --warmup 2
prof flickr python $R/bench.py --config flickr --steps 8 --warmup 2
prof decode python $R/tools/decode_bench.py
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmc_$C
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_$C -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-decode --packed-only --no-other-configs > $O/pmc_$C.log 2>&1
done
LPS=$(python -c "import json;print(json.load(open('$R/profiles/${ROUND}_bench_n1.json'))['roofline']['launches_per_step'])")
ALG=$(python -c "import json;print(json.load(open('$R/profiles/${ROUND}_bench_n1.json'))['roofline']['algorithmic_bytes_per_launch'])")
python $R/tools/pmc_traffic.py $(find $O/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find $O/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1) \
   --launches-per-step $LPS --alg-bytes-per-launch $ALG > $O/${ROUND}_pmc_traffic.json 2> $O/${ROUND}_pmc_traffic.err
# the same two PMC passes for the bf16 configs and the scheduled-sampling leg
for C in full_gc_kar flickr kar_ss25; do
  for P in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/pmc_${C}_$P
    rocprofv3 --pmc $P --kernel-trace --output-format csv -d $O/pmc_${C}_$P -- python $R/bench.py --config $C --steps 3 --warmup 2 --no-cpu-baseline --no-decode --packed-only > $O/pmc_${C}_$P.log 2>&1
  done
  LPS=$(python -c "import json;print(json.load(open('$O/${ROUND}_bench_$C.json'))['roofline']['launches_per_step'])")
  ALG=$(python -c "import json;print(json.load(open('$O/${ROUND}_bench_$C.json'))['roofline']['algorithmic_bytes_per_launch'])")
  python $R/tools/pmc_traffic.py $(find $O/pmc_${C}_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find $O/pmc_${C}_WRITE_SIZE -name "*counter_collection.csv" | head -1) \
     --launches-per-step $LPS --alg-bytes-per-launch $ALG > $O/${ROUND}_pmc_traffic_$C.json 2> $O/${ROUND}_pmc_traffic_$C.err
done
cut -c1-700 $R/profiles/${ROUND}_bench_n1.json; cut -c1-400 $O/${ROUND}_bench_full_gc_kar.json; cut -c1-400 $O/${ROUND}_bench_flickr.json
head -14 $O/${ROUND}_train_kernel_stats.txt | cut -c1-150; head -12 $O/${ROUND}_full_gc_kar_kernel_stats.txt | cut -c1-150; cat $O/${ROUND}_pmc_traffic.json | head -20
ROUND=$ROUND bash $R/tools/pmc_decode_legs.sh > $O/${ROUND}_pmc_decode_legs.log 2>&1      # GEMM traffic + kernel stats of the three throughput decode legs
ROUND=$ROUND bash $R/tools/pmc_decode.sh > $O/${ROUND}_pmc_decode.log 2>&1                # FETCH / WRITE of the one-image greedy loop -> ${ROUND}_pmc_decode.json
