cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
prof() { local name=$1; shift; rm -rf $O/prof_$name
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -- "$@" > $O/prof_$name.log 2>&1
  timeout 120 python $R/tools/rocprof_summary.py $O/prof_$name $O/r05b_${name}_kernel_stats.txt > /dev/null; rm -rf $O/prof_$name; }
for C in kar full_gc_kar flickr; do
prof $C python $R/bench.py --config $C --steps 6 --warmup 2 --no-cpu-baseline --no-decode --packed-only --no-other-configs
grep -E "mid_|lstm_fwd|attn_fwd" $O/r05b_${C}_kernel_stats.txt | cut -c1-120
done
