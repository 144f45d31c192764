cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
prof() { local name=$1; shift; rm -rf $O/prof_$name
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -- "$@" > $O/prof_$name.log 2>&1
  timeout 120 python $R/tools/rocprof_summary.py $O/prof_$name $O/r05a_${name}_kernel_stats.txt > /dev/null; rm -rf $O/prof_$name; }
prof full_gc_kar python $R/bench.py --config full_gc_kar --steps 8 --warmup 2
prof flickr python $R/bench.py --config flickr --steps 8 --warmup 2
prof train python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-decode --packed-only --no-other-configs
head -40 $O/r05a_full_gc_kar_kernel_stats.txt | cut -c1-160
