cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
for C in flickr kar full_gc_kar; do
rm -rf $O/gap_$C
rocprofv3 --kernel-trace --output-format csv -d $O/gap_$C -- python $R/bench.py --config $C --steps 12 --warmup 3 --no-cpu-baseline --no-decode --packed-only --no-other-configs > $O/gap_$C.log 2>&1
echo "== $C"; python $R/tools/gap_report.py $O/gap_$C 4 0.6 | head -22
rm -rf $O/gap_$C
done
