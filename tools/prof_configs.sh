# rocprofv3 kernel stats (and optionally the idle-gap report) of the train configs in one job:
#   CONFIGS="kar full_gc_kar flickr" TAG=r05x [GAPS=1] [GREP="attn_|lstm_"] [EXTRA="--no-p8"] bash tools/prof_configs.sh
# -> gpurun_out/${TAG}_${config}_kernel_stats.txt (tools/rocprof_summary.py), gpurun_out/${TAG}_${config}_gaps.txt (tools/gap_report.py)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=${TAG:-prof}
for C in ${CONFIGS:-kar full_gc_kar flickr}; do
  rm -rf $O/prof_$C
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$C -- python $R/bench.py --config $C --steps ${STEPS:-8} --warmup 2 --no-cpu-baseline --no-decode \
      --packed-only --no-other-configs $EXTRA > $O/prof_$C.log 2>&1
  timeout 120 python $R/tools/rocprof_summary.py $O/prof_$C $O/${TAG}_${C}_kernel_stats.txt > /dev/null
  [ -n "$GAPS" ] && python $R/tools/gap_report.py $O/prof_$C 4 0.6 > $O/${TAG}_${C}_gaps.txt
  echo "== $C"; head -4 $O/${TAG}_${C}_kernel_stats.txt | cut -c1-160
  if [ -n "$GREP" ]; then grep -E "$GREP" $O/${TAG}_${C}_kernel_stats.txt | cut -c1-130; else sed -n 5,16p $O/${TAG}_${C}_kernel_stats.txt | cut -c1-130; fi
  [ -n "$GAPS" ] && head -3 $O/${TAG}_${C}_gaps.txt
  rm -rf $O/prof_$C
done
