# A/B of one bench.py switch over the three train configs in ONE job (same box): tools/ab_bench.sh "--no-p8" " " ...
R=${GRAFT_REPO_ROOT:-.}
for C in ${CONFIGS:-kar full_gc_kar flickr}; do
  for V in "$@"; do
    python $R/bench.py --config $C --steps ${STEPS:-12} --warmup 3 --no-cpu-baseline --no-decode --packed-only --no-other-configs $V 2>&1 | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$C', '[$V]', d['ms_per_step'], 'ms', d['value'], 'img/s; gemm', r['gemm_ms_per_step'], 'ms frac', r['frac'], 'launches', r['launches_per_step'], 'loss', d['final_loss'])"
  done
done
