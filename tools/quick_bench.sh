# Three train configs, headline numbers only (no CPU legs / decode / extras): python bench.py lines cut to the essentials.
R=${GRAFT_REPO_ROOT:-.}
for C in kar full_gc_kar flickr; do
  python $R/bench.py --config $C --steps 10 --warmup 3 --no-cpu-baseline --no-decode --packed-only --no-other-configs 2>&1 | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$C', d['value'], 'img/s', d['ms_per_step'], 'ms; gemm', r['gemm_ms_per_step'], 'ms', r['achieved'], 'TF/s frac', r['frac'], 'launches', r['launches_per_step'], 'fwd_bwd_only', d.get('fwd_bwd_only'), 'loss', d['final_loss'])"
done
