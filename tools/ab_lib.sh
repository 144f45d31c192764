# Same-box A/B of two BUILDS of libsubgc_hip.so (boxes differ by +-2 %, two runs in one job do not): build the library at two revisions into
# sub-gc_amd/subgc/lib_a.so and lib_b.so (git-ignored, they travel with gpurun), then on the GPU box
#   bash tools/ab_lib.sh ["full_gc_kar flickr kar"] [repeats]
# copies each over libsubgc_hip.so in turn and prints value / ms per step / GEMM ms of the train legs.  How the dropout-mask reduce pass and
# the 15-part cut were judged in round 6.  Restore the library afterwards (python sub-gc_amd/build.py --force).
R=${GRAFT_REPO_ROOT:-.}; L=$R/sub-gc_amd/subgc
for r in $(seq 1 ${2:-2}); do for v in a b; do cp $L/lib_$v.so $L/libsubgc_hip.so; for C in ${1:-full_gc_kar flickr kar}; do
  python $R/bench.py --config $C --steps 10 --warmup 3 --no-cpu-baseline --no-decode --packed-only --no-other-configs 2>&1 | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v $C', d['value'], d['ms_per_step'], d['roofline']['gemm_ms_per_step'])"
done; done; done
