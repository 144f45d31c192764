# A/B timing of two builds of the library placed in sub-gc_amd/build/exp/lib_*.so
R=$GRAFT_REPO_ROOT
cp $R/sub-gc_amd/subgc/libsubgc_hip.so /tmp/lib_orig.so
for f in $R/sub-gc_amd/build/exp/lib_*.so; do
  cp $f $R/sub-gc_amd/subgc/libsubgc_hip.so
  echo "== $(basename $f)"
  timeout 200 python $R/tools/gemm_bench.py ${GEMM_ARGS} 2>&1 | tail -${TAILN:-21} | cut -c1-75
done
cp /tmp/lib_orig.so $R/sub-gc_amd/subgc/libsubgc_hip.so
