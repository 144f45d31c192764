# timing-only ablations of the x3 GEMM main loop (results of these builds are WRONG by construction)
R=$GRAFT_REPO_ROOT
cp $R/sub-gc_amd/subgc/libsubgc_hip.so /tmp/lib_orig.so
for f in $R/sub-gc_amd/build/exp/lib_*.so; do
  cp $f $R/sub-gc_amd/subgc/libsubgc_hip.so
  echo "== $(basename $f)"
  SUBGC_GEMM_X3=1 timeout 120 python $R/tools/gemm_bench.py --only "logits" 2>&1 | grep logits
done
cp /tmp/lib_orig.so $R/sub-gc_amd/subgc/libsubgc_hip.so
