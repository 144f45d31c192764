# PMC counters of named kernels over a short bench run:  KERNELS="attn_bwd_group attn_fwd_group" CONFIG=full_gc_kar bash tools/pmc_kernel.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
CONFIG=${CONFIG:-full_gc_kar}
for SET in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES"; do
  N=$(echo $SET | tr ' ' '_')
  rm -rf $O/pk_$N
  rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $O/pk_$N -- python $R/bench.py --config $CONFIG --steps 2 --warmup 1 --no-cpu-baseline --no-decode --packed-only --no-other-configs > $O/pk.log 2>&1
done
python - <<PY
import csv, glob, collections, os
ks = os.environ.get("KERNELS", "attn_bwd_group attn_fwd_group").split()
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob("$O/pk_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = next((x for x in ks if x in r["Kernel_Name"]), None)
        if k is None: continue
        a = agg[k][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k, d in agg.items():
    print(k)
    for c, (n, v) in sorted(d.items()): print(f"   {c:28s} per launch {v / n:16.1f}   ({n} launches)")
PY
rm -rf $O/pk_*
