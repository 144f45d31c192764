# FETCH_SIZE / WRITE_SIZE of the fp32 GEMM on single shapes of the Sub_GC_Kar step -> gpurun_out/${ROUND}_pmc_traffic_shapes.txt
# (bytes past L2 per launch against the algorithmic bytes of the shape; FETCH x 2.0 on gfx950, KB -> bytes)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
ROUND=${ROUND:-r03}
OUT=$O/${ROUND}_pmc_traffic_shapes.txt
: > $OUT
for SH in "tn,4000,1000,7303" "tn,9488,1000,7303" "nt,7303,9488,1000" "nn,7303,1000,9488" "nt,7303,4000,1000" "nt,640,4000,3000" "nn,640,3000,4000" "nt,8320,1024,512" "nt,4736,512,1024"; do
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/pmcs_$C
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmcs_$C -- python $R/tools/gemm_bench.py --shape "$SH" --iters 4 > /dev/null 2>&1
  done
  python - "$SH" >> $OUT <<PY
import csv, glob, sys, collections
sh = sys.argv[1]; mode, M, N, K = sh.split(","); M, N, K = int(M), int(N), int(K)
def load(c):
    f = glob.glob("$O/pmcs_%s/**/*counter_collection.csv" % c, recursive=True)[0]
    tot, disp = collections.Counter(), collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        fam = "gemm" if "gemm_f32" in k else "reduce" if "splitk_reduce" in k else None
        if fam:
            tot[fam] += float(r["Counter_Value"]); disp[fam].add(r["Dispatch_Id"])
    return tot, {k: len(v) for k, v in disp.items()}
f, nf = load("FETCH_SIZE"); w, nw = load("WRITE_SIZE")
calls = nf.get("gemm", 1)
fetch = (f["gemm"] + f["reduce"]) * 2.0 * 1024 / calls
write = (w["gemm"] + w["reduce"]) * 1024 / calls
alg = 4.0 * (M * K + K * N + M * N)
print(f"{sh:24s} fetched {fetch / 1e6:8.1f} MB  written {write / 1e6:7.1f} MB  per call (incl. its reduce pass: {nf.get('reduce', 0) // max(calls, 1)}); algorithmic {alg / 1e6:7.1f} MB  -> x{(fetch + write) / alg:.2f}")
PY
done
rm -rf $O/pmcs_*
cat $OUT
