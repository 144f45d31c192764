# FETCH_SIZE / WRITE_SIZE of the GEMM launches of the three throughput decode legs (greedy batched, beam-2 batched, MRNN top-k) ->
# gpurun_out/${ROUND}_pmc_decode_legs.json (ROUND=r04 by default); rocprofv3 kernel stats of the same commands beside it.
# Counters in SEPARATE --pmc passes with --kernel-trace only (MI355X_MICROARCH.md, HBM section).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
ROUND=${ROUND:-r05}
PASSES=${PASSES:-2}
for LEG in greedy_batched beam2_batched mrnn; do
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/pmcl_${LEG}_$C
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmcl_${LEG}_$C -- python $R/tools/decode_legs.py $LEG $PASSES > $O/pmcl_${LEG}_$C.log 2>&1
  done
  rm -rf $O/prof_leg_$LEG
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_leg_$LEG -- python $R/tools/decode_legs.py $LEG $PASSES > $O/prof_leg_$LEG.log 2>&1
  timeout 120 python $R/tools/rocprof_summary.py $O/prof_leg_$LEG $O/${ROUND}_decode_${LEG}_kernel_stats.txt > /dev/null
done
python - <<PY
import csv, glob, json, collections
FAM = ("gemm_f32_kernel", "gemm_f32_splitk_kernel", "splitk_reduce_kernel", "gemm_skinny")
def load(leg, c):
    f = glob.glob("$O/pmcl_%s_%s/**/*counter_collection.csv" % (leg, c), recursive=True)[0]
    tot, disp = collections.defaultdict(float), collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != c:
            continue
        fam = next((x for x in FAM if x in r["Kernel_Name"]), None)
        if fam is None:
            continue
        tot[fam] += float(r["Counter_Value"]); disp[fam].add(r["Dispatch_Id"])
    return tot, {k: len(v) for k, v in disp.items()}
out = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace) over tools/decode_legs.py <leg> $PASSES (1 warm-up + $PASSES passes); "
                 "FETCH x 2.0 (gfx950 halving, profiles/r02_pmc_calibration.txt), KiB -> bytes; one launch = one main GEMM kernel (+ its split-K reduce)"}
for leg in ("greedy_batched", "beam2_batched", "mrnn"):
    f, fd = load(leg, "FETCH_SIZE"); w, wd = load(leg, "WRITE_SIZE")
    mains = sum(v for k, v in fd.items() if k != "splitk_reduce_kernel")
    passes = $PASSES + 1
    calls = None                                   # GEMM calls per pass as the leg itself counted them (a call may be two main kernels: row cut)
    for line in open("$O/pmcl_%s_FETCH_SIZE.log" % leg, errors="replace"):
        if line.startswith("gemm_calls_per_pass"):
            calls = int(line.split()[1])
    per_pass = calls if calls else mains // passes
    launches = per_pass * passes
    out[leg] = {"gemm_main_dispatches": mains, "passes_profiled": passes, "gemm_launches_per_pass": per_pass,
                "main_kernels_per_launch": mains / max(launches, 1),
                "traffic_bytes_per_launch": (2.0 * 1024 * sum(f.values()) + 1024 * sum(w.values())) / max(launches, 1),
                "fetch_bytes_per_launch": 2.0 * 1024 * sum(f.values()) / max(launches, 1), "write_bytes_per_launch": 1024 * sum(w.values()) / max(launches, 1),
                "kernel_dispatches": fd}
json.dump(out, open("$O/${ROUND}_pmc_decode_legs.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
