# FETCH_SIZE / WRITE_SIZE of the one-image decode loop's weight-streaming kernels -> gpurun_out/${ROUND}_pmc_decode.json (ROUND=r06 by default)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
ROUND=${ROUND:-r06}
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmcd_$C
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmcd_$C -- python $R/tools/decode_bench.py 16 50 > $O/pmcd_$C.log 2>&1
done
python - <<PY
import csv, glob, json, collections
def load(c):
    f = glob.glob("$O/pmcd_%s/**/*counter_collection.csv" % c, recursive=True)[0]
    d = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        import re
        if "lstm_cell_pick_kernel" in k:
            key = "cell_pick"
        elif "gemm_skinny_mfma_kernel" in k:
            a = [x.strip() for x in re.search(r"gemm_skinny_mfma_kernel<([^>]*)>", k).group(1).split(",")]     # WAVES, D, LSTM, MT, PICK, WB16, DUAL
            flag = lambda i: len(a) > i and a[i] == "true"
            lstm, mt, pick, dual = flag(2), a[3], flag(4), flag(6)
            key = ("dual_pick" if pick else "dual") if dual else ("lstm" if lstm else ("plain_1tile" if mt == "1" else "other"))
        else:
            continue
        d[key][0] += 1; d[key][1] += float(r["Counter_Value"])
    return d
f, w = load("FETCH_SIZE"), load("WRITE_SIZE")
out = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over tools/decode_bench.py 16 50 (tools/pmc_decode.sh); FETCH x 2.0 (gfx950), KB -> bytes",
       "kernels": {k: {"launches": f[k][0], "hbm_fetch_bytes_per_launch": round(f[k][1] * 2.0 * 1024 / max(f[k][0], 1)),
                       "hbm_write_bytes_per_launch": round(w[k][1] * 1024 / max(w[k][0], 1))} for k in f}}
json.dump(out, open("$O/${ROUND}_pmc_decode.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
