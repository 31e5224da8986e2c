# PMC passes for the persistent Cholesky kernel (separate runs per counter, no other tracing)
export TMPDIR=/tmp
TAG=${1:-r1}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_chol_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python $GRAFT_REPO_ROOT/tools/chol_trace.py run 6000 > $OUT/fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python $GRAFT_REPO_ROOT/tools/chol_trace.py run 6000 > $OUT/write.log 2>&1
python - <<PY
import csv, glob, json
out = {}
for name in ("fetch", "write"):
    fs = glob.glob("$OUT/%s/**/*counter_collection.csv" % name, recursive=True)
    vals = []
    for f in fs:
        for r in csv.DictReader(open(f)):
            if "chol_mega_kernel" in r.get("Kernel_Name", ""):
                vals.append(float(r["Counter_Value"]))
    out[name] = {"n": len(vals), "mean_counter_value_KB": (sum(vals) / len(vals)) if vals else None, "values": vals}
print(json.dumps(out))
json.dump(out, open("$OUT/summary.json", "w"))
PY
find $OUT -name "*.csv" -size +5M -delete
