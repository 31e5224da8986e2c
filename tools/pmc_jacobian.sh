# PMC passes for the residual+Jacobian kernel (separate runs per counter group, no other tracing):
# writes gpurun_out/pmc_<tag>/{fetch,write}/... and a small JSON summary
export TMPDIR=/tmp
TAG=${1:-r1}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp
python $GRAFT_REPO_ROOT/tools/jac_only.py 2 > /dev/null 2>&1   # builds the scene cache
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python $GRAFT_REPO_ROOT/tools/jac_only.py 5 > $OUT/fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python $GRAFT_REPO_ROOT/tools/jac_only.py 5 > $OUT/write.log 2>&1
python - <<PY
import csv, glob, json
out = {}
for name in ("fetch", "write"):
    fs = glob.glob("$OUT/%s/**/*counter_collection.csv" % name, recursive=True)
    vals = []
    for f in fs:
        for r in csv.DictReader(open(f)):
            if "ba_linearize_kernel<true, true>" in r.get("Kernel_Name", ""):
                vals.append(float(r["Counter_Value"]))
    out[name] = {"n": len(vals), "mean_counter_value": (sum(vals) / len(vals)) if vals else None}
print(json.dumps(out))
json.dump(out, open("$OUT/summary.json", "w"))
PY
find $OUT -name "*.csv" -size +5M -delete
