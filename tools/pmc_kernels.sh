# SQ / LDS counters of the assembly kernels of one LM iteration (separate PMC-only passes, no tracing):
# usage: bash tools/pmc_kernels.sh <tag> ; writes gpurun_out/pmc_kernels_<tag>.json
export TMPDIR=/tmp
TAG=${1:-r2}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_kernels_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp
python $GRAFT_REPO_ROOT/tools/lm_only.py 1 > /dev/null 2>&1   # builds the scene cache
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/sq -- python $GRAFT_REPO_ROOT/tools/lm_only.py 3 > $OUT/sq.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python $GRAFT_REPO_ROOT/tools/lm_only.py 3 > $OUT/fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python $GRAFT_REPO_ROOT/tools/lm_only.py 3 > $OUT/write.log 2>&1
python - <<PY
import csv, glob, json, collections
out = collections.defaultdict(lambda: collections.defaultdict(list))
for name in ("sq", "fetch", "write"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % name, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r.get("Kernel_Name", "").split("(")[0]
            if k.startswith("stba::ba_") or "schur" in k:
                out[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {k: {c: sum(v) / len(v) for c, v in d.items()} | {"launches": max(len(v) for v in d.values())} for k, d in out.items()}
print(json.dumps(res, indent=1))
json.dump(res, open("$GRAFT_REPO_ROOT/gpurun_out/pmc_kernels_$TAG.json", "w"), indent=1)
PY
tail -2 $OUT/sq.log
rm -rf $OUT
