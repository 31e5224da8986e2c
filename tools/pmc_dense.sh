# MFMA-utilisation / L2 / traffic counters of the dense form's product kernel (yyt_tile_kernel): separate PMC-only passes, no tracing.
# usage: bash tools/pmc_dense.sh <tag> [cams pts]  -> gpurun_out/<tag>_pmc_dense.json
export TMPDIR=/tmp
TAG=${1:-tmp}; CAMS=${2:-1000}; PTS=${3:-20000}
R=$GRAFT_REPO_ROOT
OUT=/tmp/pmc_dense_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp
HEAD=$(python -c "import importlib,sys; sys.path.insert(0,'$R'); print(importlib.import_module('slam-tricks_amd.build').build_head())")
i=0
for SET in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 900 rocprofv3 --pmc $SET --output-format csv -d $OUT/p$i -- python $R/tools/dense_schur_time.py $CAMS $PTS > $OUT/p$i.log 2>&1
done
python - <<PY
import csv, glob, json, collections
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "yyt_tile_kernel" in r.get("Kernel_Name", ""):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
d = {k: sum(v) / len(v) for k, v in acc.items()}
d["launches_averaged"] = max([len(v) for v in acc.values()] or [0])
n = 6 * $CAMS; lda = (n + 1 + 127) // 128 * 128
if d.get("GRBM_GUI_ACTIVE"):
    d["kernel_shader_cycles"] = d["GRBM_GUI_ACTIVE"] / 8.0
    if "SQ_VALU_MFMA_BUSY_CYCLES" in d: d["mfma_utilisation"] = d["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * d["kernel_shader_cycles"])
if d.get("TCC_HIT_sum") is not None and d.get("TCC_MISS_sum") is not None and d["TCC_HIT_sum"] + d["TCC_MISS_sum"] > 0:
    d["l2_hit_rate"] = d["TCC_HIT_sum"] / (d["TCC_HIT_sum"] + d["TCC_MISS_sum"])
if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
    d["hbm_bytes_per_launch"] = 2 * d["FETCH_SIZE"] * 1024 + d["WRITE_SIZE"] * 1024      # gfx950: FETCH_SIZE doubled (MI355X_MICROARCH.md)
d["algorithmic_flop"] = float(n) * n * 3.0 * $PTS
d["y_bytes"] = 8.0 * lda * 3 * $PTS
res = {"tool": "rocprofv3 --pmc <set> (separate PMC-only passes, no tracing): tools/pmc_dense.sh", "head": "$HEAD", "kernel": "yyt_tile_kernel",
       "scene": "$CAMS cameras x $PTS landmarks, 59 % visibility", "counters": d}
print(json.dumps(res, indent=1))
json.dump(res, open("$R/gpurun_out/${TAG}_pmc_dense.json", "w"), indent=1)
PY
