# MFMA-utilisation / L2 / stall counters of the persistent factorisation kernel (north_star: "rocprof ... MFMA utilisation").
# Separate PMC-only passes (no tracing), one per counter block, at n = 6000 and a large size.
# usage: bash tools/pmc_chol_mfma.sh <tag> [n ...]  -> gpurun_out/pmc_chol_mfma_<tag>.json  (copy to profiles/pmc_chol_mfma.json)
export TMPDIR=/tmp
TAG=${1:-r4}; shift
SIZES=${@:-6000 24000}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_chol_mfma_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp
HEAD=$(python -c "import importlib,sys; sys.path.insert(0,'$R'); print(importlib.import_module('slam-tricks_amd.build').build_head())")
for N in $SIZES; do
  i=0
  for SET in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
             "SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
             "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
             "TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE" \
             "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    timeout 900 rocprofv3 --pmc $SET --output-format csv -d $OUT/n${N}_p$i -- python $R/tools/chol_trace.py run $N > $OUT/n${N}_p$i.log 2>&1
  done
done
python - <<PY
import csv, glob, json, collections, re
res = {"tool": "rocprofv3 --pmc <set> (separate PMC-only passes, no tracing): tools/pmc_chol_mfma.sh", "head": "$HEAD", "sizes": {}}
for N in "$SIZES".split():
    acc = collections.defaultdict(list)
    for f in glob.glob("$OUT/n%s_p*/**/*counter_collection.csv" % N, recursive=True):
        for r in csv.DictReader(open(f)):
            if "chol_mega_kernel" in r.get("Kernel_Name", ""):
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    d = {k: sum(v) / len(v) for k, v in acc.items()}
    d["launches_averaged"] = max([len(v) for v in acc.values()] or [0])
    ms = None
    for f in glob.glob("$OUT/n%s_p1.log" % N):
        m = re.search(r"chol ms ([0-9.]+)", open(f).read())
        if m: ms = float(m.group(1))
    d["ms_per_factorisation_under_the_profiler"] = ms
    n = int(N); nblk = (n + 1 + 127) // 128
    # one v_mfma_f64_16x16x4_f64 = 2048 flop = 64 cycles of one SIMD's FP64 matrix pipe (profiles/mfma_f64_microbench.txt)
    # SQ_VALU_MFMA_BUSY_CYCLES: busy cycles of the matrix pipes summed over the 1024 SIMDs (measured: exactly 64 per
    # SQ_INSTS_MFMA).  GRBM_GUI_ACTIVE comes back summed over the 8 XCDs (43.7 M for a 2.34 ms kernel = 8 x 5.46 M cycles at
    # 2.33 GHz), so the kernel's shader cycles are GRBM_GUI_ACTIVE / 8 and
    #   MFMA utilisation = busy cycles / (1024 SIMDs x kernel cycles)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "GRBM_GUI_ACTIVE" in d and d["GRBM_GUI_ACTIVE"] > 0:
        d["kernel_shader_cycles"] = d["GRBM_GUI_ACTIVE"] / 8.0
        d["mfma_utilisation"] = d["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * d["kernel_shader_cycles"])
        if d.get("SQ_INSTS_MFMA"): d["mfma_busy_cycles_per_instruction"] = d["SQ_VALU_MFMA_BUSY_CYCLES"] / d["SQ_INSTS_MFMA"]
    if "SQ_VALU_MFMA_BUSY_CYCLES" in d and d.get("SQ_BUSY_CU_CYCLES"):
        d["mfma_busy_over_busy_cu_cycles"] = d["SQ_VALU_MFMA_BUSY_CYCLES"] / d["SQ_BUSY_CU_CYCLES"]
    if d.get("TCC_HIT_sum") is not None and d.get("TCC_MISS_sum") is not None and d["TCC_HIT_sum"] + d["TCC_MISS_sum"] > 0:
        d["l2_hit_rate"] = d["TCC_HIT_sum"] / (d["TCC_HIT_sum"] + d["TCC_MISS_sum"])
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        d["hbm_bytes_per_launch"] = 2 * d["FETCH_SIZE"] * 1024 + d["WRITE_SIZE"] * 1024   # gfx950: FETCH_SIZE doubled (MI355X_MICROARCH.md)
        d["algorithmic_bytes_per_launch"] = 8.0 * (128 * nblk) ** 2
    d["algorithmic_flop"] = n ** 3 / 3.0 + n ** 2 / 2.0
    res["sizes"][N] = d
print(json.dumps(res, indent=1))
json.dump(res, open("$R/gpurun_out/pmc_chol_mfma_$TAG.json", "w"), indent=1)
PY
for N in $SIZES; do tail -2 $OUT/n${N}_p1.log; done
find $OUT -name "*.csv" -size +2M -delete
