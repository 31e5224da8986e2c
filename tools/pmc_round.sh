# Regenerates the PMC evidence of a round from the library at hand, stamped with its build head (slam-tricks_amd/BUILD_HEAD:
# git head + content hash of the sources), so that every `traffic` figure of a bench line can be held against the build it
# belongs to.  Separate PMC-only passes (no tracing), as MI355X_MICROARCH.md prescribes.
# usage: bash tools/pmc_round.sh <tag> [chol sizes...]   ->  gpurun_out/<tag>_pmc_jacobian.json, <tag>_pmc_assembly_kernels.json,
#                                                            <tag>_pmc_chol_mfma.json   (copy them to profiles/)
export TMPDIR=/tmp
TAG=${1:-r4}; shift
SIZES=${@:-6000}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
HEAD=$(python -c "import importlib,sys; sys.path.insert(0,'$R'); print(importlib.import_module('slam-tricks_amd.build').build_head())")
W=/tmp/pmc_round_$TAG; rm -rf $W; mkdir -p $W
cd /tmp
python $R/tools/jac_only.py 2 > /dev/null 2>&1   # builds the scene cache
# ---- residual + Jacobian kernel
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $W/jf -- python $R/tools/jac_only.py 5 > $W/jf.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $W/jw -- python $R/tools/jac_only.py 5 > $W/jw.log 2>&1
# ---- assembly kernels of an LM iteration
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $W/ks -- python $R/tools/lm_only.py 3 > $W/ks.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $W/kf -- python $R/tools/lm_only.py 3 > $W/kf.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $W/kw -- python $R/tools/lm_only.py 3 > $W/kw.log 2>&1
python - <<PY
import csv, glob, json, collections
head = "$HEAD"
def vals(d, pred):
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r.get("Kernel_Name", "")
            if pred(k): out[k.split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return out
# Jacobian kernel: gfx950 FETCH_SIZE counts wide coalesced reads at half their bytes (MI355X_MICROARCH.md, HBM): doubled
jf = vals("$W/jf", lambda k: "ba_linearize_kernel<true, true>" in k); jw = vals("$W/jw", lambda k: "ba_linearize_kernel<true, true>" in k)
f = [v for d in jf.values() for v in d.get("FETCH_SIZE", [])]; w = [v for d in jw.values() for v in d.get("WRITE_SIZE", [])]
if f and w:
    fk, wk = sum(f) / len(f), sum(w) / len(w)
    json.dump({"kernel": "ba_linearize_kernel<true, true> (C5: 1 000 000 observations per launch)", "head": head,
               "tool": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, no other tracing), tools/pmc_round.sh",
               "FETCH_SIZE_KB_mean": fk, "WRITE_SIZE_KB_mean": wk, "launches_averaged": min(len(f), len(w)),
               "correction": "gfx950: FETCH_SIZE doubled (128-B requests tallied at 64 B); WRITE_SIZE as is",
               "hbm_bytes_per_launch": 2 * fk * 1024 + wk * 1024, "algorithmic_bytes_per_launch": 106.5e6},
              open("$O/${TAG}_pmc_jacobian.json", "w"), indent=1)
res = {}
for d in ("$W/ks", "$W/kf", "$W/kw"):
    for k, cs in vals(d, lambda k: k.startswith("stba::ba_") or "schur" in k or "trial_finish" in k or "linear_finish" in k).items():
        res.setdefault(k, {}).update({c: sum(v) / len(v) for c, v in cs.items()})
        res[k]["launches"] = max(res[k].get("launches", 0), max(len(v) for v in cs.values()))
for k, d in res.items():
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        d["hbm_bytes_per_launch_fetch_doubled"] = 2 * d["FETCH_SIZE"] * 1024 + d["WRITE_SIZE"] * 1024
        d["hbm_bytes_per_launch_raw"] = d["FETCH_SIZE"] * 1024 + d["WRITE_SIZE"] * 1024
json.dump({"head": head, "tool": "rocprofv3 --pmc (SQ/LDS set | FETCH_SIZE | WRITE_SIZE: three separate passes over tools/lm_only.py 3, C5)",
           "note": "FETCH_SIZE on gfx950 tallies 128-B requests of wide coalesced reads at 64 B (doubled in *_fetch_doubled); gathers of 64-B records "
                   "are narrower requests and the raw figure is the better one for the Schur kernel (round 3 quoted raw)", "kernels": res},
          open("$O/${TAG}_pmc_assembly_kernels.json", "w"), indent=1)
print("jacobian", json.load(open("$O/${TAG}_pmc_jacobian.json"))["hbm_bytes_per_launch"] if f and w else None)
for k, d in res.items():
    if "schur" in k: print(k, {c: round(v / 1e3, 1) if c in ("FETCH_SIZE", "WRITE_SIZE") else v for c, v in d.items()})
PY
bash $R/tools/pmc_chol_mfma.sh $TAG $SIZES > $W/chol.log 2>&1
cp $O/pmc_chol_mfma_$TAG.json $O/${TAG}_pmc_chol_mfma.json 2>/dev/null
tail -3 $W/chol.log
rm -rf $W
