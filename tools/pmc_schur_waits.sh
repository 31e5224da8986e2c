# where the waves of the Schur kernel spend their cycles (PMC-only pass): usage: bash tools/pmc_schur_waits.sh
export TMPDIR=/tmp
OUT=/tmp/pmc_schur; rm -rf $OUT; mkdir -p $OUT
cd /tmp
python $GRAFT_REPO_ROOT/tools/lm_only.py 1 > /dev/null 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS --output-format csv -d $OUT/a -- python $GRAFT_REPO_ROOT/tools/lm_only.py 3 > $OUT/a.log 2>&1
python - <<PY
import csv, glob, collections
out = collections.defaultdict(list)
for f in glob.glob("/tmp/pmc_schur/a/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "schur" in r.get("Kernel_Name", ""): out[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in out.items(): print(k, round(sum(v) / len(v) / 1e6, 2), "M")
PY
tail -2 $OUT/a.log
