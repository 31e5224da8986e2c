export TMPDIR=/tmp
cd /tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "TCC_[A-Z_a-z0-9]*" | sort -u | tr '\n' ' ' | cut -c1-1500
echo
OUT=/tmp/tcc; rm -rf $OUT; mkdir -p $OUT
timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum --output-format csv -d $OUT -- python $GRAFT_REPO_ROOT/tools/chol_trace.py run 6000 > $OUT/log.txt 2>&1
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("/tmp/tcc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "chol_mega_kernel" in r.get("Kernel_Name",""):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items(): print(k, sum(v)/len(v), len(v))
PY
tail -2 $OUT/log.txt
