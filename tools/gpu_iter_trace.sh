# usage: bash tools/gpu_iter_trace.sh ; kernel timeline of one LM iteration -> stdout
export TMPDIR=/tmp
OUT=/tmp/iter_trace
rm -rf $OUT; mkdir -p $OUT
cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python $GRAFT_REPO_ROOT/tools/iter_trace.py run > $OUT/run.log 2>&1
tail -1 $OUT/run.log
python $GRAFT_REPO_ROOT/tools/iter_trace.py $OUT
python - <<'PY'
import csv, glob
f = glob.glob("/tmp/iter_trace/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
lin = [i for i, r in enumerate(rows) if "ba_linearize_kernel<true, true>" in r["Kernel_Name"]]
a, b = lin[-3], lin[-2]
t0 = int(rows[a]["Start_Timestamp"])
n_b = 0
for r in rows[a:b]:
    name = r["Kernel_Name"].split("(")[0].replace("stba::", "").replace("void ", "")
    if "chol_bwd" in name:
        n_b += 1
        if n_b > 2: continue
    print(f'{name[:44]:44s} start={(int(r["Start_Timestamp"])-t0)/1e3:8.1f} dur={(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3:8.1f}')
PY
