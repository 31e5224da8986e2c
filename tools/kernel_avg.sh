# average device time of the LM iteration's kernels whose name matches $1 (rocprofv3 --kernel-trace --stats of tools/lm_phases.py): bash tools/kernel_avg.sh <regex>
export TMPDIR=/tmp
P=/tmp/prof_kavg; rm -rf $P; mkdir -p $P
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $P -- python $GRAFT_REPO_ROOT/tools/lm_phases.py 20 > /dev/null 2>&1)
python - "$1" <<'PY'
import csv, glob, re, sys
f = glob.glob("/tmp/prof_kavg/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if re.search(sys.argv[1], r["Name"]): print(f'{r["Name"][:70]:70s} {r["Calls"]:>5s} {float(r["AverageNs"]) / 1e3:8.2f} us')
PY
