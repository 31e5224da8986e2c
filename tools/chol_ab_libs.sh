# A/B of two builds of the library (tmp_libs/<name>.so), interleaved: bash tools/chol_ab_libs.sh n rounds nameA nameB ...
n=$1; rounds=$2; shift 2
for r in $(seq $rounds); do for v in "$@"; do cp tmp_libs/$v.so slam-tricks_amd/libstba.so; echo -n "$v "; python - <<PY
import importlib, sys
sys.path.insert(0, ".")
st = importlib.import_module("slam-tricks_amd")
st.cholesky_time_split($n, reps=3)
print("%.4f %.4f" % st.cholesky_time_split($n, reps=20))
PY
done; done
