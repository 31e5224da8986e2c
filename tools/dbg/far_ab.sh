export TMPDIR=/tmp
for r in 1 2 3 4 5; do for a in 0 6 8 12 16; do echo -n "FAR $a "; STBA_MEGA_FAR=$a STBA_LIB=tmp_libs/dbg.so python - <<PY 2>/dev/null | tail -1
import importlib, sys
sys.path.insert(0, ".")
st = importlib.import_module("slam-tricks_amd")
st.cholesky_time_split(6000, reps=3)
print("%.4f %.4f" % st.cholesky_time_split(6000, reps=25))
PY
done; done
