# builds a SECOND copy of the library with the scheduling-experiment knobs compiled in (environment variables read by
# dense_chol.hip) -> tmp_libs/<name>.so; use it with STBA_LIB=tmp_libs/<name>.so.  The product library reads no environment.
# usage: bash tools/build_dbg.sh [name] [extra hipcc flags...]
NAME=${1:-dbg}; shift
R=$(cd $(dirname $0)/.. && pwd)
mkdir -p $R/tmp_libs/obj_$NAME
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -munsafe-fp-atomics -Wno-unused-function -Wno-unused-result -DSTBA_DEBUG_KNOBS $@"
pids=""
for s in dense_chol.hip ba_kernels.hip stba_engine.hip pg_engine.hip small_dense.hip two_view.hip calib_io.cpp comm.cpp; do
  o=$R/tmp_libs/obj_$NAME/${s%.*}.o
  /opt/rocm/bin/hipcc $FLAGS -c $R/slam-tricks_amd/csrc/$s -o $o & pids="$pids $!"
done
for p in $pids; do wait $p || exit 1; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tmp_libs/$NAME.so $R/tmp_libs/obj_$NAME/*.o -ldl && echo built $R/tmp_libs/$NAME.so
