# usage: bash tools/exp/build_variant.sh <name> <extra -D flags...> ; builds tools/exp/libstba_<name>.so
set -e
cd "$(dirname "$0")/../.."
NAME=$1; shift
F="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -munsafe-fp-atomics -Wno-unused-function -Wno-unused-result"
/opt/rocm/bin/hipcc $F "$@" -c slam-tricks_amd/csrc/dense_chol.hip -o /tmp/dense_chol_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/exp/libstba_$NAME.so /tmp/dense_chol_$NAME.o slam-tricks_amd/csrc/ba_kernels.o slam-tricks_amd/csrc/stba_engine.o slam-tricks_amd/csrc/pg_engine.o
echo built tools/exp/libstba_$NAME.so
