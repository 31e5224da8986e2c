# usage: tools/run_trace.sh <tag> [ENV=VALUE ...]   -- trace of one factorisation (debug build), analysis into gpurun_out/
tag=$1; shift
export TMPDIR=/tmp
env "$@" STBA_MEGA_TRACE=/tmp/mega_$tag.bin python tools/mega_trace.py run 6000 > gpurun_out/mega_trace_$tag.txt 2>&1
python tools/mega_trace.py /tmp/mega_$tag.bin >> gpurun_out/mega_trace_$tag.txt 2>&1
