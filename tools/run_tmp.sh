export TMPDIR=/tmp
run() { echo -n "$1 : "; env $1 timeout 300 python tools/mega_trace.py run 6000 2>/dev/null | grep "chol ms"; }
for rep in 1 2; do
run "X=0"
run "STBA_MEGA_QROWS=1"
run "STBA_MEGA_QROWS=3"
run "STBA_MEGA_QROWS=4"
run "STBA_MEGA_ROWMAP=2"
run "STBA_MEGA_ROWMAP=0"
run "STBA_MEGA_DUR=23,23,19,25,16.5,20"
run "STBA_MEGA_DUR=23,21,19,24,16,18"
run "STBA_MEGA_DUR=29,23,19,27,16.5,20"
run "STBA_MEGA_RES=2,2"
run "STBA_MEGA_BLEVEL=0.5"
run "STBA_MEGA_BLEVEL=2"
done
