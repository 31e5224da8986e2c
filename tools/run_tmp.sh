export TMPDIR=/tmp
for rep in 1 2 3; do
for cfg in "2 3 2" "2 3 1" "4 5 1" "4 5 2" "8 5 1" "4 8 1"; do
set -- $cfg
echo "BATCH $1 BLAG $2 PRE $3"
STBA_MEGA_PREDRAW=$3 STBA_MEGA_BATCH=$1 STBA_MEGA_BLAG=$2 timeout 300 python tools/mega_trace.py run 6000
done
done
