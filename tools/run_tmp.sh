export TMPDIR=/tmp
for kv in "STBA_MEGA_HI=3" "STBA_MEGA_FUSET=1" "STBA_MEGA_QFROM=24" "STBA_MEGA_BATCH=4 STBA_MEGA_BLAG=5 STBA_MEGA_PREDRAW=1" "STBA_MEGA_BATCH=1" "STBA_BWD_WIDE=0" "STBA_LM_SPECULATE=0"; do
echo "== $kv"
env $kv timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "cholesky or c5_full or st20_reference" 2>&1 | grep -E "passed|failed"
done
