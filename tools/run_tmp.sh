export TMPDIR=/tmp
for rep in 1 2 3; do
for f in 0 1; do
echo "FUSET $f"
STBA_MEGA_FUSET=$f timeout 300 python tools/mega_trace.py run 6000
done
done
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | grep -E "passed|failed"
timeout 300 python tools/mega_stress.py 3000 10 1 2>&1 | tail -1
timeout 300 python tools/mega_stress.py 1500 20 2 2>&1 | tail -1
STBA_MEGA_TRACE=/tmp/mega.bin timeout 300 python tools/mega_trace.py run 6000
timeout 100 python tools/mega_trace.py /tmp/mega.bin > gpurun_out/mega_trace_d3.txt
