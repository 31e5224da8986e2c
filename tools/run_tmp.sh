export TMPDIR=/tmp
for rep in 1 2 3; do
for wv in 0 1; do
echo "WIDE $wv"
STBA_MEGA_WIDE=$wv timeout 300 python tools/mega_trace.py run 6000
done
done
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | grep -E "passed|failed"
timeout 300 python tools/mega_stress.py 3000 10 1 2>&1 | tail -1
STBA_MEGA_TRACE=/tmp/mega.bin timeout 300 python tools/mega_trace.py run 6000
timeout 100 python tools/mega_trace.py /tmp/mega.bin > gpurun_out/mega_trace_d3.txt
python - <<'PY'
import numpy as np
raw = open('/tmp/mega.bin','rb').read()
nt = int(np.frombuffer(raw[:4], np.int32)[0])
tasks = np.frombuffer(raw[4:4+16*nt], np.int32).reshape(nt,4).copy()
tr = np.frombuffer(raw[4+16*nt:4+16*nt+64*nt], np.int64).reshape(nt,8)
ty = tasks[:,0] & 0xff; nb = np.maximum(1,(tasks[:,0]>>16)&0xff); wide = (tasks[:,0]>>25)&1
run = (tr[:,3]-tr[:,2])/100.0
for n_ in (1,2):
    for w_ in (0,1):
        m = (ty==3)&(nb==n_)&(wide==w_)&(tasks[:,2]<47)
        if m.any(): print("U nb",n_,"wide",w_,"n",m.sum(),"run mean",run[m].mean().round(2),"med",np.median(run[m]).round(2))
PY
