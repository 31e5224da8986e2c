export TMPDIR=/tmp; timeout 60 ./tools/exp/diag2_test.bin | tee gpurun_out/diag2_test.txt
