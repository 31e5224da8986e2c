export TMPDIR=/tmp
timeout 60 ./tools/exp/diag2_test.bin > gpurun_out/diag2_test.txt
timeout 60 ./tools/exp/diag2_test_nots.bin | tee gpurun_out/diag2_test_nots.txt
