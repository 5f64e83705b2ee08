#!/bin/bash
# GPU call 11: GENERIC wires decoded by a pre-pass (zk_o0_generic); the streaming kernel at K = 4 / 2 / 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_full_witness.py -m gpu -x -q > gpurun_out/r03_k_tests.txt 2>&1; tail -3 gpurun_out/r03_k_tests.txt
for k in 4 2; do
  ZKWG_X3_K_O0=$k timeout 600 python tools/bench_full.py > gpurun_out/r03_k_full_k$k.txt 2>&1; tail -4 gpurun_out/r03_k_full_k$k.txt
done
ZKWG_X3_K_O0=2 ZKWG_O0_EMAILS_PER_WG=4 timeout 600 python tools/bench_full.py > gpurun_out/r03_k_full_k2e4.txt 2>&1; tail -4 gpurun_out/r03_k_full_k2e4.txt
