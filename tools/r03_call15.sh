#!/bin/bash
# GPU call 15: stream priorities / sub-batch size of the complete --O0 pipeline, then (1024,1536)
mkdir -p gpurun_out
for p in "-1,0" "0,-1" "0,0"; do
  echo "prio $p"; ZKWG_BENCH_PRIO=$p timeout 600 python tools/bench_full.py 2>/dev/null | grep -o '"complete O0".*' | cut -c1-120
done | tee gpurun_out/r03_o_prio.txt
for q in 512 2048; do
  echo "prep $q"; ZKWG_BENCH_PREP=$q timeout 600 python tools/bench_full.py 2>/dev/null | grep -o '"complete O0".*' | cut -c1-120
done | tee -a gpurun_out/r03_o_prio.txt
timeout 900 python tools/bench_full.py 1024 1536 2>/dev/null | tee gpurun_out/r03_o_full_1024.json | grep -o '"complete O0".*'
