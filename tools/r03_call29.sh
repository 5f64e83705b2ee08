#!/bin/bash
# GPU call 29: emails per workgroup of zk_expand3_o0 (descriptor loads amortised over E emails), same box
for e in 8 16 4; do
  echo "ZKWG_O0_EMAILS_PER_WG=$e"
  ZKWG_O0_EMAILS_PER_WG=$e timeout 600 python tools/bench_full.py 2>/dev/null | tail -1 | grep -o '"complete O0".*' | cut -c1-330
  ZKWG_O0_EMAILS_PER_WG=$e timeout 600 python tools/bench_abc.py 2>/dev/null | tail -1 | cut -c1-420
done | tee gpurun_out/r03_z_epw.txt
