#!/bin/bash
# tuning sweep of zk_expand geometry: slots per WG x threads per WG x emails per WG
for cfg in "1024 256 1" "256 256 8" "256 256 16" "256 512 8" "256 512 16" "512 256 8" "512 512 8" "128 256 16" "256 256 32" "1024 256 4"; do
  set -- $cfg
  echo -n "portion=$1 threads=$2 emails_per_wg=$3: "
  ZKWG_PORTION=$1 ZKWG_EXPAND_THREADS=$2 ZKWG_EMAILS_PER_WG=$3 python bench.py --cpu-sample 0 --steps 3 --distinct 64 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['roofline']['achieved'], d['roofline']['avg_launch_ms'])"
done
