#!/bin/bash
# tuning sweep of zk_expand geometry (slots per WG x threads per WG)
for cfg in "1024 256" "2048 256" "4096 256" "8192 256" "4096 512" "16384 256"; do
  set -- $cfg
  echo -n "portion=$1 threads=$2: "
  ZKWG_PORTION=$1 ZKWG_EXPAND_THREADS=$2 python bench.py --cpu-sample 0 --steps 3 --distinct 64 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['roofline']['achieved'], d['roofline']['avg_launch_ms'])"
done
