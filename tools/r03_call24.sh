#!/bin/bash
# GPU call 24: zk_net_eval lanes per email (ZKWG_NET_LANES: 16 = 4 emails per wavefront, 32 = 2, 64 = 1) on the stand-in and
# the real-size regex templates, same box
T1=zk-email-verify_amd/data/templates/zk-regex-circom/circuits/common/body_hash_regex.circom
T2=tests/golden/regex_style/body_hash_regex_unshared.circom
for t in $T1 $T2; do for l in 16 32 64; do
  echo "template $(basename $t) lanes $l"
  ZKWG_NET_LANES=$l timeout 300 python bench.py --regex $t --steps 8 --warmup 2 --cpu-sample 0 --pmc-traffic 0 --other-configs 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['value'], d['roofline']['frac'], d['kernel_ms_per_launch'])"
done; done | tee gpurun_out/r03_u_net_lanes.txt
