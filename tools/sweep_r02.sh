#!/bin/bash
# round-2 same-box sweeps: RSA throttle under the wave-parallel zk_rsa, inversion share, prep granularity
OUT=gpurun_out/r02d; mkdir -p $OUT
(timeout 400 python -m pytest tests/test_rsa_gpu.py tests/test_ev_gpu.py tests/test_fuzz_gpu.py -m gpu -x -q 2>&1 | tail -4) > $OUT/pytest.log; cat $OUT/pytest.log
B="python bench.py --pmc-traffic 0 --other-configs 0 --cpu-sample 0 --steps 5 --warmup 2"
for t in 0 4 8 16; do $B --rsa-throttle $t > $OUT/throttle_$t.json 2>/dev/null; done
ZKWG_DEBUG_SKIP_INV=1 $B --rsa-throttle 0 > $OUT/skipinv_t0.json 2>/dev/null
ZKWG_POS_SIDE=1 $B > $OUT/pos_side_t4.json 2>/dev/null
$B --prep-batch 2048 > $OUT/prep2048.json 2>/dev/null
$B --ring 3 > $OUT/ring3.json 2>/dev/null
$B --rsa-throttle 4 > $OUT/throttle_4_again.json 2>/dev/null
for f in $OUT/*.json; do python - "$f" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[1].split('/')[-1], d['value'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['kernel_ms_per_launch'])
PY
done
