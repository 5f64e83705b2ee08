#!/bin/bash
# round 3, GPU call 9: zk_net_eval with 2 emails per wavefront (32 lanes per step) vs 64 lanes
OUT=$PWD/gpurun_out; mkdir -p $OUT
B="--steps 4 --warmup 2 --other-configs 0 --pmc-traffic 0 --cpu-sample 0"
T=zk-email-verify_amd/data/templates/zk-regex-circom/circuits/common/body_hash_regex.circom
U=tests/golden/regex_style/body_hash_regex_unshared.circom
timeout 900 python -m pytest tests/test_regex_template.py tests/test_full_witness.py -m gpu -q > $OUT/r03_i_tests.log 2>&1
echo "tests rc=$?" >> $OUT/r03_i_tests.log
tail -4 $OUT/r03_i_tests.log
for L in 32 64 16; do
  ZKWG_NET_LANES=$L timeout 300 python bench.py $B --regex $T 2>/dev/null | tail -1 > $OUT/r03_i_regex_L$L.json
done
ZKWG_NET_LANES=32 timeout 300 python bench.py $B --regex $T --prep-batch 4096 2>/dev/null | tail -1 > $OUT/r03_i_regex_L32_p4096.json
ZKWG_NET_LANES=32 timeout 300 python bench.py $B --regex $T --prep-batch 1024 2>/dev/null | tail -1 > $OUT/r03_i_regex_L32_p1024.json
for L in 32 64; do
  ZKWG_NET_LANES=$L timeout 300 python bench.py $B --regex $U 2>/dev/null | tail -1 > $OUT/r03_i_unshared_L$L.json
done
timeout 300 python bench.py $B 2>/dev/null | tail -1 > $OUT/r03_i_builtin.json
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/r03_i_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d["roofline"]
        print(f.split("/")[-1], d["value"], r["avg_launch_ms"], r["frac"], d["kernel_ms_per_launch"])
    except Exception as e:
        print(f, "ERR", e)
PY
