#!/bin/bash
# round 3, GPU call 1: parity of the LDS-staged zk_expand2 + same-box A/B against the round-2 kernel
OUT=$PWD/gpurun_out; mkdir -p $OUT
B="--steps 10 --warmup 3 --other-configs 0 --pmc-traffic 0 --cpu-sample 0"
timeout 1200 python -m pytest tests/test_ev_gpu.py tests/test_sha_gpu.py tests/test_rsa_gpu.py tests/test_masks.py \
  tests/test_configs_gpu.py::test_config1_batch256_bit_exact tests/test_configs_gpu.py::test_fused_montgomery_expand_equals_expand_then_convert \
  tests/test_configs_gpu.py::test_prover_handoff_montgomery_round_trip tests/test_multi.py tests/test_fuzz_gpu.py \
  tests/test_circom_fixture_gpu.py tests/test_soft_line_breaks.py tests/test_regex_template.py -m gpu -x -q > $OUT/r03_a_tests.log 2>&1
echo "tests rc=$?" >> $OUT/r03_a_tests.log
tail -5 $OUT/r03_a_tests.log
for v in 1 2 1 2; do
  ZKWG_EXPAND_V=$v timeout 300 python bench.py $B 2>/dev/null | tail -1 > $OUT/r03_a_ab_v${v}_$RANDOM.json
done
for p in 512 1024 4096; do
  ZKWG_PORTION=$p timeout 300 python bench.py $B 2>/dev/null | tail -1 > $OUT/r03_a_portion_$p.json
done
for v in 1 2; do
  ZKWG_EXPAND_V=$v timeout 300 python bench.py $B --montgomery 1 --batch 2048 --steps 5 2>/dev/null | tail -1 > $OUT/r03_a_mont_v$v.json
done
ZKWG_EMAILS_PER_WG=4 timeout 300 python bench.py $B 2>/dev/null | tail -1 > $OUT/r03_a_epw4.json
timeout 120 ./tools/fillbench > $OUT/r03_a_fillbench.txt 2>&1
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/r03_a_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d["roofline"]
        print(f.split("/")[-1], d["value"], r["avg_launch_ms"], r["achieved"], r["frac"], r.get("box_fill_GBps"), d["kernel_ms_per_launch"])
    except Exception as e:
        print(f, "ERR", e)
PY
cat $OUT/r03_a_fillbench.txt | head -70
