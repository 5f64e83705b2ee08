#!/bin/bash
# round 3, GPU call 4: zk_expand3 with K slots per thread: sweep K x XCD remap, Montgomery, O0 (+ kernel breakdown), poseidon variants
OUT=$PWD/gpurun_out; mkdir -p $OUT
B="--steps 10 --warmup 3 --other-configs 0 --pmc-traffic 0 --cpu-sample 0"
timeout 900 python -m pytest tests/test_ev_gpu.py tests/test_sha_gpu.py tests/test_rsa_gpu.py tests/test_masks.py \
  tests/test_configs_gpu.py::test_config1_batch256_bit_exact tests/test_configs_gpu.py::test_fused_montgomery_expand_equals_expand_then_convert \
  tests/test_full_witness.py tests/test_circom_fixture_gpu.py tests/test_regex_template.py -m gpu -x -q > $OUT/r03_d_tests.log 2>&1
echo "tests rc=$?" >> $OUT/r03_d_tests.log
tail -5 $OUT/r03_d_tests.log
for k in 1 2 4; do for x in 1 0; do
  ZKWG_X3_K=$k ZKWG_XCD_REMAP=$x timeout 300 python bench.py $B 2>/dev/null | tail -1 > $OUT/r03_d_k${k}_xcd${x}.json
done; done
ZKWG_EXPAND_V=2 timeout 300 python bench.py $B 2>/dev/null | tail -1 > $OUT/r03_d_v2.json
ZKWG_POS_LANE=1 timeout 300 python bench.py $B 2>/dev/null | tail -1 > $OUT/r03_d_k2_poslane.json
for k in 2 4; do
  ZKWG_X3_K=$k timeout 300 python bench.py $B --montgomery 1 --batch 2048 --steps 5 2>/dev/null | tail -1 > $OUT/r03_d_mont_k$k.json
done
for k in 1 2 4; do
  ZKWG_X3_K_O0=$k timeout 600 python tools/bench_full.py > $OUT/r03_d_full_o0_576_k$k.json 2> $OUT/r03_d_full_o0.err
done
( cd /tmp && export TMPDIR=/tmp && ZKWG_X3_K_O0=2 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r03_d_prof_o0 -- python $OLDPWD/tools/bench_full.py > /dev/null 2> $OUT/r03_d_prof_o0.log )
find $OUT/r03_d_prof_o0 -name "*kernel_stats.csv" | head -1 | xargs -r -I{} cp {} $OUT/r03_d_o0_kernel_stats.csv
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/r03_d_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        if "roofline" in d:
            r = d["roofline"]
            print(f.split("/")[-1], d["value"], r["avg_launch_ms"], r["achieved"], r["frac"], r.get("box_fill_GBps"), d["kernel_ms_per_launch"])
        else:
            print(f.split("/")[-1], {k: (v["witnesses_per_s"], v["GBps_written"], v["kernel_ms"]["zk_expand"]) for k, v in d.items()})
    except Exception as e:
        print(f, "ERR", e)
PY
head -12 $OUT/r03_d_o0_kernel_stats.csv | cut -c1-150
