#!/bin/bash
# round 3, GPU call 3: full GPU suite with zk_expand3 (default) + same-box A/B v1/v2/v3 + Montgomery + O0
OUT=$PWD/gpurun_out; mkdir -p $OUT
B="--steps 10 --warmup 3 --other-configs 0 --pmc-traffic 0 --cpu-sample 0"
for v in 3 1 3 2; do
  ZKWG_EXPAND_V=$v timeout 300 python bench.py $B 2>/dev/null | tail -1 > $OUT/r03_c_ab_v${v}_$RANDOM.json
done
for v in 1 3; do
  ZKWG_EXPAND_V=$v timeout 300 python bench.py $B --montgomery 1 --batch 2048 --steps 5 2>/dev/null | tail -1 > $OUT/r03_c_mont_v$v.json
done
ZKWG_XCD_REMAP=0 timeout 300 python bench.py $B 2>/dev/null | tail -1 > $OUT/r03_c_noxcd_v3.json
timeout 600 python tools/bench_full.py > $OUT/r03_c_full_o0_576.json 2> $OUT/r03_c_full_o0_576.err
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/r03_c_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        if "roofline" in d:
            r = d["roofline"]
            print(f.split("/")[-1], d["value"], r["avg_launch_ms"], r["achieved"], r["frac"], r.get("box_fill_GBps"), d["kernel_ms_per_launch"])
        else:
            print(f.split("/")[-1], d)
    except Exception as e:
        print(f, "ERR", e)
PY
tail -3 $OUT/r03_c_full_o0_576.err
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/r03_c_tests.log 2>&1
echo "tests rc=$?" >> $OUT/r03_c_tests.log
tail -15 $OUT/r03_c_tests.log
