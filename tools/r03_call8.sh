#!/bin/bash
# round 3, GPU call 8: O0 with hoisted loads; serial (non-overlapped) pipeline variants
OUT=$PWD/gpurun_out; mkdir -p $OUT
B="--steps 10 --warmup 3 --other-configs 0 --pmc-traffic 0 --cpu-sample 0"
timeout 600 python -m pytest tests/test_full_witness.py tests/test_intake.py -m gpu -q > $OUT/r03_h_tests.log 2>&1
echo "tests rc=$?" >> $OUT/r03_h_tests.log
tail -4 $OUT/r03_h_tests.log
for k in 2 4; do for e in 4 8; do
  ZKWG_X3_K_O0=$k ZKWG_O0_EMAILS_PER_WG=$e timeout 600 python tools/bench_full.py > $OUT/r03_h_full_o0_576_k${k}e$e.json 2>> $OUT/r03_h_full_o0.err
done; done
ZKWG_X3_K_O0=2 ZKWG_O0_EMAILS_PER_WG=8 timeout 900 python tools/bench_full.py 1024 1536 > $OUT/r03_h_full_o0_1024_k2e8.json 2>> $OUT/r03_h_full_o0.err
ZKWG_X3_K_O0=4 ZKWG_O0_EMAILS_PER_WG=8 timeout 900 python tools/bench_full.py 1024 1536 > $OUT/r03_h_full_o0_1024_k4e8.json 2>> $OUT/r03_h_full_o0.err
timeout 300 python bench.py $B 2>/dev/null | tail -1 > $OUT/r03_h_overlap.json
timeout 300 python bench.py $B --no-overlap 1 --prep-batch 4096 2>/dev/null | tail -1 > $OUT/r03_h_serial_p4096.json
timeout 300 python bench.py $B --no-overlap 1 --prep-batch 2048 2>/dev/null | tail -1 > $OUT/r03_h_serial_p2048.json
timeout 300 python bench.py $B --prep-batch 2048 2>/dev/null | tail -1 > $OUT/r03_h_overlap_p2048.json
timeout 300 python bench.py $B --prep-batch 512 --ring 3 2>/dev/null | tail -1 > $OUT/r03_h_overlap_p512.json
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/r03_h_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        if "roofline" in d:
            r = d["roofline"]
            print(f.split("/")[-1], d["value"], r["avg_launch_ms"], r["achieved"], r["frac"], r.get("box_fill_GBps"), d["kernel_ms_per_launch"])
        else:
            print(f.split("/")[-1], {k: (v["witnesses_per_s"], v["GBps_written"], v["create_s"], v["kernel_ms"]["zk_expand"]) for k, v in d.items()})
    except Exception as e:
        print(f, "ERR", e)
PY
tail -3 $OUT/r03_h_full_o0.err
