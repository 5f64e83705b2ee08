#!/bin/bash
# round 3, GPU call 6: fixes (host expansion, O0 rows) + geometry sweep under the lighter prepare settings
OUT=$PWD/gpurun_out; mkdir -p $OUT
B="--steps 10 --warmup 3 --other-configs 0 --pmc-traffic 0 --cpu-sample 0 --rsa-throttle 0"
timeout 900 python -m pytest tests/test_host_expand.py tests/test_intake.py tests/test_full_witness.py tests/test_multi.py \
  tests/test_configs_gpu.py::test_fused_montgomery_expand_equals_expand_then_convert tests/test_ev_gpu.py -m gpu -q > $OUT/r03_f_tests.log 2>&1
echo "tests rc=$?" >> $OUT/r03_f_tests.log
tail -12 $OUT/r03_f_tests.log
for k in 4 8; do
  ZKWG_X3_K=$k timeout 300 python bench.py $B 2>/dev/null | tail -1 > $OUT/r03_f_k${k}.json
  ZKWG_X3_K=$k timeout 300 python bench.py $B --prep-cus 32 --prep-cu-stride 8 2>/dev/null | tail -1 > $OUT/r03_f_k${k}_cus32s8.json
done
timeout 300 python bench.py $B --prep-cus 24 --prep-cu-stride 8 2>/dev/null | tail -1 > $OUT/r03_f_k4_cus24s8.json
timeout 300 python bench.py $B --prep-cus 32 --prep-cu-stride 4 2>/dev/null | tail -1 > $OUT/r03_f_k4_cus32s4.json
timeout 300 python bench.py $B --prep-cus 48 --prep-cu-stride 4 2>/dev/null | tail -1 > $OUT/r03_f_k4_cus48s4.json
ZKWG_X3_K=8 timeout 300 python bench.py $B --no-overlap 1 2>/dev/null | tail -1 > $OUT/r03_f_serial_k8.json
timeout 300 python bench.py $B --montgomery 1 --batch 2048 --steps 5 --prep-cus 32 --prep-cu-stride 8 2>/dev/null | tail -1 > $OUT/r03_f_mont_k4_cus32s8.json
ZKWG_X3_K=8 timeout 300 python bench.py $B --montgomery 1 --batch 2048 --steps 5 2>/dev/null | tail -1 > $OUT/r03_f_mont_k8.json
for e in 4 8; do
  ZKWG_O0_EMAILS_PER_WG=$e timeout 600 python tools/bench_full.py > $OUT/r03_f_full_o0_576_e$e.json 2>> $OUT/r03_f_full_o0.err
done
ZKWG_X3_K_O0=2 ZKWG_O0_EMAILS_PER_WG=8 timeout 600 python tools/bench_full.py > $OUT/r03_f_full_o0_576_k2e8.json 2>> $OUT/r03_f_full_o0.err
ZKWG_O0_EMAILS_PER_WG=8 timeout 900 python tools/bench_full.py 1024 1536 > $OUT/r03_f_full_o0_1024.json 2>> $OUT/r03_f_full_o0.err
( cd /tmp && export TMPDIR=/tmp && ZKWG_O0_EMAILS_PER_WG=8 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r03_f_prof_o0 -- python $OLDPWD/tools/bench_full.py > /dev/null 2> $OUT/r03_f_prof_o0.log )
find $OUT/r03_f_prof_o0 -name "*kernel_stats.csv" | head -1 | xargs -r -I{} cp {} $OUT/r03_f_o0_kernel_stats.csv
rm -rf $OUT/r03_f_prof_o0
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/r03_f_*.json")):
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
head -9 $OUT/r03_f_o0_kernel_stats.csv | cut -c1-150
