#!/bin/bash
# round 3, GPU call 5: isolation experiments for zk_expand (no overlap, CU-masked prepare), O0 one-pass v2 (chains, E emails/WG), host expansion
OUT=$PWD/gpurun_out; mkdir -p $OUT
B="--steps 10 --warmup 3 --other-configs 0 --pmc-traffic 0 --cpu-sample 0"
timeout 900 python -m pytest tests/test_full_witness.py tests/test_intake.py tests/test_host_expand.py tests/test_ev_gpu.py tests/test_multi.py \
  tests/test_configs_gpu.py::test_fused_montgomery_expand_equals_expand_then_convert -m gpu -x -q > $OUT/r03_e_tests.log 2>&1
echo "tests rc=$?" >> $OUT/r03_e_tests.log
tail -8 $OUT/r03_e_tests.log
timeout 300 python bench.py $B 2>/dev/null | tail -1 > $OUT/r03_e_base_k4.json
for k in 2 4; do
  ZKWG_X3_K=$k timeout 300 python bench.py $B --no-overlap 1 2>/dev/null | tail -1 > $OUT/r03_e_serial_k$k.json
done
ZKWG_EXPAND_V=1 timeout 300 python bench.py $B --no-overlap 1 2>/dev/null | tail -1 > $OUT/r03_e_serial_v1.json
for cu in 16 32 64; do
  timeout 300 python bench.py $B --prep-cus $cu --rsa-throttle 0 2>/dev/null | tail -1 > $OUT/r03_e_cus${cu}.json
done
timeout 300 python bench.py $B --prep-cus 32 --prep-cu-stride 8 --rsa-throttle 0 2>/dev/null | tail -1 > $OUT/r03_e_cus32_stride8.json
timeout 300 python bench.py $B --rsa-throttle 0 2>/dev/null | tail -1 > $OUT/r03_e_throttle0.json
for e in 1 4 8; do
  ZKWG_O0_EMAILS_PER_WG=$e timeout 600 python tools/bench_full.py > $OUT/r03_e_full_o0_576_e$e.json 2>> $OUT/r03_e_full_o0.err
done
ZKWG_X3_K_O0=2 timeout 600 python tools/bench_full.py > $OUT/r03_e_full_o0_576_k2.json 2>> $OUT/r03_e_full_o0.err
timeout 900 python tools/bench_full.py 1024 1536 > $OUT/r03_e_full_o0_1024.json 2>> $OUT/r03_e_full_o0.err
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r03_e_prof_o0 -- python $OLDPWD/tools/bench_full.py > /dev/null 2> $OUT/r03_e_prof_o0.log )
find $OUT/r03_e_prof_o0 -name "*kernel_stats.csv" | head -1 | xargs -r -I{} cp {} $OUT/r03_e_o0_kernel_stats.csv
rm -rf $OUT/r03_e_prof_o0
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/r03_e_*.json")):
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
head -8 $OUT/r03_e_o0_kernel_stats.csv | cut -c1-150
tail -3 $OUT/r03_e_full_o0.err
