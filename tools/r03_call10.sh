#!/bin/bash
# round 3, GPU call 10: O0 row kernels moved into prepare (overlap with the previous sub-batch's expansion)
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest tests/test_full_witness.py tests/test_intake.py -m gpu -q > $OUT/r03_j_tests.log 2>&1
echo "tests rc=$?" >> $OUT/r03_j_tests.log
tail -3 $OUT/r03_j_tests.log
timeout 600 python tools/bench_full.py > $OUT/r03_j_full_o0_576.json 2>> $OUT/r03_j_full_o0.err
timeout 900 python tools/bench_full.py 1024 1536 > $OUT/r03_j_full_o0_1024.json 2>> $OUT/r03_j_full_o0.err
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/r03_j_*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1], {k: (v["witnesses_per_s"], v["GBps_written"], v["create_s"], v["kernel_ms"]) for k, v in d.items()})
PY
