#!/bin/bash
# GPU call 25: constant-one wire as an immediate (rows with a constant term become narrow): parity, then the O0 and A.w|B.w|C.w rates
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_full_witness.py tests/test_r1cs.py tests/test_fpmul.py -m gpu -x -q 2>&1 | tail -3
timeout 600 python tools/bench_abc.py 2>/dev/null | tail -1 | tee gpurun_out/r03_x_abc.json | cut -c1-600
timeout 600 python tools/bench_full.py 2>/dev/null | tail -1 | tee gpurun_out/r03_x_full_576.json | grep -o '"complete O0".*' | cut -c1-400
