#!/bin/bash
# GPU call 33/34: Montgomery kernels without scratch memory (zk_fr_half4 without a runtime index; zk_mont_slow takes the reference sources by value): parity of the
# Montgomery paths, then the headline pipeline with --montgomery 1 and the prover-stage rates
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_configs_gpu.py tests/test_r1cs.py tests/test_full_witness.py -m gpu -x -q -k "montgomery or compact_image or complete or full" 2>&1 | tail -3
timeout 300 python bench.py --montgomery 1 --steps 8 --warmup 2 --cpu-sample 0 --pmc-traffic 0 --other-configs 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('headline montgomery', d['value'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])" | tee gpurun_out/r03_zx_mont.txt
timeout 600 python tools/bench_abc.py 2>/dev/null | tail -1 | tee gpurun_out/r03_zx_abc.json | cut -c1-420
