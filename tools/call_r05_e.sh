set -x
OUT=gpurun_out; mkdir -p $OUT
timeout 300 tools/chunkbench 96 1 58 2>&1 | tee $OUT/r05_e_chunkbench.txt | tail -12
timeout 900 python -m pytest tests/test_prove.py tests/test_msm.py -m gpu -x -q -k "fixed_base or synthetic or equals_the_oracle" 2>&1 | tail -6 | tee $OUT/r05_e_prove_tests.txt
timeout 400 python tools/bench_prove.py 2>&1 | tail -1 | tee $OUT/r05_e_bench_prove.json
for l in 20; do timeout 200 python tools/bench_msm.py --log2 $l 2>&1 | tail -1 | tee -a $OUT/r05_e_msm_bench.txt; done
