set -x
OUT=gpurun_out; mkdir -p $OUT
T=$PWD/zk-email-verify_amd/data/templates/zk-regex-circom/circuits/common/body_hash_regex.circom
timeout 900 python -m pytest tests/test_prove.py tests/test_msm.py -m gpu -x -q -k "synthetic or fixed_base or equals_the_oracle or linear" 2>&1 | tail -5 | tee $OUT/r05_j_prove_tests.txt
timeout 300 python tools/bench_prove.py 2>&1 | tail -1 | tee $OUT/r05_j_bench_prove.json
ZKWG_MSM_PRECOMP=0 timeout 300 python tools/bench_prove.py 2>&1 | tail -1 | tee -a $OUT/r05_j_bench_prove.json
timeout 200 python tools/bench_msm.py --log2 20 2>&1 | tail -1 | tee $OUT/r05_j_msm_bench.txt
bash tools/gpu_call.sh r05_j "benchq:--montgomery 1" env:ZKWG_X3_K=2 "benchq:--montgomery 1" env:ZKWG_X3_K=4 "benchq:--regex $T --prep-batch 1024" "benchq:--regex $T --prep-batch 4096" "benchq:--regex $T"
