set -x
OUT=gpurun_out; mkdir -p $OUT
T=$PWD/zk-email-verify_amd/data/templates/zk-regex-circom/circuits/common/body_hash_regex.circom
U=$PWD/tests/golden/regex_style/body_hash_regex_unshared.circom
timeout 900 python -m pytest tests/test_prove.py tests/test_msm.py -m gpu -x -q -k "not h_sized" 2>&1 | tail -15 | tee $OUT/r05_d_prove_tests.txt
timeout 400 python tools/bench_prove.py 2>&1 | tail -3 | tee $OUT/r05_d_bench_prove.json
bash tools/gpu_call.sh r05_d files:tests/test_regex_template.py benchq "benchq:--regex $T" "benchq:--regex $U"
