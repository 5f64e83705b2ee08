set -x
OUT=gpurun_out; mkdir -p $OUT
T=$PWD/zk-email-verify_amd/data/templates/zk-regex-circom/circuits/common/body_hash_regex.circom
U=$PWD/tests/golden/regex_style/body_hash_regex_unshared.circom
bash tools/gpu_call.sh r05_c files:tests/test_regex_template.py benchq "benchq:--regex $T" "benchq:--regex $U" "prof:--regex $T"
cp $OUT/r05_c_kernel_stats.csv $OUT/r05_c_template_kernel_stats.csv
