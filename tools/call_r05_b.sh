set -x
OUT=gpurun_out; mkdir -p $OUT
T=zk-email-verify_amd/data/templates/zk-regex-circom/circuits/common/body_hash_regex.circom
U=tests/golden/regex_style/body_hash_regex_unshared.circom
ls $T $U
timeout 200 tools/mulbench $OUT/r05_b_mulbench.json 2>&1 | tail -12 | tee $OUT/r05_b_mulbench.txt
bash tools/gpu_call.sh r05_b files:tests/test_regex_template.py,tests/test_host_expand.py,tests/test_ev_gpu.py benchq "benchq:--regex $T" "benchq:--regex $U" "prof:--regex $T"
cp $OUT/r05_b_kernel_stats.csv $OUT/r05_b_template_kernel_stats.csv
