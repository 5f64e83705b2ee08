set -x
OUT=gpurun_out; mkdir -p $OUT
T=zk-email-verify_amd/data/templates/@zk-email/zk-regex-circom/circuits/common/body_hash_regex.circom
ls $T
timeout 120 tools/mulbench $OUT/r05_a_mulbench.json 2>&1 | tee $OUT/r05_a_mulbench.txt
timeout 600 python -m pytest tests/test_msm.py -m gpu -x -q 2>&1 | tail -15 | tee $OUT/r05_a_msm_tests.txt
for l in 18 20; do timeout 200 python tools/bench_msm.py --log2 $l 2>&1 | tail -2 | tee -a $OUT/r05_a_msm_bench.txt; done
timeout 200 python tools/bench_msm.py --log2 20 --witness 1 2>&1 | tail -2 | tee -a $OUT/r05_a_msm_bench.txt
bash tools/gpu_call.sh r05_a benchq "benchq:--regex $T" env:ZKWG_NET_FILL_LATE=1 "benchq:--regex $T" env:ZKWG_NET_FILL_LATE=0
