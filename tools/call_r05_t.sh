set -x
# multi-exponentiation latency: bucket sums by bit planes (ZKWG_MSM_PLANES=1, default) against the (S, A) tree (=0), with 8 bases per lane
# in zk_msm_ones, 8-way joins and slices of 16 / 8 / 8; parity first
OUT=gpurun_out; mkdir -p $OUT
export GPU_MAX_HW_QUEUES=16
timeout 900 python -m pytest tests/test_msm.py tests/test_prove.py -m gpu -x -q 2>&1 | tail -5 | tee $OUT/r05_t_tests.txt
for P in 1 0; do
  export ZKWG_MSM_PLANES=$P
  timeout 300 python tools/bench_prove.py 2>&1 | tail -1 | tee -a $OUT/r05_t_bench_prove.json
  timeout 600 python tools/bench_prove.py --max-header 1024 --max-body 1536 --emails 8 --slots 24 --proofs 72 2>&1 | tail -1 | tee -a $OUT/r05_t_bench_prove.json
done
