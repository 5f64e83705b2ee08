set -x
# the counting sort with workgroup-local histograms (ZKWG_MSM_LDS_SORT=1, default) against one global atomic per digit (=0): parity, one
# H-shaped sum alone, the batched prover at both sizes
OUT=gpurun_out; mkdir -p $OUT
export GPU_MAX_HW_QUEUES=16
timeout 900 python -m pytest tests/test_msm.py tests/test_prove.py -m gpu -x -q 2>&1 | tail -5 | tee $OUT/r05_v_tests.txt
for L in 1 0; do
  export ZKWG_MSM_LDS_SORT=$L
  timeout 200 python tools/bench_msm.py --log2 21 --reps 5 2>&1 | tail -1 | tee -a $OUT/r05_v_msm.json
  timeout 300 python tools/bench_prove.py 2>&1 | tail -1 | tee -a $OUT/r05_v_bench_prove.json
  timeout 600 python tools/bench_prove.py --max-header 1024 --max-body 1536 --emails 8 --slots 24 --proofs 72 2>&1 | tail -1 | tee -a $OUT/r05_v_bench_prove.json
done
