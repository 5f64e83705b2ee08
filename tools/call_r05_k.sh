set -x
OUT=gpurun_out; mkdir -p $OUT
export GPU_MAX_HW_QUEUES=16
timeout 1200 python -m pytest tests/test_prove.py -m gpu -x -q 2>&1 | tail -8 | tee $OUT/r05_k_prove_tests.txt
timeout 300 python tools/bench_prove.py 2>&1 | tail -1 | tee $OUT/r05_k_bench_prove.json
timeout 600 python tools/bench_prove.py --max-header 1024 --max-body 1536 --emails 8 --slots 24 --proofs 72 2>&1 | tail -1 | tee -a $OUT/r05_k_bench_prove.json
