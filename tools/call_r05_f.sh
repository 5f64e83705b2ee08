set -x
OUT=gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_prove.py -m gpu -x -q -k "synthetic" 2>&1 | tail -6 | tee $OUT/r05_f_prove_tests.txt
timeout 400 python tools/bench_prove.py 2>&1 | tail -1 | tee $OUT/r05_f_bench_prove.json
timeout 400 python tools/bench_prove.py --slots 16 --proofs 64 2>&1 | tail -1 | tee -a $OUT/r05_f_bench_prove.json
bash tools/gpu_call.sh r05_f benchq "benchq:--place-ring 1" "benchq:--place-ring 0" "benchq:--montgomery 1" "benchq:--montgomery 1 --place-ring 1"
( time bash tools/gpu_call.sh r05_f bench ) 2>&1 | tail -40
