set -x
OUT=gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_prove.py tests/test_msm.py -m gpu -x -q -k "synthetic or fixed_base or equals_the_oracle" 2>&1 | tail -5 | tee $OUT/r05_i_prove_tests.txt
timeout 300 python tools/bench_prove.py --slots 16 --proofs 64 2>&1 | tail -1 | tee $OUT/r05_i_bench_prove.json
GPU_MAX_HW_QUEUES=8 timeout 300 python tools/bench_prove.py --slots 16 --proofs 64 2>&1 | tail -1 | tee -a $OUT/r05_i_bench_prove.json
GPU_MAX_HW_QUEUES=16 timeout 300 python tools/bench_prove.py --slots 32 --proofs 96 2>&1 | tail -1 | tee -a $OUT/r05_i_bench_prove.json
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/r05_i_prof -- python $OLDPWD/tools/bench_prove.py --slots 16 --proofs 32 > /dev/null 2> $OLDPWD/$OUT/r05_i_prof.log )
S=$(find $OUT/r05_i_prof -name "*kernel_stats.csv" | head -1); [ -n "$S" ] && cp $S $OUT/r05_i_prove_kernel_stats.csv && head -30 $S | cut -c1-150
rm -rf $OUT/r05_i_prof
