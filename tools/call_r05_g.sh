set -x
OUT=gpurun_out; mkdir -p $OUT
timeout 200 tools/mulbench $OUT/r05_g_mulbench.json 2>&1 | tail -9 | tee $OUT/r05_g_mulbench.txt
bash tools/gpu_call.sh r05_g alltests smoke
timeout 400 python tools/bench_prove.py --slots 16 --proofs 64 2>&1 | tail -1 | tee $OUT/r05_g_bench_prove.json
bash tools/gpu_call.sh r05_g rslb bench
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/r05_g_nttprof -- python $OLDPWD/tools/bench_ntt.py > $OLDPWD/$OUT/r05_g_ntt.json 2> $OLDPWD/$OUT/r05_g_nttprof.log )
S=$(find $OUT/r05_g_nttprof -name "*kernel_stats.csv" | head -1); [ -n "$S" ] && cp $S $OUT/r05_g_ntt_kernel_stats.csv && head -8 $S
rm -rf $OUT/r05_g_nttprof; tail -2 $OUT/r05_g_ntt.json
