set -x
# rocprofv3 kernel statistics of the batched prover at the headline circuit's size (precomputed windows, 24 proofs in flight)
OUT=$PWD/gpurun_out; mkdir -p $OUT
REPO=$PWD
export GPU_MAX_HW_QUEUES=16
( cd /tmp && export TMPDIR=/tmp && timeout 800 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r05_s_prof -- \
    python $REPO/tools/bench_prove.py --max-header 1024 --max-body 1536 --emails 8 --slots 24 --proofs 72 > $OUT/r05_s_bench_prove.json 2> $OUT/r05_s_prof.log )
S=$(find $OUT/r05_s_prof -name "*kernel_stats.csv" | head -1); [ -n "$S" ] && cp $S $OUT/r05_s_prove_kernel_stats.csv && head -24 $S | cut -c1-160
rm -rf $OUT/r05_s_prof; tail -1 $OUT/r05_s_bench_prove.json | cut -c1-600
