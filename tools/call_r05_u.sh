set -x
# per-kernel times of ONE multi-exponentiation alone: H-shaped (2^21 full-size scalars) and witness-shaped (2^21, mostly bits)
OUT=$PWD/gpurun_out; mkdir -p $OUT
REPO=$PWD
for W in 0 1; do
( cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r05_u_prof$W -- \
    python $REPO/tools/bench_msm.py --log2 21 --reps 5 --witness $W > $OUT/r05_u_msm$W.json 2> $OUT/r05_u_prof$W.log )
S=$(find $OUT/r05_u_prof$W -name "*kernel_stats.csv" | head -1); [ -n "$S" ] && cp $S $OUT/r05_u_msm${W}_kernel_stats.csv && head -16 $S | cut -c1-140
rm -rf $OUT/r05_u_prof$W; tail -1 $OUT/r05_u_msm$W.json
done
