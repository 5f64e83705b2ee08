#!/bin/bash
# GPU call 18: per-kernel times of the prover-stage-1-from-the-image pipeline (tools/bench_abc.py)
TAG=${TAG:-p}
mkdir -p gpurun_out/r03_${TAG}_prof; cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r03_${TAG}_prof -o abc --output-format csv -- python $GRAFT_REPO_ROOT/tools/bench_abc.py > $GRAFT_REPO_ROOT/gpurun_out/r03_${TAG}_abc.txt 2>&1
cd $GRAFT_REPO_ROOT; tail -1 gpurun_out/r03_${TAG}_abc.txt | cut -c1-900
f=$(find gpurun_out/r03_${TAG}_prof -name '*kernel_stats.csv' | head -1); head -16 $f | cut -c1-150; cp $f gpurun_out/r03_${TAG}_abc_kernel_stats.csv; rm -rf gpurun_out/r03_${TAG}_prof
