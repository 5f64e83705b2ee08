#!/bin/bash
# GPU call 12/13: per-kernel times of the complete --O0 pipeline (which of the row kernels bounds prepare)
mkdir -p gpurun_out/r03_${TAG:-l}_prof; cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r03_${TAG:-l}_prof -o o0 --output-format csv -- python $GRAFT_REPO_ROOT/tools/bench_full.py > $GRAFT_REPO_ROOT/gpurun_out/r03_${TAG:-l}_full.txt 2>&1
cd $GRAFT_REPO_ROOT; tail -2 gpurun_out/r03_${TAG:-l}_full.txt
f=$(find gpurun_out/r03_${TAG:-l}_prof -name '*kernel_stats.csv' | head -1); head -20 $f; cp $f gpurun_out/r03_${TAG:-l}_kernel_stats.csv; rm -rf gpurun_out/r03_${TAG:-l}_prof
