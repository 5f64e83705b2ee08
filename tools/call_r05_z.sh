set -x
# Evidence call at HEAD: the whole GPU suite, smoke, the default bench line (PMC traffic, other configurations, CPU baseline), the
# rocprofv3 kernel statistics of the headline pipeline, removeSoftLineBreaks, the prover at both sizes
TAG=${1:-r05_z}
OUT=gpurun_out; mkdir -p $OUT
bash tools/gpu_call.sh $TAG alltests smoke bench prof rslb
timeout 300 python tools/bench_prove.py 2>&1 | tail -1 | tee $OUT/${TAG}_bench_prove.json
timeout 600 python tools/bench_prove.py --max-header 1024 --max-body 1536 --emails 8 --slots 24 --proofs 72 2>&1 | tail -1 | tee -a $OUT/${TAG}_bench_prove.json
