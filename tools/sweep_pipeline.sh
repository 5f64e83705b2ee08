# same-box sweep of the pipeline knobs (each configuration twice, interleaved)
timeout 300 python bench.py --steps 3 --warmup 1 --cpu-sample 0 > /dev/null 2>&1
for rep in 1 2; do
for cfg in "3 1024" "2 1024" "4 1024" "0 1024" "3 512" "3 2048"; do
  set -- $cfg
  timeout 300 python bench.py --steps 5 --warmup 2 --cpu-sample 0 --rsa-throttle $1 --prep-batch $2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('throttle=$1 prep=$2', d['value'], d['roofline']['achieved'], d['kernel_ms_per_launch']['zk_rsa'])"
done; done
