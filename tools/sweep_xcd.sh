timeout 300 python bench.py --steps 3 --warmup 1 --cpu-sample 0 > /dev/null 2>&1
for rep in 1 2 3; do
for cfg in "2048 512" "4096 512" "2048 1024" "4096 1024" "3072 1024"; do
  set -- $cfg
  ZKWG_PORTION=$1 timeout 300 python bench.py --steps 5 --warmup 2 --cpu-sample 0 --tile $2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('portion=$1 tile=$2', d['value'], d['roofline']['achieved'])"
done; done
