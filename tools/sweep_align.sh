for al in 0 4096 256 0 4096; do
  timeout 200 python bench.py --out-align $al --steps 5 --warmup 2 --cpu-sample 0 --pmc-traffic 0 --other-configs 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('align', $al, d['value'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
done
