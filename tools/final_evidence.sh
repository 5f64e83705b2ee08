OUT=$PWD/gpurun_out; REPO=$PWD; TAG=r01_g
timeout 600 python bench.py 2>/dev/null | tail -1 > $OUT/${TAG}_bench.json
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -- python $REPO/bench.py --cpu-sample 0 > $OUT/${TAG}_prof_bench.json 2> $OUT/${TAG}_prof.log )
python - <<PY
import json
for f in ("bench","prof_bench"):
    d = json.loads(open("$OUT/${TAG}_%s.json" % f).read().strip().splitlines()[-1])
    print(f, d["value"], d["steps"], d["warmup"], d["roofline"]["avg_launch_ms"], d["roofline"]["achieved"], d["roofline"]["frac"], d.get("cpu_baseline",{}).get("value"))
PY
find $OUT/${TAG}_prof -name "*kernel_stats.csv" | head -1 | xargs -r head -4 | cut -c1-160
