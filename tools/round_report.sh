#!/bin/bash
# Evidence run for one round on a GPU box (invoked through gpurun from the repo root):
# full GPU test suite, smoke, the default bench line, a rocprofv3 kernel-stats pass of the same command,
# and the side configurations quoted in BASELINE.md.  Everything lands in gpurun_out/<tag>_*.
TAG=${1:-r01_final}
OUT=$PWD/gpurun_out
mkdir -p $OUT
REPO=$PWD
( timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -5 ) > $OUT/${TAG}_pytest_gpu.txt
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 ) > $OUT/${TAG}_smoke.txt
timeout 600 python bench.py 2>/dev/null | tail -1 > $OUT/${TAG}_bench.json
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -- \
    python $REPO/bench.py --steps 2 --warmup 1 --cpu-sample 0 > $OUT/${TAG}_prof_bench.json 2> $OUT/${TAG}_prof.log )
timeout 300 python bench.py --batch 256 --tile 256 --prep-batch 256 --distinct 256 --steps 10 --warmup 2 --cpu-sample 0 2>/dev/null | tail -1 > $OUT/${TAG}_c2_batch256.json
timeout 600 python bench.py --max-body 65536 --body-len 60000 --batch 1024 --tile 32 --prep-batch 128 --distinct 32 --steps 2 --warmup 1 --cpu-sample 32 2>/dev/null | tail -1 > $OUT/${TAG}_c5_longbody.json
timeout 600 python bench.py --remove-soft-line-breaks 1 --batch 4096 --tile 256 --prep-batch 4096 --ring 4 --steps 12 --warmup 2 --cpu-sample 0 2>/dev/null | tail -1 > $OUT/${TAG}_rslb.json
for f in pytest_gpu.txt smoke.txt; do echo "== $f"; cat $OUT/${TAG}_$f; done
for f in bench c2_batch256 c5_longbody rslb; do echo "== $f"; python - <<PY
import json
try:
    d = json.loads(open("$OUT/${TAG}_$f.json").read().strip().splitlines()[-1])
    print(d["value"], d["roofline"]["achieved"], d["roofline"]["frac"], d.get("cpu_baseline", {}).get("value"), d["kernel_ms_per_launch"])
except Exception as e:
    print("failed:", e)
PY
done
find $OUT/${TAG}_prof -name "*kernel_stats.csv" | head -1 | xargs -r head -12
