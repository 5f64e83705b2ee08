#!/bin/bash
# Evidence run for one round on a GPU box (invoked through gpurun from the repo root):
# full GPU test suite, smoke, the default bench line (with in-invocation PMC traffic, other_configs and
# cpu_baseline), a rocprofv3 kernel-stats pass of the same pipeline, a VALU-instruction PMC pass for the
# zk_rsa issue "roofline", and the removeSoftLineBreaks variant.  Everything lands in gpurun_out/<tag>_*.
TAG=${1:-r02_final}
OUT=$PWD/gpurun_out
mkdir -p $OUT
REPO=$PWD
( timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -5 ) > $OUT/${TAG}_pytest_gpu.txt
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 ) > $OUT/${TAG}_smoke.txt
timeout 900 python bench.py 2>$OUT/${TAG}_bench.err | tail -1 > $OUT/${TAG}_bench.json
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -- \
    python $REPO/bench.py --steps 5 --warmup 2 --cpu-sample 0 --pmc-traffic 0 --other-configs 0 > $OUT/${TAG}_prof_bench.json 2> $OUT/${TAG}_prof.log )
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES --output-format csv -d $OUT/${TAG}_pmc_valu -- \
    python $REPO/bench.py --steps 1 --warmup 0 --cpu-sample 0 --pmc-traffic 0 --other-configs 0 --distinct 64 > /dev/null 2> $OUT/${TAG}_pmc_valu.log )
timeout 600 python bench.py --remove-soft-line-breaks 1 --batch 4096 --tile 256 --prep-batch 4096 --ring 4 --steps 12 --warmup 2 --cpu-sample 0 --pmc-traffic 0 --other-configs 0 2>/dev/null | tail -1 > $OUT/${TAG}_rslb.json
for f in pytest_gpu.txt smoke.txt; do echo "== $f"; cat $OUT/${TAG}_$f; done
for f in bench rslb; do echo "== $f"; python - <<PY
import json
try:
    d = json.loads(open("$OUT/${TAG}_$f.json").read().strip().splitlines()[-1])
    print(d["value"], d["roofline"]["achieved"], d["roofline"]["frac"], d["roofline"].get("traffic"), d.get("cpu_baseline", {}), d["kernel_ms_per_launch"])
    print(json.dumps(d.get("other_configs", {}), indent=1))
except Exception as e:
    print("failed:", e)
PY
done
S=$(find $OUT/${TAG}_prof -name "*kernel_stats.csv" | head -1); [ -n "$S" ] && cp $S $OUT/${TAG}_kernel_stats.csv && head -14 $S
python - <<PY
import csv, glob, json
f = glob.glob("$OUT/${TAG}_pmc_valu/**/*counter_collection.csv", recursive=True)
if f:
    agg = {}
    for r in csv.DictReader(open(f[0])):
        k = (r["Kernel_Name"].split("(")[0], r["Counter_Name"])
        agg.setdefault(k, []).append(float(r["Counter_Value"]))
    out = {"%s:%s" % k: {"launches": len(v), "mean": sum(v) / len(v)} for k, v in agg.items()}
    json.dump(out, open("$OUT/${TAG}_pmc_valu.json", "w"), indent=1)
    for k, v in sorted(out.items()):
        print(k, v)
else:
    print("no VALU counter file")
PY
rm -rf $OUT/${TAG}_prof $OUT/${TAG}_pmc_valu
