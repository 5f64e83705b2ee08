#!/bin/bash
# HBM traffic of zk_expand from PMC counters, two separate rocprofv3 passes (WRITE_SIZE, FETCH_SIZE) as
# /opt/skills/guides/MI355X_MICROARCH.md prescribes; writes gpurun_out/<tag>_pmc_{WRITE,FETCH}_SIZE.csv and
# gpurun_out/<tag>_pmc_traffic.json (copy the three files into profiles/ to have bench.py report `traffic`).
TAG=${1:-r01_e}
OUT=$PWD/gpurun_out
REPO=$PWD
mkdir -p $OUT
for C in WRITE_SIZE FETCH_SIZE; do
  ( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/${TAG}_pmc_$C -- \
      python $REPO/bench.py --steps 1 --warmup 0 --cpu-sample 0 --distinct 64 > /dev/null 2> $OUT/${TAG}_pmc_$C.log )
done
python - <<PY
import csv, glob, json
res = {}
for C in ("WRITE_SIZE", "FETCH_SIZE"):
    f = glob.glob("$OUT/${TAG}_pmc_%s/*/*counter_collection.csv" % C)
    rows = list(csv.DictReader(open(f[0])))
    agg = {}
    for r in rows:
        k = r["Kernel_Name"].split("(")[0]
        agg.setdefault(k, []).append(float(r["Counter_Value"]))
    with open("$OUT/${TAG}_pmc_%s.csv" % C, "w") as o:
        o.write("Kernel_Name,Counter_Name,Launches,Mean_Counter_Value\n")
        for k, v in agg.items():
            o.write("%s,%s,%d,%f\n" % (k, C, len(v), sum(v) / len(v)))
    v = agg["zk_expand_256"]
    res[C] = sum(v) / len(v)
W, tile = 1776821, 512
write_b = res["WRITE_SIZE"] * 1024
fetch_b = res["FETCH_SIZE"] * 1024 * 2   # gfx950: FETCH_SIZE counts 64 B per 128 B request (guide, HBM section)
alg = 32 * W * tile
json.dump({"source": "tools/pmc_traffic.sh: rocprofv3 --kernel-trace --pmc WRITE_SIZE / --pmc FETCH_SIZE (separate passes) -- python bench.py --steps 1 --warmup 0 --cpu-sample 0 --distinct 64",
           "kernel": "zk_expand_256", "workload": "EmailVerifier(1024,1536,121,17,0,0,0,0)", "tile": tile, "witness_len": W,
           "WRITE_SIZE_KB_per_launch": res["WRITE_SIZE"], "FETCH_SIZE_KB_per_launch_raw": res["FETCH_SIZE"],
           "write_bytes_per_launch": write_b, "fetch_bytes_per_launch_corrected": fetch_b,
           "write_over_algorithmic": write_b / alg, "traffic_bytes_per_launch": write_b + fetch_b},
          open("$OUT/${TAG}_pmc_traffic.json", "w"), indent=1)
print("WRITE/alg", write_b / alg, "fetch GB", fetch_b / 1e9)
PY
