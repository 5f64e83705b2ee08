#!/bin/bash
# GPU call 31: HBM traffic of the complete --O0 pipeline per kernel (PMC WRITE_SIZE / FETCH_SIZE, separate passes with --kernel-trace only,
# as /opt/skills/guides/MI355X_MICROARCH.md prescribes), tools/bench_full.py on EmailVerifier(576,192)
OUT=$PWD/gpurun_out; REPO=$PWD; mkdir -p $OUT
for C in WRITE_SIZE FETCH_SIZE; do
  ( cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/r03_pmc_o0_$C -- python $REPO/tools/bench_full.py > /dev/null 2> $OUT/r03_pmc_o0_$C.log )
done
python - <<PY
import csv, glob, json
res = {}
for C in ("WRITE_SIZE", "FETCH_SIZE"):
    f = glob.glob("$OUT/r03_pmc_o0_%s/**/*counter_collection.csv" % C, recursive=True)
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"].split("(")[0]
        if k.startswith("zk_"):
            res.setdefault(k, {}).setdefault(C, []).append(float(r["Counter_Value"]))
out = {}
for k, v in res.items():
    w = v.get("WRITE_SIZE", [0]); fch = v.get("FETCH_SIZE", [0])
    out[k] = {"launches": len(w), "write_GB_per_launch": sum(w) / len(w) * 1024 / 1e9, "fetch_GB_per_launch_corrected_x2": sum(fch) / len(fch) * 1024 * 2 / 1e9}
W, tile = 3113238, 256
out["_note"] = "zk_expand3_o0_k2: %d emails per launch, algorithmic bytes %.3f GB (32 B x %d wires x %d emails); FETCH_SIZE x2 = the guide's gfx950 correction" % (tile, 32 * W * tile / 1e9, W, tile)
json.dump(out, open("$OUT/r03_pmc_o0_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
rm -rf $OUT/r03_pmc_o0_WRITE_SIZE $OUT/r03_pmc_o0_FETCH_SIZE
