#!/bin/bash
# GPU call 32: HBM traffic of the prover-stage-1-from-the-image pipeline per kernel (PMC WRITE_SIZE / FETCH_SIZE, separate passes with --kernel-trace only,
# as /opt/skills/guides/MI355X_MICROARCH.md prescribes), tools/bench_abc.py on EmailVerifier(576,192)
OUT=$PWD/gpurun_out; REPO=$PWD; mkdir -p $OUT
for C in WRITE_SIZE FETCH_SIZE; do
  ( cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/r03_pmc_abc_$C -- python $REPO/tools/bench_abc.py > /dev/null 2> $OUT/r03_pmc_abc_$C.log )
done
python - <<PY
import csv, glob, json
res = {}
for C in ("WRITE_SIZE", "FETCH_SIZE"):
    f = glob.glob("$OUT/r03_pmc_abc_%s/**/*counter_collection.csv" % C, recursive=True)
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"].split("(")[0]
        if k.startswith("zk_"):
            res.setdefault(k, {}).setdefault(C, []).append(float(r["Counter_Value"]))
out = {}
for k, v in res.items():
    w = v.get("WRITE_SIZE", [0]); fch = v.get("FETCH_SIZE", [0])
    out[k] = {"launches": len(w), "write_GB_per_launch": sum(w) / len(w) * 1024 / 1e9, "fetch_GB_per_launch_corrected_x2": sum(fch) / len(fch) * 1024 * 2 / 1e9}
W, tile = 2261421, 256
out["_note"] = "zk_expand3_o0_k2 (A.w|B.w|C.w of EmailVerifier(576,192)): %d emails per launch, algorithmic bytes %.3f GB (32 B x %d values x %d emails); FETCH_SIZE x2 = the guide's gfx950 correction" % (tile, 32 * W * tile / 1e9, W, tile)
json.dump(out, open("$OUT/r03_pmc_abc_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
rm -rf $OUT/r03_pmc_abc_WRITE_SIZE $OUT/r03_pmc_abc_FETCH_SIZE
