#!/bin/bash
# One parameterised GPU call (replaces the per-call scripts of earlier rounds): run through gpurun from the repo root,
#     gpurun --timeout 1500 -- 'bash tools/gpu_call.sh r04_a newtests beside bench'
# Every step writes gpurun_out/<tag>_<step>.* and prints a short summary; summaries worth keeping are copied to profiles/.
#   tests:<expr>  pytest -m gpu -k <expr>      alltests  the whole GPU suite      smoke  __graft_entry__.smoke()
#   files:<a,b>   pytest -m gpu on these test files
#   beside        tools/beside.py (which prepare kernel costs zk_expand how much)
#   bench         the default bench.py line     benchq  the headline only (no PMC / other configs / CPU baseline)
#   prof          rocprofv3 --kernel-trace --stats of the headline pipeline
#   rslb          removeSoftLineBreaks = 1 variant (rslb:<label> appends to <tag>_rslb_variants.json)      abc  tools/bench_abc.py      o0  tools/bench_full.py
#   pmc:<tool.py> per-kernel HBM traffic of a tool (two rocprofv3 --pmc passes)      pmcv:<tool.py>  VALU wavefront-instructions per launch of its kernels
#   prove[:<args>]  tools/bench_prove.py (appends one JSON line to <tag>_bench_prove.json)      provep[:<args>]  the same under rocprofv3 --kernel-trace --stats
#   msm:<args>    tools/bench_msm.py (msmp:<args> under rocprofv3)      ntt[:<args>]  tools/bench_ntt.py (nttp[:<args>] under rocprofv3)      run:<command>  anything else (output tail -> <tag>_run.txt)
#   env:K=V       export K=V for the following steps
TAG=$1; shift
OUT=$PWD/gpurun_out; mkdir -p $OUT
REPO=$PWD
for step in "$@"; do
  echo "=== $step"
  case $step in
    env:*) export "${step#env:}" ;;
    tests:*) timeout 1500 python -m pytest tests -m gpu -x -q -k "${step#tests:}" 2>&1 | tail -25 | tee $OUT/${TAG}_tests.txt ;;
    files:*) timeout 1500 python -m pytest $(echo "${step#files:}" | tr ',' ' ') -m gpu -x -q 2>&1 | tail -25 | tee $OUT/${TAG}_tests.txt ;;
    alltests) timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -6 | tee $OUT/${TAG}_pytest_gpu.txt ;;
    smoke) timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/${TAG}_smoke.txt ;;
    beside) timeout 600 python tools/beside.py --out $OUT/${TAG}_beside.json 2>&1 | tail -20 ;;
    beside:*) timeout 600 python tools/beside.py --modes "${step#beside:}" --out $OUT/${TAG}_beside.json 2>&1 | tail -20 ;;
    bench) timeout 1200 python bench.py 2>$OUT/${TAG}_bench.err | tail -1 > $OUT/${TAG}_bench.json
           python - <<PY
import json
try:
    d = json.loads(open("$OUT/${TAG}_bench.json").read().strip().splitlines()[-1])
    r = d["roofline"]
    print("value", d["value"], "frac", r["frac"], "alone", r.get("alone_frac"), "fill", r.get("box_fill_GBps"), "traffic x", r.get("traffic_over_algorithmic"), d.get("cpu_baseline", {}).get("value"))
    print(d["kernel_ms_per_launch"])
    for k, v in d.get("other_configs", {}).items():
        print(" ", k, {a: b for a, b in v.items() if a not in ("note", "sample", "gate_list", "unit")})
except Exception as e:
    print("bench failed:", e); print(open("$OUT/${TAG}_bench.err").read()[-1500:])
PY
           ;;
    benchq) timeout 600 python bench.py --steps 8 --warmup 2 --cpu-sample 0 --pmc-traffic 0 --other-configs 0 2>$OUT/${TAG}_benchq.err | tail -1 | tee $OUT/${TAG}_benchq.json | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('headline', d['value'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['kernel_ms_per_launch'])" ;;
    benchq:*) timeout 600 python bench.py --steps 8 --warmup 2 --cpu-sample 0 --pmc-traffic 0 --other-configs 0 ${step#benchq:} 2>$OUT/${TAG}_benchq.err | tail -1 | tee -a $OUT/${TAG}_benchq_variants.json | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('headline ${step#benchq:}', d['value'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['kernel_ms_per_launch'])" ;;
    prof) ( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -- \
              python $REPO/bench.py --steps 5 --warmup 2 --cpu-sample 0 --pmc-traffic 0 --other-configs 0 > $OUT/${TAG}_prof_bench.json 2> $OUT/${TAG}_prof.log )
          S=$(find $OUT/${TAG}_prof -name "*kernel_stats.csv" | head -1); [ -n "$S" ] && cp $S $OUT/${TAG}_kernel_stats.csv && head -16 $S
          rm -rf $OUT/${TAG}_prof ;;
    prof:*) ( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -- \
              python $REPO/bench.py --steps 5 --warmup 2 --cpu-sample 0 --pmc-traffic 0 --other-configs 0 ${step#prof:} > $OUT/${TAG}_prof_bench.json 2> $OUT/${TAG}_prof.log )
          S=$(find $OUT/${TAG}_prof -name "*kernel_stats.csv" | head -1); [ -n "$S" ] && cp $S $OUT/${TAG}_kernel_stats.csv && head -16 $S
          rm -rf $OUT/${TAG}_prof ;;
    rslb) timeout 600 python bench.py --remove-soft-line-breaks 1 --batch 4096 --tile 256 --prep-batch 4096 --ring 6 --steps 12 --warmup 2 --cpu-sample 0 --pmc-traffic 0 --other-configs 0 2>/dev/null | tail -1 | tee $OUT/${TAG}_rslb.json | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('rslb', d['value'], d['kernel_ms_per_launch'])" ;;
    rslb:*) timeout 600 python bench.py --remove-soft-line-breaks 1 --batch 4096 --tile ${RSLB_TILE:-256} --prep-batch ${RSLB_PREP:-4096} --ring ${RSLB_RING:-6} --steps 12 --warmup 2 --cpu-sample 0 --pmc-traffic 0 --other-configs 0 $RSLB_ARGS 2>/dev/null | tail -1 | tee -a $OUT/${TAG}_rslb_variants.json | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('rslb ${step#rslb:}', d['value'], d['kernel_ms_per_launch'])" ;;
    abc) timeout 900 python tools/bench_abc.py 2>/dev/null | tail -1 | tee $OUT/${TAG}_abc.json | cut -c1-600 ;;
    o0) timeout 900 python tools/bench_full.py 2>/dev/null | tail -1 | tee $OUT/${TAG}_o0.json | cut -c1-600 ;;
    pmc:*) # per-kernel HBM traffic of a tool: two separate rocprofv3 --pmc passes (WRITE_SIZE, FETCH_SIZE; KiB units, FETCH x 2 on gfx950)
      TOOL=${step#pmc:}
      for CTR in WRITE_SIZE FETCH_SIZE; do
        ( cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --pmc $CTR --output-format csv -d $OUT/${TAG}_pmc_$CTR -- python $REPO/$TOOL > /dev/null 2> $OUT/${TAG}_pmc_$CTR.log )
      done
      python - <<PY
import csv, glob, json
agg = {}
for ctr, scale in (("WRITE_SIZE", 1024.0), ("FETCH_SIZE", 2048.0)):
    f = glob.glob("$OUT/${TAG}_pmc_%s/**/*counter_collection.csv" % ctr, recursive=True)
    if not f:
        print("no counter file for", ctr); continue
    for r in csv.DictReader(open(f[0])):
        if r.get("Counter_Name", ctr) != ctr: continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")      # (templated kernels print as "void zk_...<...>(...)")
        a = agg.setdefault(k, {"launches": 0, "WRITE_SIZE": 0.0, "FETCH_SIZE": 0.0, "n": {"WRITE_SIZE": 0, "FETCH_SIZE": 0}})
        a[ctr] += float(r["Counter_Value"]) * scale; a["n"][ctr] += 1
out = {k: {"launches": v["n"]["WRITE_SIZE"], "write_GB_per_launch": v["WRITE_SIZE"] / max(v["n"]["WRITE_SIZE"], 1) / 1e9,
           "fetch_GB_per_launch_corrected_x2": v["FETCH_SIZE"] / max(v["n"]["FETCH_SIZE"], 1) / 1e9} for k, v in agg.items() if k.startswith("zk_")}
json.dump(out, open("$OUT/${TAG}_pmc_traffic.json", "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["write_GB_per_launch"] * kv[1]["launches"])[:14]:
    print("%-28s launches %4d  write %8.3f GB  fetch %8.3f GB per launch" % (k, v["launches"], v["write_GB_per_launch"], v["fetch_GB_per_launch_corrected_x2"]))
PY
      rm -rf $OUT/${TAG}_pmc_WRITE_SIZE $OUT/${TAG}_pmc_FETCH_SIZE ;;
    pmcv:*) # VALU wavefront-instructions per launch of every kernel of a tool (one rocprofv3 --pmc pass: SQ_INSTS_VALU SQ_WAVES)
      TOOL=${step#pmcv:}
      ( cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES --output-format csv -d $OUT/${TAG}_pmcv -- python $REPO/$TOOL > /dev/null 2> $OUT/${TAG}_pmcv.log )
      python - <<PY
import csv, glob, json
agg = {}
f = glob.glob("$OUT/${TAG}_pmcv/**/*counter_collection.csv", recursive=True)
for r in (csv.DictReader(open(f[0])) if f else []):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    a = agg.setdefault(k, {"SQ_INSTS_VALU": 0.0, "SQ_WAVES": 0.0, "n": 0})
    if r["Counter_Name"] in a:
        a[r["Counter_Name"]] += float(r["Counter_Value"])
        a["n"] += r["Counter_Name"] == "SQ_WAVES"
out = {k: {"launches": v["n"], "valu_wave_insts_per_launch": v["SQ_INSTS_VALU"] / max(v["n"], 1), "waves_per_launch": v["SQ_WAVES"] / max(v["n"], 1)} for k, v in agg.items() if "zk_" in k}
json.dump(out, open("$OUT/${TAG}_pmc_valu.json", "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["valu_wave_insts_per_launch"] * kv[1]["launches"])[:14]:
    print("%-36s launches %4d  VALU wave-insts %14.0f  waves %9.0f per launch" % (k[:36], v["launches"], v["valu_wave_insts_per_launch"], v["waves_per_launch"]))
PY
      rm -rf $OUT/${TAG}_pmcv ;;
    prove) timeout 900 python tools/bench_prove.py 2>$OUT/${TAG}_prove.err | tail -1 | tee -a $OUT/${TAG}_bench_prove.json | cut -c1-1500; tail -3 $OUT/${TAG}_prove.err ;;
    prove:*) timeout 900 python tools/bench_prove.py ${step#prove:} 2>$OUT/${TAG}_prove.err | tail -1 | tee -a $OUT/${TAG}_bench_prove.json | cut -c1-1500; tail -3 $OUT/${TAG}_prove.err ;;
    provep|provep:*) A=""; [ "$step" != provep ] && A="${step#provep:}"
          ( cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_pprof -- \
              python $REPO/tools/bench_prove.py $A > $OUT/${TAG}_prove_prof.json 2> $OUT/${TAG}_pprof.log )
          S=$(find $OUT/${TAG}_pprof -name "*kernel_stats.csv" | head -1); [ -n "$S" ] && cp $S $OUT/${TAG}_prove_kernel_stats.csv && head -40 $S | cut -c1-170
          rm -rf $OUT/${TAG}_pprof; tail -1 $OUT/${TAG}_prove_prof.json | cut -c1-800 ;;
    msmp:*) N=$(echo "${step#msmp:}" | tr -c 'a-zA-Z0-9' '_')
          ( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_mprof -- \
              python $REPO/tools/bench_msm.py ${step#msmp:} > $OUT/${TAG}_msmp_$N.json 2> $OUT/${TAG}_mprof.log )
          S=$(find $OUT/${TAG}_mprof -name "*kernel_stats.csv" | head -1); [ -n "$S" ] && cp $S $OUT/${TAG}_msm_kernel_stats_$N.csv && head -24 $S | cut -c1-150
          rm -rf $OUT/${TAG}_mprof; tail -1 $OUT/${TAG}_msmp_$N.json ;;
    msm:*) timeout 600 python tools/bench_msm.py ${step#msm:} 2>&1 | tail -1 | tee -a $OUT/${TAG}_msm.json ;;
    nttp|nttp:*) A=""; [ "$step" != nttp ] && A="${step#nttp:}"
          ( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_nprof -- \
              python $REPO/tools/bench_ntt.py $A > $OUT/${TAG}_ntt_prof.json 2> $OUT/${TAG}_nprof.log )
          S=$(find $OUT/${TAG}_nprof -name "*kernel_stats.csv" | head -1); [ -n "$S" ] && cp $S $OUT/${TAG}_ntt_kernel_stats.csv && head -10 $S | cut -c1-170
          rm -rf $OUT/${TAG}_nprof; tail -1 $OUT/${TAG}_ntt_prof.json | cut -c1-600 ;;
    ntt) timeout 600 python tools/bench_ntt.py 2>&1 | tail -1 | tee -a $OUT/${TAG}_ntt.json | cut -c1-800 ;;
    ntt:*) timeout 600 python tools/bench_ntt.py ${step#ntt:} 2>&1 | tail -1 | tee -a $OUT/${TAG}_ntt.json | cut -c1-800 ;;
    run:*) timeout 1500 bash -c "${step#run:}" 2>&1 | tail -30 | tee -a $OUT/${TAG}_run.txt ;;
    *) echo "unknown step $step" ;;
  esac
done
