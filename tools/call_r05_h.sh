set -x
OUT=gpurun_out; mkdir -p $OUT
timeout 120 tools/vmm_probe 2>&1 | tee $OUT/r05_h_vmm_probe.txt
timeout 300 tools/chunkbench 200 1 58 2>&1 | tee $OUT/r05_h_chunkbench.txt | tail -8
timeout 900 python -m pytest tests/test_multi.py -m gpu -x -q 2>&1 | tail -6 | tee $OUT/r05_h_multi_tests.txt
bash tools/gpu_call.sh r05_h benchq "benchq:--place-ring 1" "benchq:--place-ring 0"
python - <<'PY'
import json
for f in ("gpurun_out/r05_h_benchq.json",):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(d.get("ring_placement"))
PY
