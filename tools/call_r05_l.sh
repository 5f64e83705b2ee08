set -x
OUT=gpurun_out; mkdir -p $OUT
T=$PWD/zk-email-verify_amd/data/templates/zk-regex-circom/circuits/common/body_hash_regex.circom
U=$PWD/tests/golden/regex_style/body_hash_regex_unshared.circom
bash tools/gpu_call.sh r05_l files:tests/test_regex_template.py benchq "benchq:--regex $T" "benchq:--regex $U" env:ZKWG_NET_DENSE_OFF=1 "benchq:--regex $T" "benchq:--regex $U"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r05_l_benchq.json").read().strip().splitlines()[-1]); r=d["roofline"]; print("builtin", d["value"], r["avg_launch_ms"], r.get("alone_launch_ms"))
for l in open("gpurun_out/r05_l_benchq_variants.json"):
    d=json.loads(l); r=d["roofline"]; print(d["value"], r["frac"], r["avg_launch_ms"], r.get("alone_launch_ms"), r.get("alone_frac"))
PY
