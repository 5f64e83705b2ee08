set -x
# zk_rslb_chunks after the three-way additions and the two-outputs-per-pass staged mixes: variants 2 (registers) and 4-7 (staged)
bash tools/gpu_call.sh r05_o env:ZKWG_RSLB_V=6 files:tests/test_soft_line_breaks.py \
  env:ZKWG_RSLB_V=2 rslb:v2 env:ZKWG_RSLB_V=6 rslb:v6 env:ZKWG_RSLB_V=4 rslb:v4 env:ZKWG_RSLB_V=7 rslb:v7
