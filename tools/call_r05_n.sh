set -x
# zk_rslb_chunks: the dense mixes through the staging area (variants 4-7: ~110-130 registers instead of 446, so that zk_expand keeps
# its wavefronts beside it) against variant 2 of call r05_m; parity tests with variant 6 first
bash tools/gpu_call.sh r05_n env:ZKWG_RSLB_V=6 files:tests/test_soft_line_breaks.py \
  env:ZKWG_RSLB_V=2 rslb:v2 env:ZKWG_RSLB_V=4 rslb:v4 env:ZKWG_RSLB_V=6 rslb:v6 env:ZKWG_RSLB_V=7 rslb:v7 env:ZKWG_RSLB_V=5 rslb:v5
