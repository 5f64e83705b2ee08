set -x
# zk_rslb_chunks in 29-bit limb form (zkwg_poseidon29.h): parity tests with the default and the most different variant, then the
# removeSoftLineBreaks pipeline with each of the four evaluator variants
bash tools/gpu_call.sh r05_m files:tests/test_soft_line_breaks.py env:ZKWG_RSLB_V=3 files:tests/test_soft_line_breaks.py \
  env:ZKWG_RSLB_V=0 rslb:v0 env:ZKWG_RSLB_V=1 rslb:v1 env:ZKWG_RSLB_V=2 rslb:v2 env:ZKWG_RSLB_V=3 rslb:v3
