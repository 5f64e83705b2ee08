set -x
# removeSoftLineBreaks: what the kernels take ALONE (no overlap, chain on the caller's stream) beside the overlapped pipeline, and the
# pipeline's tile / prepare sub-batch sizes
export ZKWG_RSLB_V=6
bash tools/gpu_call.sh r05_p rslb:v6 "env:RSLB_ARGS=--no-overlap 1" env:ZKWG_RSLB_SYNC=1 rslb:v6_alone env:ZKWG_RSLB_V=2 rslb:v2_alone \
  env:ZKWG_RSLB_V=6 env:ZKWG_RSLB_SYNC=0 env:RSLB_ARGS= env:RSLB_TILE=512 rslb:v6_tile512 env:RSLB_TILE=256 env:RSLB_PREP=2048 rslb:v6_prep2048 \
  env:RSLB_PREP=4096 env:RSLB_RING=6 rslb:v6_ring6
