set -x
# removeSoftLineBreaks: the prepare kernels (chunk hashes) and zk_expand on disjoint sets of compute units
export ZKWG_RSLB_V=6
bash tools/gpu_call.sh r05_q "env:RSLB_ARGS=--prep-cus 128 --prep-cu-stride 2" rslb:v6_cus128s2 "env:RSLB_ARGS=--prep-cus 128" rslb:v6_cus128 \
  "env:RSLB_ARGS=--prep-cus 144" rslb:v6_cus144 "env:RSLB_ARGS=--prep-cus 112" rslb:v6_cus112 "env:RSLB_ARGS=--prep-cus 160" rslb:v6_cus160
