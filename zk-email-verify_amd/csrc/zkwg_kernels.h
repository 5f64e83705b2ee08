// Kernel declarations (definitions in zkwg_kernels_*.hip).
#pragma once
#include <hip/hip_runtime.h>
#include "zkwg_sched.h"
#include "zkwg_o0.h"

__global__ void zk_sha_chain(ZkSched s, ZkBufs B);   // zkwg_kernels_sha.hip
__global__ void zk_sha_trace(ZkSched s, ZkBufs B);   // zkwg_kernels_sha.hip
__global__ void zk_rsa(ZkSched s, ZkBufs B);         // zkwg_kernels_rsa.hip
__global__ void zk_fpmul_small(ZkSched s, ZkBufs B);
__global__ void zk_poseidon9(ZkSched s, ZkBufs B);   // zkwg_kernels_rsa.hip
__global__ void zk_poseidon9_wave(ZkSched s, ZkBufs B);
__global__ void zk_poseidon9_g16(ZkSched s, ZkBufs B);
__global__ void zk_misc_ev(ZkSched s, ZkBufs B);     // zkwg_kernels_misc.hip
__global__ void zk_net_eval(ZkSched s, ZkBufs B);    // zkwg_kernels_net.hip
__global__ void zk_net_scan(ZkSched s, ZkBufs B);
__global__ void zk_rslb_chunks_v0(ZkSched s, ZkBufs B); // zkwg_kernels_rslb.hip (v0..v3: evaluator variants, zkwg_poseidon29.h)
__global__ void zk_rslb_chunks_v1(ZkSched s, ZkBufs B);
__global__ void zk_rslb_chunks_v2(ZkSched s, ZkBufs B);
__global__ void zk_rslb_chunks_v3(ZkSched s, ZkBufs B);
__global__ void zk_rslb_chunks_v4(ZkSched s, ZkBufs B);
__global__ void zk_rslb_chunks_v5(ZkSched s, ZkBufs B);
__global__ void zk_rslb_chunks_v6(ZkSched s, ZkBufs B);
__global__ void zk_rslb_chunks_v7(ZkSched s, ZkBufs B);
#include "zkwg_rslb_wave.h"   // ZK_RS_MERGE_LANES
__global__ void zk_rslb_merge(ZkSched s, ZkBufs B);
__global__ void zk_rslb_merge1(ZkSched s, ZkBufs B);     // the same chain, one lane per email in limb form
__global__ void zk_rslb_classify(ZkSched s, ZkBufs B);   // constant chunks: the units to hash | the all-zero ones
__global__ void zk_rslb_fill_const(ZkSched s, ZkBufs B);
__global__ void zk_rslb_scan(ZkSched s, ZkBufs B);
__global__ void zk_r1cs_check(const u64* row_ptr, const u32* wire, const Fr* coef, const u8* kind, u32 m,
                              const u8* wit, u64 stride, unsigned long long* first_bad);  // zkwg_kernels_r1cs.hip
__global__ void zk_r1cs_eval(const u64* row_ptr, const u32* wire, const Fr* coef, const u8* kind, u32 m, const u8* wit, u64 stride,
                             u8* out, u64 out_stride, int mont);  // zkwg_kernels_r1cs.hip
// zkwg_kernels_expand3.hip: one piece of 256 K slots per workgroup (K = 1, 2, 4), standard / Montgomery form, kept-v1 / numbered circuits
#define ZK_X3_DECL(K) \
  __global__ void zk_expand3_k##K(ZkX3 A); __global__ void zk_expand3_mont_k##K(ZkX3 A); \
  __global__ void zk_expand3_o0_k##K(ZkX3 A, ZkO0Dev O); __global__ void zk_expand3_o0_mont_k##K(ZkX3 A, ZkO0Dev O); \
  __global__ void zk_expand3_o0p_k##K(ZkX3 A, ZkO0Dev O); __global__ void zk_expand3_o0p_mont_k##K(ZkX3 A, ZkO0Dev O);
ZK_X3_DECL(1) ZK_X3_DECL(2) ZK_X3_DECL(4)
__global__ void zk_expand3_k8(ZkX3 A); __global__ void zk_expand3_mont_k8(ZkX3 A);
__global__ void zk_image_to_mont(ZkX3 A);
__global__ void zk_expand3_o0b_k1(ZkX3 A, ZkO0Dev O); __global__ void zk_expand3_o0b_mont_k1(ZkX3 A, ZkO0Dev O);
__global__ void zk_expand3_o0c_k1(ZkX3 A, ZkO0Dev O); __global__ void zk_expand3_o0c_mont_k1(ZkX3 A, ZkO0Dev O);
__global__ void zk_o0_rows_small(ZkX3 A, ZkO0Dev O);
__global__ void zk_o0_chains_small(ZkX3 A, ZkO0Dev O);
__global__ void zk_o0_generic(ZkX3 A, ZkO0Dev O);
__global__ void zk_o0_rows_small_long(ZkX3 A, ZkO0Dev O);
#define ZK_ROW_EMAILS 8   // emails per thread of zk_o0_rows_small
__global__ void zk_o0_rows_fr(ZkX3 A, ZkO0Dev O);
__global__ void zk_mont_convert(Fr* v, u64 n, int to_mont);  // zkwg_kernels_handoff.hip
__global__ void zk_gen_inputs(ZkSched s, ZkDkimBatch D, u8* recs, int* gen_status, u32 n);  // zkwg_kernels_inputs.hip
