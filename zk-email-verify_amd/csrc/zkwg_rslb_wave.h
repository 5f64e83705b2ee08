// The Poseidon(2) merge chain of PoseidonModular (packages/circuits/utils/hash.circom:76-80) for 64 / ZK_RS_MERGE_LANES emails per
// wavefront: the body of the kernel zk_rslb_merge (zkwg_kernels_rslb.hip), in a header of its own so that the SAME code also
// compiles for the host on the simulated wavefront of tests/native/wavesim.h (the workgroup barrier of a one-wavefront
// workgroup is an exchange point) -- tests/native/wavetest.cpp, tests/test_soft_line_breaks.py.
#pragma once
#include "zkwg_fr.h"
#include "zkwg_sched.h"
#include "zkwg_poseidon_sparse.h"

#define ZK_RS_MERGE_LANES 4u   // lanes per email (3 state elements + 1 converter)
#if defined(ZKWG_WAVESIM)
#define ZK_RSM_LANE() zk_wavesim_lane()
#define ZK_RSM_SYNC() zk_wavesim_sync()
#define ZK_RSM_DEV inline
#else
#define ZK_RSM_LANE() (threadIdx.x)
#define ZK_RSM_SYNC() __syncthreads()
#define ZK_RSM_DEV __device__ __forceinline__
#endif

struct ZkRsMergeLds {
  Fr cm[(8 + 57) * 3];                      // additive constants in Montgomery form: c_first[4][3] | c_part[57][3] | c_last[4][3]
  Fr sh_y[64 / ZK_RS_MERGE_LANES][4];       // per email group: x5, x2, x4 of lane 0 (Montgomery)
  Fr sh_p[64 / ZK_RS_MERGE_LANES][ZK_RS_MERGE_LANES];   // per email group: one value per lane
};

// emails [block * NG, block * NG + NG) of a batch of n_emails; frv_all = the batch's field-element images (img_fr each);
// pos2 = the t = 3 table of zk_build_poseidon_sparse(3, 57)
ZK_RSM_DEV void zk_rslb_merge_wave(ZkRsMergeLds& S, const Fr* __restrict__ pos2, Fr* frv_all, u32 block, u32 n_emails, u32 img_fr, u32 rs_nch,
                                   u32 f_rs_chunk, u32 f_rs_hash) {
  constexpr u32 T = 3, RP = 57, GL = ZK_RS_MERGE_LANES, NG = 64u / GL;
  Fr* cm = S.cm;
  Fr (*sh_y)[4] = S.sh_y;
  Fr (*sh_p)[GL] = S.sh_p;
  const u32 lane = ZK_RSM_LANE(), g = lane / GL, j = lane % GL;
  const u32 e_raw = block * NG + g;
  const bool live = e_raw < n_emails;
  const u32 e = live ? e_raw : n_emails - 1u;      // idle groups shadow the last email (no stores)
  Fr* frv = frv_all + (u64)e * img_fr;
  const Fr* tab = pos2;
  const Fr* c_first = tab;
  const Fr* mt = tab + 4 * T;
  const Fr* c_part = mt + T * T;
  const Fr* s_part = c_part + RP * T;
  const Fr* bt = s_part + RP * (2 * T - 1);
  const Fr* c_last = bt + T * T;
  for (u32 i = lane; i < (8 + RP) * T; i += 64u) {
    const Fr c = i < 4 * T ? c_first[i] : (i < (4 + RP) * T ? c_part[i - 4 * T] : c_last[i - (4 + RP) * T]);
    cm[i] = fr_mont_mul(c, fr_R2());
  }
  ZK_RSM_SYNC();
  const Fr* cm_first = cm;
  const Fr* cm_part = cm + 4 * T;
  const Fr* cm_last = cm_part + RP * T;
  const Fr one = fr_from_u64(1);
  // lane 3 of the group converts the previous partial round's three S-box signals (x^2, x^4, x^5 of lane 0, Montgomery form)
  // to standard form for the image, one per step A, B, C
  Fr pend2 = fr_zero(), pend4 = fr_zero(), pend5 = fr_zero();
  u32 pend_idx = 0xffffffffu;     // image index of the pending round's `out` signal (then in2, in4)
  Fr* emit = frv;
  Fr x = fr_zero();
  // dense mix: x_i = sum_k x_k mat[k*T + i]
  auto dense = [&](const Fr* __restrict__ mat) {
    sh_p[g][j] = x;
    ZK_RSM_SYNC();
    if (j < T) {
      FrWide w;
      fr_wide_zero(w);
      for (u32 k = 0; k < T; ++k) fr_wide_mac(w, sh_p[g][k], mat[k * T + j]);
      x = fr_wide_redc(w);
    }
    ZK_RSM_SYNC();
  };
  const bool conv = j == T;
  Fr out_m = fr_mont_mul(frv[f_rs_chunk], fr_R2());      // _out, Montgomery form (all lanes hold it)
  for (u32 c = 1; c < rs_nch; ++c) {
    // state = [0, _out, chunk_hash_c]
    const Fr hc = fr_mont_mul(frv[f_rs_chunk + c], fr_R2());
    x = j == 1 ? out_m : (j == 2 ? hc : fr_zero());
    emit = frv + f_rs_hash + zk_rs_chunk_off(c) + ZK_P16_KEPT;
    for (u32 half = 0; half < 2; ++half) {
      if (half == 1) {
        for (u32 k = 0; k < RP; ++k) {
          const Fr* sk = s_part + k * (2 * T - 1);
          if (j < T) x = fr_add(x, cm_part[k * T + j]);
          const bool flush = conv && pend_idx != 0xffffffffu && live;
          // A: lane 0: x^2 | lanes 1, 2: u_j * s_j | lane 3: previous x^2 -> standard form (in2)
          const Fr r1 = fr_mont_mul(j < T ? x : pend2, j == 0 ? x : (j < T ? sk[j] : one));
          if (flush) emit[pend_idx + 1u] = r1;
          sh_p[g][j] = r1;
          // B: lane 0: x^4 | lane 3: previous x^4 -> standard form (in4)
          const Fr r2 = fr_mont_mul(conv ? pend4 : r1, conv ? one : r1);
          if (flush) emit[pend_idx + 2u] = r2;
          // C: lane 0: x^5 | lane 3: previous x^5 -> standard form (out)
          const Fr r3 = fr_mont_mul(conv ? pend5 : r2, conv ? one : x);
          if (flush) emit[pend_idx] = r3;
          if (j == 0) { sh_y[g][0] = r3; sh_y[g][1] = r1; sh_y[g][2] = r2; }
          ZK_RSM_SYNC();
          const Fr y0 = sh_y[g][0];
          if (conv) { pend5 = y0; pend2 = sh_y[g][1]; pend4 = sh_y[g][2]; pend_idx = 3u * (8u * T + k); }
          // D: lane 0: y0 * n00 | lanes 1, 2: y0 * w_j
          const Fr r4 = fr_mont_mul(y0, j == 0 ? sk[0] : (j < T ? sk[T - 1 + j] : one));
          if (j == 0) x = fr_add(fr_add(r4, sh_p[g][1]), sh_p[g][2]);
          else if (j < T) x = fr_add(x, r4);
          ZK_RSM_SYNC();
        }
        dense(bt);   // z = B u
      }
      for (u32 r = 0; r < 4; ++r) {
        const Fr* cc = (half ? cm_last : cm_first) + r * T;
        if (j < T) x = fr_add(x, cc[j]);
        const bool flush = conv && pend_idx != 0xffffffffu && live;
        const Fr x2 = fr_mont_mul(j < T ? x : pend2, j < T ? x : one);        // lane 3 flushes the last partial round's signals
        const Fr x4 = fr_mont_mul(conv ? pend4 : x2, conv ? one : x2);
        const Fr x5 = fr_mont_mul(conv ? pend5 : x4, conv ? one : x);
        if (flush) { emit[pend_idx + 1u] = x2; emit[pend_idx + 2u] = x4; emit[pend_idx] = x5; }
        if (conv) pend_idx = 0xffffffffu;
        const Fr e5 = fr_mont_mul(x5, one), e2 = fr_mont_mul(x2, one), e4 = fr_mont_mul(x4, one);
        if (j < T && live) {
          Fr* o = emit + 3u * ((half * 4u + r) * T + j);
          o[0] = e5; o[1] = e2; o[2] = e4;
        }
        if (j < T) x = x5;
        dense(mt);
      }
    }
    // out[0] of this permutation = lane 0's element: the next permutation's _out on lane 1
    sh_p[g][j] = x;
    ZK_RSM_SYNC();
    out_m = sh_p[g][0];
    ZK_RSM_SYNC();
  }
  const Fr r = fr_mont_mul(out_m, one);
  if (j == 0 && live) frv[f_rs_chunk] = r;
}
