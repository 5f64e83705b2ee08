// RemoveSoftLineBreaks(maxBody) witness values (template flag removeSoftLineBreaks,
// packages/circuits/email-verifier.circom:148-156, helpers/remove-soft-line-breaks.circom:14-126):
//   r = PoseidonModular(2*maxBody)(encoded || decoded)   (utils/hash.circom:49-82)
//   random-linear-combination sums sumEnc / sumDec over powers of r, final IsEqual.
// This is the one genuinely Fr-heavy block of the path (about 1.2 M field products per email at
// maxBody = 1536), and it is embarrassingly parallel in its first stage:
//   zk_rslb_chunks : one LANE per (email, 16-byte chunk): Poseidon(16), 612 S-box signals + the digest
//   zk_rslb_merge  : 8 lanes per email: the Poseidon(2) merge chain (inherently serial per email)
//   zk_rslb_scan   : one lane per email: the r-power scans and the final comparison
// The S-box signals go straight into the image's Fr area; zk_expand copies them into the witness.
#include "zkwg_dev.h"
#include "zkwg_kernels.h"
#include "zkwg_poseidon_sparse.h"
#include "zkwg_poseidon29.h"
#include "zkwg_rslb_wave.h"

// State in LDS as 9 x 29-bit limbs, limb-major (element j, limb l of lane t at st[(l * 17 + j) * 64 + t]: every access is one
// conflict-free 32-bit word per lane); the dense mixes copy it into registers.  The whole permutation runs in limb form
// (zkwg_poseidon29.h); only the 612 emitted S-box signals and the digest are packed into 4 x 64-bit words.
template <int V>
__device__ __forceinline__ void zk_rslb_chunks_body(const ZkSched& s, const ZkBufs& B) {
  __shared__ u32 st[9 * 17 * 64];
  const u32 lane = threadIdx.x;
  u64 unit = (u64)blockIdx.x * 64 + lane;
  if (B.rs_list) {                      // constant chunks apart (zk_rslb_classify): the lanes take the units that need hashing
    if (unit >= B.rs_cnt[0]) return;
    unit = B.rs_list[unit];
  }
  const u32 e = (u32)(unit / s.rs_nch), c = (u32)(unit % s.rs_nch);
  if (e >= B.n_emails) return;
  const u8* rec = B.in + (u64)e * s.in_stride;
  const u32 half = s.rs_nch / 2;   // chunks [0, half): encoded (= emailBody), [half, 2 half): decoded
  const uint4 raw = *(const uint4*)(rec + (c < half ? s.fr[1].in_data + 16u * c : s.in_off[11] + 16u * (c - half)));
  const u32 w[4] = {raw.x, raw.y, raw.z, raw.w};
  u32* stl = st + lane;
#pragma unroll
  for (u32 l = 0; l < 9; ++l) {
#pragma unroll
    for (u32 j = 0; j < 17; ++j) stl[(l * 17 + j) * 64] = (l == 0 && j > 0) ? ((w[(j - 1) >> 2] >> (8 * ((j - 1) & 3))) & 255u) : 0u;
  }
  Fr* frv = B.frv + (u64)e * s.img_fr;
  const Fr h = zk_poseidon29<17, V>(stl, 64, 17 * 64, B.pos16_l29, 68, frv + s.f_rs_hash + zk_rs_chunk_off(c), B.rs_stage + unit, (size_t)B.rs_units);
  frv[s.f_rs_chunk + c] = h;
}

// V = evaluator variant (zkwg_poseidon29.h): which of the four runs is ZKWG_RSLB_V's / the measured default's business (zkwg_api.hip)
#define ZK_RSLB_CHUNKS(V) \
  __global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void zk_rslb_chunks_v##V(ZkSched s, ZkBufs B) { zk_rslb_chunks_body<V>(s, B); }
ZK_RSLB_CHUNKS(0)
ZK_RSLB_CHUNKS(1)
ZK_RSLB_CHUNKS(2)
ZK_RSLB_CHUNKS(3)
ZK_RSLB_CHUNKS(4)
ZK_RSLB_CHUNKS(5)
ZK_RSLB_CHUNKS(6)
ZK_RSLB_CHUNKS(7)

// Constant chunks.  Both halves of the hashed string are zero-padded to maxBody (the encoded body behind its SHA padding, the decoded body
// behind its last byte), so a third of the chunks of a 1 KB body at maxBody = 1536 -- more for shorter bodies -- are sixteen zero bytes, and
// Poseidon(16)(0, ..., 0) is a constant of the circuit: its 612 S-box signals and its digest are computed once per handle (zkwg_api.hip, the
// same zk_poseidon29 on the host).  zk_rslb_classify splits a batch's units into the ones to hash and the constant ones (one global atomic
// per wavefront and list), zk_rslb_chunks runs on the first list with full wavefronts, zk_rslb_fill_const copies the table for the second
// (one wavefront per unit: 19.6 KB of coalesced stores).  Same image either way (tests/test_soft_line_breaks.py); ZKWG_RSLB_CONST_CHUNKS=0
// hashes every unit as before.
__global__ __launch_bounds__(256) void zk_rslb_classify(ZkSched s, ZkBufs B) {
  const u64 unit = (u64)blockIdx.x * 256 + threadIdx.x;
  const u32 e = (u32)(unit / s.rs_nch), c = (u32)(unit % s.rs_nch);
  const bool live = e < B.n_emails;
  bool zero = false;
  if (live) {
    const u8* rec = B.in + (u64)e * s.in_stride;
    const u32 half = s.rs_nch / 2;
    const uint4 raw = *(const uint4*)(rec + (c < half ? s.fr[1].in_data + 16u * c : s.in_off[11] + 16u * (c - half)));
    zero = (raw.x | raw.y | raw.z | raw.w) == 0u;
  }
  const u64 mz = __ballot(live && zero), mn = __ballot(live && !zero);
  const u32 lane = threadIdx.x & 63u;
  const u64 below = lane ? (~0ull >> (64u - lane)) : 0ull;
  u32 bn = 0, bz = 0;
  if (lane == 0) {
    if (mn) bn = atomicAdd(&B.rs_cnt[0], (u32)__popcll(mn));
    if (mz) bz = atomicAdd(&B.rs_cnt[1], (u32)__popcll(mz));
  }
  bn = __shfl(bn, 0); bz = __shfl(bz, 0);
  if (live && !zero) B.rs_list[bn + (u32)__popcll(mn & below)] = (u32)unit;
  if (live && zero) B.rs_list[B.rs_units + bz + (u32)__popcll(mz & below)] = (u32)unit;
}
__global__ __launch_bounds__(64) void zk_rslb_fill_const(ZkSched s, ZkBufs B) {
  const u32 n = B.rs_cnt[1];
  const uint4* src = (const uint4*)B.rs_zero;
  for (u32 k = blockIdx.x; k < n; k += gridDim.x) {
    const u32 unit = B.rs_list[B.rs_units + k];
    const u32 e = unit / s.rs_nch, c = unit % s.rs_nch;
    Fr* frv = B.frv + (u64)e * s.img_fr;
    uint4* dst = (uint4*)(frv + s.f_rs_hash + zk_rs_chunk_off(c));
    for (u32 i = threadIdx.x; i < 2u * ZK_P16_KEPT; i += 64u) dst[i] = src[i];
    if (threadIdx.x < 2u) ((uint4*)(frv + s.f_rs_chunk + c))[threadIdx.x] = src[2u * ZK_P16_KEPT + threadIdx.x];
  }
}

// The merge chain  _out = Poseidon(2)([_out, chunk_hash])  (utils/hash.circom:76-80) is inherently serial per email:
// rs_nch - 1 permutations (191 for maxBody = 1536), each 8 full + 57 partial rounds.  Round 3 ran it one LANE per email in
// standard form: ~8 dependent Montgomery products per round, 289 ms per 4,096 emails on 64 wavefronts.  A quarter-rate
// v_mad_u64_u32 costs a wavefront the same 16 cycles whether one lane or 64 need it, so the chain is run like
// zk_poseidon9_g16: 4 LANES per email (16 emails per wavefront), state in MONTGOMERY form on lanes 0..2 (the S-box is 3
// dependent products), every step ONE product executed by all lanes with lane-specific operands -- in a partial round lane 0
// computes x^2, x^4, x^5 while lanes 1, 2 form their first-row products and lane 3 converts the previous round's three S-box
// signals to standard form for the image.  A partial round is 4 dependent products, a full round 6 + the dense mix: ~0.27 k per
// permutation instead of ~0.5 k, and nothing but the chain itself is serial.  The result r goes to frv[f_rs_chunk] (the
// chunk digests are not witness signals and are dead once merged); zk_rslb_scan reads it.
__global__ __launch_bounds__(64) void zk_rslb_merge(ZkSched s, ZkBufs B) {
  __shared__ ZkRsMergeLds S;
  zk_rslb_merge_wave(S, B.pos2, B.frv, blockIdx.x, B.n_emails, s.img_fr, s.rs_nch, s.f_rs_chunk, s.f_rs_hash);
}

// Round 6: the same chain one LANE per email through the limb-form evaluator of the chunk hashes (zk_poseidon29<3>: S-boxes and mixes in
// 9 x 29-bit limbs, one reduction per dot product, nothing exchanged between lanes).  A permutation is ~0.6 k dependent product-sized
// steps instead of the four-lane version's ~0.3 k, but a wavefront carries 64 emails instead of 16 and a product is ~250 instructions
// instead of ~315: 2.5 x fewer instructions issued per email.  That is what counts beside zk_rslb_chunks and zk_expand, which leave the
// chain no idle issue slots; the chain's own latency is hidden by the ring of prepared batches either way.
// ZKWG_RSLB_MERGE_LANES=4 selects zk_rslb_merge.
__global__ __launch_bounds__(64) void zk_rslb_merge1(ZkSched s, ZkBufs B) {
  __shared__ u32 st[9 * 3 * 64];
  const u32 lane = threadIdx.x, e = blockIdx.x * 64 + lane;
  if (e >= B.n_emails) return;
  // 64 wavefronts per 4,096 emails, each a chain of 191 x 0.6 k dependent steps: the SIMD's arbiter serves it first (s_setprio), so that
  // its latency is its own and not three times that beside the throughput kernels' wavefronts -- which lose a 64th of the chip's issue slots
  if (B.rs_prio) __builtin_amdgcn_s_setprio(3);
  Fr* frv = B.frv + (u64)e * s.img_fr;
  u32* stl = st + lane;
  Fr out = frv[s.f_rs_chunk];
  for (u32 c = 1; c < s.rs_nch; ++c) {
    u32 a[9], b[9];
    zk_l29_from_fr(out, a);
    zk_l29_from_fr(frv[s.f_rs_chunk + c], b);
#pragma unroll
    for (u32 l = 0; l < 9; ++l) { stl[(l * 3 + 0) * 64] = 0u; stl[(l * 3 + 1) * 64] = a[l]; stl[(l * 3 + 2) * 64] = b[l]; }      // [0, _out, chunk_hash_c]
    out = zk_poseidon29<3, 2>(stl, 64, 3 * 64, B.pos2_l29, 57, frv + s.f_rs_hash + zk_rs_chunk_off(c) + ZK_P16_KEPT);
  }
  frv[s.f_rs_chunk] = out;
}

// the r-power scans (remove-soft-line-breaks.circom:72-126) and the final comparison: one lane per email, r from zk_rslb_merge
__global__ __launch_bounds__(64) void zk_rslb_scan(ZkSched s, ZkBufs B) {
  const u32 lane = threadIdx.x;
  const u32 e = blockIdx.x * 64 + lane;
  if (e >= B.n_emails) return;
  const u8* rec = B.in + (u64)e * s.in_stride;
  Fr* frv = B.frv + (u64)e * s.img_fr;
  const Fr r = frv[s.f_rs_chunk];
  const Fr rm = fr_to_mont(r);
  const u32 M = s.fr[1].max_bytes;
  const u8* enc = rec + s.fr[1].in_data;
  const u8* dec = rec + s.in_off[11];
  Fr* mux = frv + s.f_rs_mux;
  Fr* sum_enc = frv + s.f_rs_sum_enc;
  Fr* rdec = frv + s.f_rs_rdec;
  Fr* sum_dec = frv + s.f_rs_sum_dec;
  Fr rEnc = fr_zero(), sumEnc = fr_zero(), rDec = r, sumDec = fr_zero();
  // window of soft-break starts: sb0 = isSoftBreak[i], sb1 = [i-1], sb2 = [i-2]
  u32 sb1 = 0, sb2 = 0;
  u32 b0 = enc[0], b1 = enc[1], b2 = enc[2];
  for (u32 i = 0; i < M; ++i) {
    const u32 sb0 = (i + 2 < M) && b0 == 61u && b1 == 13u && b2 == 10u;
    const bool z = (sb0 | sb1 | sb2) != 0;                         // shouldZero[i] (:72-83)
    const Fr c0 = i ? fr_mont_mul(rEnc, rm) : r;                    // muxEnc[i].c[0] (:93-101)
    const Fr prev = i ? rEnc : fr_from_u64(1);
    rEnc = z ? prev : c0;
    if (i) mux[2 * i - 1] = c0;
    mux[2 * i] = rEnc;
    const u32 proc = z ? 0u : b0;                                   // processed[i] (:86-88)
    if (proc) sumEnc = fr_add(sumEnc, fr_mont_mul(rEnc, fr_to_mont(fr_from_u64(proc))));
    sum_enc[i] = sumEnc;
    if (i) { rDec = fr_mont_mul(rDec, rm); rdec[i - 1] = rDec; }
    const u32 d = dec[i];
    if (d) sumDec = fr_add(sumDec, fr_mont_mul(rDec, fr_to_mont(fr_from_u64(d))));
    sum_dec[i] = sumDec;
    sb2 = sb1; sb1 = sb0;
    b0 = b1; b1 = b2; b2 = (i + 3 < M) ? enc[i + 3] : 0u;
  }
  // isValid <== IsEqual()([sumEnc[M-1], sumDec[M-1]]); qpEncodingChecker.isValid === 1
  const Fr diff = fr_sub(sumDec, sumEnc);
  const bool valid = fr_is_zero(diff);
  frv[s.f_rs_final] = fr_from_u64(valid ? 1 : 0);
  frv[s.f_rs_final + 1] = valid ? fr_zero() : fr_from_mont(fr_mont_inv(fr_to_mont(diff)));
  if (!valid) B.status[e] = 4;
}
