// zk_expand -- the streaming expansion kernel (the HBM-write-bound kernel of the path).
//
// Every witness element is a 32-byte little-endian field element, but >95 % of the
// EmailVerifier witness is single bits and nearly all the rest fits 128 bits.  The
// compute kernels therefore leave a compact per-email IMAGE (packed bit groups, small
// integers, a few thousand genuine field elements), and this one kernel expands image
// + input record into the final `.wtns` data section, guided by the circuit's static
// segment table (zkwg_sched.h).  One workgroup writes one contiguous 32 KiB portion
// (ZK_PORTION slots) of one witness; consecutive workgroups write consecutive portions,
// so at any instant the chip writes one dense moving window of HBM (fill-like DRAM
// page locality), 1 KiB of contiguous bytes per wavefront store instruction, each
// witness byte written exactly once.
//
// Reference semantic being produced: the witness vector of
// `circuit.calculateWitness(input)` (packages/circuits/tests/email-verifier.test.ts:43),
// i.e. section 2 of the `.wtns` file.
#include "zkwg_dev.h"
#include "zkwg_kernels.h"
#include "zkwg_bh_dfa.h"

// WAVE_MODE = false: one workgroup per (portion, email group), threads interleaved over the portion.
// WAVE_MODE = true : one WAVEFRONT per (portion, email group): every wave writes its own contiguous
//                    portion (1 KiB per store instruction, back to back), 4 such waves per workgroup.
#define ZK_FOR_CHUNKS(c) \
  for (u32 c = MONT ? 2u * tid : tid; c < nch; c = MONT ? ((c & 1u) ? c + 2u * ZK_EXPAND_THREADS - 1u : c + 1u) : c + ZK_EXPAND_THREADS)
#define ZK_STORE(c, v) do { if constexpr (MONT) zk_mont_put(dst, c, v, vkeep, rtab, pm); else dst[c] = (v); } while (0)

// the rare general case (a genuine field element: inverses, v_ab, carries ...): one Montgomery product;
// kept out of line so that the 20 store sites of the kernel stay small
__device__ __noinline__ void zk_mont_general(uint4* __restrict__ dst2, uint4 a, uint4 b) {
  const Fr x{{(u64)a.x | ((u64)a.y << 32), (u64)a.z | ((u64)a.w << 32), (u64)b.x | ((u64)b.y << 32), (u64)b.z | ((u64)b.w << 32)}};
  const Fr m = fr_to_mont(x);
  dst2[0] = make_uint4((u32)m.l[0], (u32)(m.l[0] >> 32), (u32)m.l[1], (u32)(m.l[1] >> 32));
  dst2[1] = make_uint4((u32)m.l[2], (u32)(m.l[2] >> 32), (u32)m.l[3], (u32)(m.l[3] >> 32));
}
// Montgomery-form output of one slot: `lo` / `hi` are its two standard-form halves
// (`pm`: both halves already are Montgomery form -- they came from the Montgomery-form inverse table)
__device__ __forceinline__ void zk_mont_put(uint4* __restrict__ dst, u32 c, const uint4& v, uint4& vkeep,
                                            const uint4* __restrict__ rtab, bool& pm) {
  if (!(c & 1u)) { vkeep = v; return; }
  const uint4 a = vkeep, b = v;
  if (pm) { pm = false; dst[c - 1] = a; dst[c] = b; return; }
  uint4 v0, v1;
  if ((b.x | b.y | b.z | b.w | a.y | a.z | a.w) == 0u && a.x < 65536u) {
    if (a.x == 0u) { v0 = zk_zero4(); v1 = zk_zero4(); }
    else if (a.x == 1u) {
      const Fr Rm = fr_R();
      v0 = make_uint4((u32)Rm.l[0], (u32)(Rm.l[0] >> 32), (u32)Rm.l[1], (u32)(Rm.l[1] >> 32));
      v1 = make_uint4((u32)Rm.l[2], (u32)(Rm.l[2] >> 32), (u32)Rm.l[3], (u32)(Rm.l[3] >> 32));
    } else { v0 = rtab[2 * a.x]; v1 = rtab[2 * a.x + 1]; }
  } else {
    zk_mont_general(dst + (c - 1), a, b);
    return;
  }
  dst[c - 1] = v0;
  dst[c] = v1;
}
//
// MONT = true (prover hand-off): the identical segment code, but a thread produces both
// halves of a slot and writes them as x * 2^256 mod r: 0 -> 0, 1 -> R (constants), values below 2^16
// from a table of v * R, anything else one Montgomery product by R^2.  Still one write per witness byte and no read of the witness: a device prover (MSM /
// NTT kernels want Montgomery scalars) costs no extra HBM pass.
template <int ZK_BLOCK_THREADS, bool WAVE_MODE, bool MONT = false>
__device__ __forceinline__ void zk_expand_body(const ZkSched& s, const ZkBufs& B) {
  constexpr u32 ZK_EXPAND_THREADS = WAVE_MODE ? 64u : (u32)ZK_BLOCK_THREADS;
  constexpr u32 UNITS_PER_BLOCK = WAVE_MODE ? (u32)ZK_BLOCK_THREADS / 64u : 1u;
  // Workgroups are dealt to the 8 XCDs round-robin (blockIdx % 8).  Give every XCD a contiguous run of units:
  // xcd_remap = 1: one run per XCD over the whole launch; K > 1: runs of K workgroups inside groups of 8 K.
  u32 blk = blockIdx.x;
  if (B.xcd_remap == 1u) {
    const u32 per = gridDim.x >> 3;
    if (blk < per * 8u) blk = (blk & 7u) * per + (blk >> 3);
  } else if (B.xcd_remap > 1u) {
    const u32 K = B.xcd_remap, G = 8u * K;
    const u32 g = blk / G, r = blk - g * G;
    if ((g + 1u) * G <= gridDim.x) blk = g * G + (r & 7u) * K + (r >> 3);
  }
  const u32 unit = blk * UNITS_PER_BLOCK + (WAVE_MODE ? (threadIdx.x >> 6) : 0u);
  // workgroup (p, g): portion p of the emails [g*E, g*E+E) of this launch.  The segment lookup is
  // email-independent, so its latency is paid once per workgroup and amortised over E emails.
  const u32 p = unit % s.nportions;
  const u32 g = unit / s.nportions;
  const u32 E = B.emails_per_wg;
  const u32 el0 = g * E;                                   // first email (launch-local index)
  const u32 el1 = min(el0 + E, B.n_emails - B.e_first);    // one past the last
  if (el0 >= el1) return;
  const u64 slot0 = (u64)p * s.portion;
  const u64 slot1 = min(s.W, slot0 + s.portion);
  // Montgomery output: inverses of small integers come from a Montgomery-form copy of the table
  const uint4* __restrict__ invtab = (const uint4*)(MONT ? B.invtab_m : B.invtab);
  bool pm = false;
  (void)pm;
  const u32 tid = WAVE_MODE ? (threadIdx.x & 63u) : threadIdx.x;

  // Montgomery output: a thread handles the two 16-byte chunks of a slot back to back (ZK_FOR_CHUNKS), keeps the
  // low half in `vkeep` and converts + stores the pair when the high half arrives (zk_mont_put)
  uint4 vkeep = zk_zero4();
  const uint4* __restrict__ rtab = (const uint4*)B.rtab;      // v * R mod r for v < 65536
  (void)vkeep; (void)rtab;
  for (u32 si = B.first_seg[p]; si < s.nsegs; ++si) {
    const ZkSeg sg = B.segs[si];
    if (sg.slot >= slot1) break;
    const u64 lo = max(sg.slot, slot0);
    const u64 hi = min(sg.slot + sg.nslots, slot1);
    const u32 r0 = (u32)(lo - sg.slot) + sg.r0;  // first element of the logical array handled here
    const u32 nch = (u32)(hi - lo) * 2;          // chunks to write
   for (u32 el = el0; el < el1; ++el) {
    const u32 e = el + B.e_first;                // email index inside the prepared batch
    uint4* __restrict__ dst = B.wit + (u64)el * B.wit_stride16 + lo * 2;
    const u8* __restrict__ rec = B.in + (u64)e * s.in_stride;
    const u64* __restrict__ bits = B.bits + (u64)e * s.img_bits;
    const u32* __restrict__ small = B.small + (u64)e * s.img_small;
    const Fr* __restrict__ frv = B.frv + (u64)e * s.img_fr;

    switch (sg.type) {
      case ZSEG_SMALL:
        ZK_FOR_CHUNKS(c) {
          uint4 v = zk_zero4();
          if (!(c & 1u)) v.x = small[sg.src + r0 + (c >> 1)];
          ZK_STORE(c, v);
        }
        break;
      case ZSEG_FR: {
        const uint4* src = (const uint4*)(frv + sg.src + r0);
        ZK_FOR_CHUNKS(c) { const uint4 v = src[c]; ZK_STORE(c, v); }
        break;
      }
      case ZSEG_BITS:
        ZK_FOR_CHUNKS(c) {
          uint4 v = zk_zero4();
          if (!(c & 1u)) {
            u32 r = r0 + (c >> 1);
            u32 g = r / sg.a, bit = r - g * sg.a;
            v.x = (u32)(bits[sg.src + g * sg.b + (bit >> 6)] >> (bit & 63)) & 1u;
          }
          ZK_STORE(c, v);
        }
        break;
      case ZSEG_SHA_SP:
        ZK_FOR_CHUNKS(c) {
          uint4 v = zk_zero4();
          if (!(c & 1u)) {
            u32 r = r0 + (c >> 1);
            u32 i = r / ZK_SP_SLOTS, q = r - i * ZK_SP_SLOTS;
            u32 sub = min(q >> 5, 4u);
            v.x = (u32)(bits[sg.src + i * 5 + sub] >> (q - sub * 32)) & 1u;
          }
          ZK_STORE(c, v);
        }
        break;
      case ZSEG_SHA_T1:
        ZK_FOR_CHUNKS(c) {
          uint4 v = zk_zero4();
          if (!(c & 1u)) {
            u32 r = r0 + (c >> 1);
            u32 i = r / ZK_T1_SLOTS, q = r - i * ZK_T1_SLOTS;
            u32 sub = min(q >> 5, 3u);
            v.x = (u32)(bits[sg.src + i * 4 + sub] >> (q - sub * 32)) & 1u;
          }
          ZK_STORE(c, v);
        }
        break;
      case ZSEG_SHA_T2:
        ZK_FOR_CHUNKS(c) {
          uint4 v = zk_zero4();
          if (!(c & 1u)) {
            u32 r = r0 + (c >> 1);
            u32 i = r / ZK_T2_SLOTS, q = r - i * ZK_T2_SLOTS;
            u32 sub = min(q >> 5, 4u);
            v.x = (u32)(bits[sg.src + i * 5 + sub] >> (q - sub * 32)) & 1u;
          }
          ZK_STORE(c, v);
        }
        break;
      case ZSEG_ISZ: {
        const int half_tab = (int)s.inv_half;
        ZK_FOR_CHUNKS(c) {
          u32 r = r0 + (c >> 1);
          int d = (int)small[sg.src + (r >> 1)];
          uint4 v = zk_zero4();
          if (!(r & 1u)) {
            if (!(c & 1u)) v.x = (d == 0);
          } else {
            d = max(-half_tab, min(half_tab, d));
            pm = MONT; v = invtab[(u32)(d + half_tab) * 2 + (c & 1u)];
          }
          ZK_STORE(c, v);
        }
        break;
      }
      case ZSEG_SEL: {
        // 256 x ItemAtIndex(NB): per output bit k: nums[NB], then NB x (isz.out, isz.inv)
        const u32 NB = sg.a, per = 3 * NB;
        const int idx = (int)small[sg.src];
        const int half_tab = (int)s.inv_half;
        ZK_FOR_CHUNKS(c) {
          u32 r = r0 + (c >> 1), hf = c & 1u;
          u32 k = r / per, q = r - k * per;
          uint4 v = zk_zero4();
          if (q < NB) {
            if (!hf && (int)q == idx) v.x = (small[sg.b + (k >> 5)] >> (31 - (k & 31))) & 1u;
          } else {
            u32 t = q - NB, j = t >> 1;
            if (!(t & 1u)) {
              if (!hf) v.x = ((int)j == idx);
            } else {
              int d = idx - (int)j;  // isz.in = index - j
              d = max(-half_tab, min(half_tab, d));
              pm = MONT; v = invtab[(u32)(d + half_tab) * 2 + hf];
            }
          }
          ZK_STORE(c, v);
        }
        break;
      }
      case ZSEG_IN8:
        ZK_FOR_CHUNKS(c) {
          uint4 v = zk_zero4();
          if (!(c & 1u)) v.x = rec[sg.src + r0 + (c >> 1)];
          ZK_STORE(c, v);
        }
        break;
      case ZSEG_IN8MASK:
        ZK_FOR_CHUNKS(c) {
          uint4 v = zk_zero4();
          if (!(c & 1u)) { const u32 r = r0 + (c >> 1); v.x = (u32)rec[sg.src + r] * (u32)rec[sg.a + r]; }
          ZK_STORE(c, v);
        }
        break;
      case ZSEG_IN8BITS:
        ZK_FOR_CHUNKS(c) {
          uint4 v = zk_zero4();
          if (!(c & 1u)) {
            u32 r = r0 + (c >> 1);
            v.x = (u32)(rec[sg.src + (r >> 3)] >> (r & 7)) & 1u;
          }
          ZK_STORE(c, v);
        }
        break;
      case ZSEG_LIMB:
        ZK_FOR_CHUNKS(c) {
          uint4 v = zk_zero4();
          if (!(c & 1u)) v = *(const uint4*)(rec + sg.src + 16 * (r0 + (c >> 1)));
          ZK_STORE(c, v);
        }
        break;
      case ZSEG_LTBITS: {
        const long long base = (long long)(int)small[sg.src] + (1ll << sg.a);
        const u32 per = sg.a + 1;
        ZK_FOR_CHUNKS(c) {
          uint4 v = zk_zero4();
          if (!(c & 1u)) {
            u32 r = r0 + (c >> 1);
            u32 i = r / per, bit = r - i * per;
            v.x = (u32)((u64)(base - (long long)i) >> bit) & 1u;
          }
          ZK_STORE(c, v);
        }
        break;
      }
      case ZSEG_REGSEL: {
        // SelectRegexReveal (utils/regex.circom:31-37): per index i: IsEqual(i,start) (out,inv),
        // IsZero(in[i]) (out,inv), [i>0: IsZero(in[i-1]) (out,inv)], GreaterThan(bl)(i, start+43) bits
        const u32 bl = sg.a, per = 6 + bl + 1, N = sg.c;
        const int start = (int)small[sg.src];
        const int half_tab = (int)s.inv_half;
        ZK_FOR_CHUNKS(c) {
          u32 r = r0 + (c >> 1), hf = c & 1u;
          u32 i, q;
          if (r < per - 2) {  // index 0 has no "previous" IsZero
            i = 0; q = r < 4 ? r : r + 2;
          } else {
            u32 rr = r - (per - 2);
            i = 1 + rr / per; q = rr - (i - 1) * per;
          }
          uint4 v = zk_zero4();
          if (q < 6) {
            int d;
            if (q < 2) d = start - (int)i;                       // isz.in = in[1] - in[0] = startIndex - i
            else if (q < 4) d = (int)small[sg.b + (i < N ? i : N - 1)];
            else d = (int)small[sg.b + (i ? i - 1 : 0)];
            if (!(q & 1u)) {
              if (!hf) v.x = (d == 0);
            } else {
              d = max(-half_tab, min(half_tab, d));
              pm = MONT; v = invtab[(u32)(d + half_tab) * 2 + hf];
            }
          } else if (!hf) {
            long long val = (long long)start + 43 + (1ll << bl) - (long long)i;
            v.x = (u32)((u64)val >> (q - 6)) & 1u;
          }
          ZK_STORE(c, v);
        }
        break;
      }
      case ZSEG_VSHIFT: {
        const u32 N = sg.a;
        const u32 shift = small[sg.src];
        ZK_FOR_CHUNKS(c) {
          uint4 v = zk_zero4();
          if (!(c & 1u)) {
            u32 r = r0 + (c >> 1);
            u32 j = r / N, i = r - j * N;
            u32 sh = shift & ((2u << j) - 1u);
            v.x = small[sg.b + (i + sh) % N];
          }
          ZK_STORE(c, v);
        }
        break;
      }
      case ZSEG_B64BITS:
      case ZSEG_B64: {
        const int half_tab = (int)s.inv_half;
        const u32 per = sg.type == ZSEG_B64 ? 68u : 6u;
        ZK_FOR_CHUNKS(c) {
          u32 r = r0 + (c >> 1), hf = c & 1u;
          u32 g = r / per, q = r - g * per;
          const int ch = (int)small[sg.src + g];
          // lib/base64.circom:71-128
          const u32 rAZ = (ch >= 65 && ch <= 90), raz = (ch >= 97 && ch <= 122), r09 = (ch >= 48 && ch <= 57);
          const u32 sAZ = rAZ * (u32)(ch - 65);
          const u32 saz = sAZ + raz * (u32)(ch - 71);
          const u32 s09 = saz + r09 * (u32)(ch + 4);
          const u32 spl = s09 + (ch == 43) * (u32)(ch + 19);
          const u32 ssl = spl + (ch == 47) * (u32)(ch + 16);
          uint4 v = zk_zero4();
          if (sg.type == ZSEG_B64BITS) {
            if (!hf) v.x = (ssl >> q) & 1u;
          } else if (q < 8) {
            const u32 mids[8] = {rAZ, sAZ, raz, saz, r09, s09, spl, ssl};
            if (!hf) v.x = mids[q];
          } else if (q < 62) {
            u32 k = (q - 8) / 9, bit = (q - 8) - k * 9;
            // le_Z: in+256-91, ge_A: 64+256-in, le_z: in+256-123, ge_a: 96+256-in, le_9: in+256-58, ge_0: 47+256-in
            const int vals[6] = {ch + 256 - 91, 64 + 256 - ch, ch + 256 - 123, 96 + 256 - ch, ch + 256 - 58, 47 + 256 - ch};
            if (!hf) v.x = ((u32)vals[k] >> bit) & 1u;
          } else {
            u32 k = (q - 62) >> 1;
            int d = ch - (k == 0 ? 43 : (k == 1 ? 47 : 61));
            if (!((q - 62) & 1u)) {
              if (!hf) v.x = (d == 0);
            } else {
              d = max(-half_tab, min(half_tab, d));
              pm = MONT; v = invtab[(u32)(d + half_tab) * 2 + hf];
            }
          }
          ZK_STORE(c, v);
        }
        break;
      }
      case ZSEG_DFA: {
        // BodyHashRegex DFA circuit arrays (zkwg_layout.h zk_walk_bh_regex): one entry per position i of
        // in[] = [255, header...].  zk_misc_ev left one word per position: in | st<<8 | nx<<16 | st_next<<24
        // (nx = the transition out of a non-zero state, 255 = none) plus the class / primitive truth masks.
        const u32* __restrict__ pos = small + sg.src;
        const u32* __restrict__ cmask = small + s.m_dfa_cm;
        const u32* __restrict__ pmask = small + s.m_dfa_pm;
        const int half_tab = (int)s.inv_half;
        const u32 pb = sg.b, pc = sg.c;
        // PER = kept slots per position (compile time); f(i, q, word) -> small value (high half is zero)
#define ZK_DFA_LOOP(PER, IDX_OFF, EXPR)                                                   \
        ZK_FOR_CHUNKS(c) {                              \
          uint4 v = zk_zero4();                                                           \
          if (!(c & 1u)) {                                                                \
            const u32 r = r0 + (c >> 1);                                                  \
            const u32 i = r / (PER), q = r - i * (PER);                                   \
            const u32 w0 = pos[i + (IDX_OFF)];                                            \
            const u32 b = w0 & 255u, st = (w0 >> 8) & 255u, nx = (w0 >> 16) & 255u, sn = w0 >> 24; \
            (void)b; (void)st; (void)nx; (void)sn; (void)q;                               \
            v.x = (EXPR);                                                                 \
          }                                                                               \
          ZK_STORE(c, v);                                                                     \
        }
        switch (sg.a) {
          case ZDFA_EQ:
            ZK_FOR_CHUNKS(c) {
              const u32 r = r0 + (c >> 1), hf = c & 1u;
              const u32 i = r >> 1;
              int d = (int)pb - (int)(pos[i] & 255u);          // isz.in = in[1] - in[0] = ch - in[i]
              uint4 v = zk_zero4();
              if (!(r & 1u)) { if (!hf) v.x = (d == 0); }
              else { d = max(-half_tab, min(half_tab, d)); pm = MONT; v = invtab[(u32)(d + half_tab) * 2 + hf]; }
              ZK_STORE(c, v);
            }
            break;
          case ZDFA_LT: ZK_DFA_LOOP(9u, 0u, ((pc ? pb + b : pb - b) >> q) & 1u) break;
          case ZDFA_RNG: ZK_DFA_LOOP(1u, 0u, (b >= pb && b <= pc)) break;
          case ZDFA_CLS: ZK_DFA_LOOP(2u, 0u, (q == 0) ^ (__builtin_popcount(pmask[i] & pc) != 0)) break;
          case ZDFA_AND: ZK_DFA_LOOP(1u, 0u, (pb ? (st == pb) : (nx == 255u)) & ((cmask[i] >> pc) & 1u)) break;
          case ZDFA_TMP: ZK_DFA_LOOP(2u, 0u, (q == 0) ^ (nx == pb)) break;
          case ZDFA_FZE: ZK_DFA_LOOP(2u, 0u, (q == 0) ^ (nx != 255u)) break;
          case ZDFA_ST: ZK_DFA_LOOP(2u, 0u, (q == 0) ^ (sn == pb)) break;
          case ZDFA_SUB: {
            // message index i: transition st[i+1] -> st[i+2] = (st, sn) of word i+1
            u32 pubs[ZK_DFA_NPUBLIC];
#pragma unroll
            for (u32 k = 0; k < ZK_DFA_NPUBLIC; ++k) pubs[k] = ZK_DFA_PUBLIC[k][0] | ((u32)ZK_DFA_PUBLIC[k][1] << 8);
            ZK_DFA_LOOP(2u, 1u, (q == 0) ^ ((st | (sn << 8)) == pubs[0] || (ZK_DFA_NPUBLIC > 1 && (st | (sn << 8)) == pubs[ZK_DFA_NPUBLIC > 1 ? 1 : 0])))
            break;
          }
          default: break;
        }
#undef ZK_DFA_LOOP
        break;
      }
      case ZSEG_RSLB: {
        // RemoveSoftLineBreaks arrays derived from the emailBody bytes alone
        // (helpers/remove-soft-line-breaks.circom:47-91): "=\r\n" at j  <=>  isSoftBreak[j]
        const u8* __restrict__ enc = rec + sg.src;
        const int half_tab = (int)s.inv_half;
        if (sg.a == ZRS_EQ) {
          ZK_FOR_CHUNKS(c) {
            const u32 r = r0 + (c >> 1), hf = c & 1u;
            int d = (int)sg.c - (int)enc[(r >> 1) + sg.b];   // isz.in = in[1] - in[0]
            uint4 v = zk_zero4();
            if (!(r & 1u)) { if (!hf) v.x = (d == 0); }
            else { d = max(-half_tab, min(half_tab, d)); pm = MONT; v = invtab[(u32)(d + half_tab) * 2 + hf]; }
            ZK_STORE(c, v);
          }
        } else {
          const u32 M = sg.b;
          ZK_FOR_CHUNKS(c) {
            uint4 v = zk_zero4();
            if (!(c & 1u)) {
              const u32 r = r0 + (c >> 1);
              if (sg.a == ZRS_TSB) v.x = (enc[r] == 61u) & (enc[r + 1] == 13u);
              else if (sg.a == ZRS_SB) v.x = (enc[r] == 61u) & (enc[r + 1] == 13u) & (enc[r + 2] == 10u);
              else {  // processed[r]: zero inside a soft break starting at r, r-1 or r-2 (starts only below M-2)
                bool z = false;
#pragma unroll
                for (u32 k = 0; k < 3; ++k) {
                  if (r >= k && r - k + 2 < M) { const u32 j = r - k; z = z || (enc[j] == 61u && enc[j + 1] == 13u && enc[j + 2] == 10u); }
                }
                v.x = z ? 0u : (u32)enc[r];
              }
            }
            ZK_STORE(c, v);
          }
        }
        break;
      }
      case ZSEG_NET: {
        // gate values of a loaded regex template (zkwg_net_core.h): 31-bit signed integer, or the inverse
        // of one (bit 31) from the table; a negative integer -m is the field element r - m
        const int half_tab = (int)s.inv_half;
        ZK_FOR_CHUNKS(c) {
          const u32 w = small[sg.src + r0 + (c >> 1)];
          const u32 hf = c & 1u;
          int d = (int)(w << 1) >> 1;
          uint4 v = zk_zero4();
          if (w & 0x80000000u) {
            d = max(-half_tab, min(half_tab, d));
            pm = MONT; v = invtab[(u32)(d + half_tab) * 2 + hf];
          } else if (d >= 0) {
            if (!hf) v.x = (u32)d;
          } else {
            v = hf ? make_uint4(0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u)
                   : make_uint4(0xf0000001u - (u32)(-d), 0x43e1f593u, 0x79b97091u, 0x2833e848u);
          }
          ZK_STORE(c, v);
        }
        break;
      }
      default:
        break;
    }
   }
  }
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(80))) void zk_expand_256(ZkSched s, ZkBufs B) { zk_expand_body<256, false>(s, B); }
__global__ __launch_bounds__(512) void zk_expand_512(ZkSched s, ZkBufs B) { zk_expand_body<512, false>(s, B); }
__global__ __launch_bounds__(1024) void zk_expand_1024(ZkSched s, ZkBufs B) { zk_expand_body<1024, false>(s, B); }
__global__ __launch_bounds__(256) void zk_expand_wave(ZkSched s, ZkBufs B) { zk_expand_body<256, true>(s, B); }
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void zk_expand_mont_256(ZkSched s, ZkBufs B) { zk_expand_body<256, false, true>(s, B); }

