// zk_misc_ev -- the small-integer blocks of EmailVerifier's body-hash path, one wavefront per
// email: BodyHashRegex scan, SelectRegexReveal / VarShiftLeft inputs, Base64Decode and the
// final body-hash comparison (packages/circuits/email-verifier.circom:124-146).  Produces
// image values + assertion status; zk_expand derives every per-index signal on the fly.
#include "zkwg_dev.h"
#include "zkwg_kernels.h"
#include "zkwg_regex_core.h"
#include "zkwg_bh_dfa.h"

static __device__ unsigned char ZKM_DELTA[ZK_DFA_STATES][256];
static __device__ unsigned int ZKM_CLSMASK[256];
static __device__ unsigned int ZKM_PRIMMASK[256];
extern "C" int zk_misc_init_tables(void) {
  if (hipMemcpyToSymbol(HIP_SYMBOL(ZKM_DELTA), ZK_DFA_DELTA, sizeof(ZK_DFA_DELTA)) != hipSuccess) return -1;
  if (hipMemcpyToSymbol(HIP_SYMBOL(ZKM_CLSMASK), ZK_DFA_CLSMASK, sizeof(ZK_DFA_CLSMASK)) != hipSuccess) return -1;
  if (hipMemcpyToSymbol(HIP_SYMBOL(ZKM_PRIMMASK), ZK_DFA_PRIMMASK, sizeof(ZK_DFA_PRIMMASK)) != hipSuccess) return -1;
  return 0;
}

__global__ __launch_bounds__(64) void zk_misc_ev(ZkSched s, ZkBufs B) {
  const u32 e = blockIdx.x;
  if (e >= B.n_emails) return;
  const u32 lane = threadIdx.x;
  const u8* rec = B.in + (u64)e * s.in_stride;
  u64* bits = B.bits + (u64)e * s.img_bits;
  u32* small = B.small + (u64)e * s.img_small;
  const u32 N = s.fr[0].max_bytes;
  // header bytes and the reveal array are staged in LDS: the scan is a serial byte loop
  extern __shared__ u32 dyn_lds[];
  u32* rev = dyn_lds;                 // N words
  u8* hdr = (u8*)(dyn_lds + N);       // N bytes
  u8* stl = hdr + N;                  // N + 3 state bytes (st[0..N+2))
  u8* live = stl + N + 4;             // N + 3
  // the DFA transition table (7 KiB) is staged in LDS too: the scan is a chain of dependent lookups
  unsigned char (*delta_l)[256] = (unsigned char (*)[256])(live + N + 4);
  for (u32 i = lane; i < ZK_DFA_STATES * 64u; i += 64) ((u32*)delta_l)[i] = ((const u32*)ZKM_DELTA)[i];
  __shared__ u32 ok_sh;
  if (lane == 0) ok_sh = 1;
  for (u32 i = lane; i < N; i += 64) { rev[i] = 0; hdr[i] = rec[s.fr[0].in_data + i]; }
  __syncthreads();
  const u32 start = *(const u32*)(rec + s.in_off[8]);  // bodyHashIndex
  if (s.net_mode) {
    // loaded template: zk_net_eval (launched before this kernel) left reveal0 and the match output in the image
    for (u32 i = lane; i < N; i += 64) rev[i] = small[s.m_rev + i];
    __syncthreads();
  }
  if (lane == 0) {
    if (s.net_mode) {
      if (small[s.m_net_out] != 1u) atomicAnd(&ok_sh, 0u);          // bhRegexMatch === 1
    } else {
      // BodyHashRegex DFA scan (zkwg_regex_core.h): states, live chain, helper signals, reveal0
      const u32 acc_count = zk_bh_dfa_scan(hdr, N, delta_l, stl, live, small + s.m_dfa_own, rev);
      if (acc_count == 0) atomicAnd(&ok_sh, 0u);                    // bhRegexMatch === 1
      small[s.m_dfa_acc] = acc_count;
    }
    small[s.m_bh_idx] = start;
    bits[s.b_shift] = start;
    u32 blh = 0;
    for (u32 n = N - 1; n > 0; n >>= 1) ++blh;
    if (start >= (1u << blh)) atomicAnd(&ok_sh, 0u);                // VarShiftLeft.n2b = Num2Bits(log2Ceil(N))
    if ((u64)start + 43 >= (1ull << s.sel_bits)) atomicAnd(&ok_sh, 0u);  // GreaterThan(bl) Num2Bits at i = 0
  }
  __syncthreads();
  if (!s.net_mode) for (u32 i = lane; i < N; i += 64) small[s.m_rev + i] = rev[i];
  // per-position words for zk_expand's ZSEG_DFA: in | st<<8 | nx<<16 | st_next<<24, class mask, prim mask
  for (u32 i = lane; !s.net_mode && i < N + 2; i += 64) {
    const u32 b = i == 0 ? 255u : (i <= N ? hdr[i - 1] : 0u);
    const u32 st = stl[i];
    const u32 nx = (st && i <= N) ? ZKM_DELTA[st][b] : 255u;
    small[s.m_dfa_st + i] = b | (st << 8) | (nx << 16) | ((u32)stl[i + 1 <= N + 2 ? i + 1 : N + 2] << 24);
    small[s.m_dfa_cm + i] = ZKM_CLSMASK[b];
    small[s.m_dfa_pm + i] = ZKM_PRIMMASK[b];
  }
  // SelectRegexReveal assertions (utils/regex.circom:39-47)
  for (u32 i = lane; i < N; i += 64) {
    bool bad = false;
    if (i == start) bad = rev[i] == 0 || (i > 0 && rev[i - 1] != 0);
    if ((u64)i > (u64)start + 43 && rev[i] != 0) bad = true;
    if (bad) atomicAnd(&ok_sh, 0u);
  }
  // bhBase64[g] = VarShiftLeft(N, 44)(bhReveal, bodyHashIndex)[g] = rev[(g + shift) mod N]
  __shared__ u32 vals[44];
  if (lane < 44) {
    u32 ch = rev[(lane + start % N) % N];
    small[s.m_chars + lane] = ch;
    // Base64Lookup (lib/base64.circom:71-128)
    u32 v = 0;
    bool valid = true;
    if (ch >= 65 && ch <= 90) v = ch - 65;
    else if (ch >= 97 && ch <= 122) v = ch - 71;
    else if (ch >= 48 && ch <= 57) v = ch + 4;
    else if (ch == 43) v = 62;
    else if (ch == 47) v = 63;
    else if (ch == 61) v = 0;
    else valid = false;
    vals[lane] = v;
    if (!valid) atomicAnd(&ok_sh, 0u);                              // base64.circom:127
  }
  __syncthreads();
  // computedBodyHashInts[i].out === headerBodyHash[i]  (email-verifier.circom:139-146)
  if (lane < 32) {
    u32 g = lane / 3, k = lane % 3;
    u32 v0 = vals[4 * g], v1 = vals[4 * g + 1], v2 = vals[4 * g + 2], v3 = vals[4 * g + 3];
    u32 byte = k == 0 ? ((v0 << 2) | (v1 >> 4)) : (k == 1 ? (((v1 & 15) << 4) | (v2 >> 2)) : (((v2 & 3) << 6) | v3));
    u32 w = small[s.fr[1].m_digest + (lane >> 2)];
    u32 expect = (w >> (24 - 8 * (lane & 3))) & 0xff;
    if ((byte & 0xff) != expect) atomicAnd(&ok_sh, 0u);
  }
  __syncthreads();
  if (lane == 0 && !ok_sh) B.status[e] = 4;
}
