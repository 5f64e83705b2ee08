// zk_net_eval -- BodyHashRegex from a loaded circom template (zkwg_circom.h): net_lanes (32 by default) lanes per
// email -- two emails per wavefront -- walk the gate list, up to net_lanes gates (one per lane) per step; the gates of a step are mutually
// independent and only read values of earlier steps.  Results go to the email's image (zk_expand's
// ZSEG_NETP streams them out) and, when a later gate reads them, to an LDS word the loader assigned by
// liveness; an operand is an LDS offset, the image is write-only.  Steps whose records the loader proved
// exact in 32 bits take a branch-free 32-bit path (zk_net_record32), the others the 64-bit one.  One
// wavefront per SIMD is the normal occupancy (1,024 emails per prepare launch), so memory latency is
// hidden by software pipelining: the 64-byte records are fetched ZKN_DEPTH steps ahead into a register ring.
// Memory order: a wavefront's LDS accesses execute in program order and a step never needs a value of
// its own step, so no barrier is required between steps.
#include "zkwg_dev.h"
#include "zkwg_kernels.h"
#include "zkwg_net_core.h"

#define ZKN_DEPTH 8

__device__ __forceinline__ ZkNetChains zk_net_chains(const ZkSched& s, const ZkBufs& B) {
  ZkNetChains K;
  K.n_in = s.fr[0].max_bytes;
  K.f_end = s.net_chain_end; K.f_smax = s.net_chain_smax; K.f_mw = s.net_chain_mw;
  K.b_end = s.net_bchain_end; K.b_smax = s.net_bchain_smax; K.b_mw = s.net_bchain_mw; K.b_fdim = s.net_bchain_fdim;
  K.f_cls = B.net_cclass; K.f_delta = B.net_cdelta; K.f_mask = B.net_cmask;
  K.b_cls = B.net_bclass; K.b_delta = B.net_bdelta; K.b_mask = B.net_bmask;
  return K;
}

__global__ __launch_bounds__(64) void zk_net_eval(ZkSched s, ZkBufs B) {
  // s.net_lanes (16 / 32 / 64) lanes per email: 64 / net_lanes emails share the wavefront, each with its own LDS image;
  // the lanes of all of them run the same record stream (the records of a step are fetched once per lane index)
  const u32 L = s.net_lanes, EW = 64u / L;
  const u32 lane = threadIdx.x, sub = lane / L, gl = lane % L;
  const u32 e_raw = blockIdx.x * EW + sub;
  if (blockIdx.x * EW >= B.n_emails) return;
  const bool live = e_raw < B.n_emails;
  const u32 e = live ? e_raw : B.n_emails - 1u;      // an idle sub-group shadows the last email (same values, same stores)
  extern __shared__ u32 dyn_lds[];
  int* lds = (int*)dyn_lds + sub * s.net_lds_words;
  const u32 msg_base = s.net_pins;     // [gate values | message bytes | 0 | scratch | masks]
  const u32 N = s.fr[0].max_bytes;
  const u8* rec = B.in + (u64)e * s.in_stride + s.fr[0].in_data;
  // message bytes, and per byte its mask words: the truth of every byte-local boolean a per-email gate reads
  // (zkwg_circom.h localize), one table lookup per byte and word
  // ... and, when recurrences were collapsed (zkwg_circom.h chain_pass), the chains' mask words: the truth of every boolean of a
  // chain a gate of the list reads, by (class of the position, state entering it -- zk_net_scan, symbol)
  const ZkNetChains K = zk_net_chains(s, B);
  const u32 MW = s.net_mask_words, MS = MW + K.f_mw + K.b_mw;
  const u8* fstate = (const u8*)(B.small + (u64)e * s.img_small + s.m_net_st);
  const u8* bstate = (const u8*)(B.small + (u64)e * s.img_small + s.m_net_bst);
  u32* small = B.small + (u64)e * s.img_small;
  for (u32 i = gl; i < N; i += L) {
    lds[msg_base + i] = (int)rec[i];
    zk_net_mask_words(K, MW, B.net_mask_tab, i, rec, fstate, bstate, &lds[s.net_lds_masks + i * MS]);
    small[s.m_net_pw + i] = zk_net_pos_word(K, i, rec, fstate, bstate);   // what zk_expand decodes the table-served slots of position i from
  }
  if (gl == 0) lds[msg_base + N] = 0;
  const u32 scratch = msg_base + N + 1u;
  __syncthreads();
  u32* img = small + s.m_net;
  const uint4* __restrict__ R = (const uint4*)B.net_records;
  const u32* __restrict__ CNT = B.net_counts;
  const u32 nsteps = s.net_steps;
  const long long inv_limit = (long long)s.inv_half;
  bool ok = true;

  // Register ring of records, ZKN_DEPTH steps ahead.  Steps come in groups of ZKN_DEPTH (the loader pads the list with
  // empty steps); the gate counts (bit 15 = the step holds a record for the 64-bit path) are wave-uniform and read
  // with scalar loads, one group ahead of the fetch side, which itself runs one group ahead of the execute side.
  // (Counts fetched with vector loads and v_readlane made the compiler wait for vmcnt(0) at the top of every group.)
  uint4 ring[ZKN_DEPTH][4];
  u32 fetch_base = 0;                   // record index of the next step to fetch
  const u32 ngroups = (nsteps + ZKN_DEPTH - 1u) / ZKN_DEPTH;
  // every lane loads (lanes past the step's count get records of later steps and are neutralised when executed):
  // no control flow around the loads, so the compiler's s_waitcnt bookkeeping keeps the ring's distance
  auto fetch = [&](u32 n, uint4* dst) {
    const uint4* p = R + ((u64)fetch_base + gl) * 4;
    dst[0] = p[0]; dst[1] = p[1]; dst[2] = p[2]; dst[3] = p[3];
    fetch_base += n & 0x7fu;
  };
  u32 cur[ZKN_DEPTH], nxt[ZKN_DEPTH];
#pragma unroll
  for (int d = 0; d < ZKN_DEPTH; ++d) { cur[d] = CNT[d]; nxt[d] = CNT[ZKN_DEPTH + d]; }
#pragma unroll
  for (int d = 0; d < ZKN_DEPTH; ++d) fetch(cur[d], ring[d]);
  for (u32 g = 0; g < ngroups; ++g) {
    u32 nn[ZKN_DEPTH];
#pragma unroll
    for (int d = 0; d < ZKN_DEPTH; ++d) nn[d] = CNT[(g + 2u) * ZKN_DEPTH + d];   // two groups ahead (padded)
#pragma unroll
    for (int d = 0; d < ZKN_DEPTH; ++d) {
      const u32 cn = cur[d];
      const u32 n = cn & 0x7fu;
      if (cn & 0x8000u) {
        if (gl < n) {
          const u32 r[16] = {ring[d][0].x, ring[d][0].y, ring[d][0].z, ring[d][0].w, ring[d][1].x, ring[d][1].y, ring[d][1].z, ring[d][1].w,
                             ring[d][2].x, ring[d][2].y, ring[d][2].z, ring[d][2].w, ring[d][3].x, ring[d][3].y, ring[d][3].z, ring[d][3].w};
          ok &= zk_net_record(r, lds, lds, img, small + s.m_net_out, small + s.m_rev, inv_limit);
        }
      } else {
        // all lanes run the record they hold; the lanes past the count write to scratch words
        const bool act = gl < n;
        const u32 r[16] = {ring[d][0].x, act ? ring[d][0].y : s.net_total, ring[d][0].z, act ? ring[d][0].w : scratch,
                           ring[d][1].x, ring[d][1].y, ring[d][1].z, ring[d][1].w,
                           ring[d][2].x, ring[d][2].y, ring[d][2].z, ring[d][2].w, ring[d][3].x, ring[d][3].y, ring[d][3].z, ring[d][3].w};
        zk_net_record32(r, lds, lds, img);
      }
      __builtin_amdgcn_wave_barrier();
      fetch(nxt[d], ring[d]);   // refill this ring entry with the step ZKN_DEPTH ahead
    }
#pragma unroll
    for (int d = 0; d < ZKN_DEPTH; ++d) { cur[d] = nxt[d]; nxt[d] = nn[d]; }
  }
  const u64 bad = __ballot(!ok);
  const u64 mine = (L == 64u ? ~0ull : ((1ull << L) - 1ull)) << (sub * L);
  if ((bad & mine) != 0ull && gl == 0 && live) B.status[e] = 4;
}

// zk_net_scan -- the collapsed recurrences (zkwg_circom.h chain_pass): one lane per email walks
// state' = delta[class of the position][state][symbol] forward over the bytes the forward chain covers, then backward over those of
// the backward chain (whose symbol includes the forward state), and leaves the state ENTERING every position in the image, one
// byte each.  ~2 N dependent table lookups per email (the tables stay in the cache: a few KB per class).
__global__ __launch_bounds__(64) void zk_net_scan(ZkSched s, ZkBufs B) {
  const u32 e = blockIdx.x * 64u + threadIdx.x;
  if (e >= B.n_emails) return;
  zk_net_scan_email(zk_net_chains(s, B), B.in + (u64)e * s.in_stride + s.fr[0].in_data, B.small + (u64)e * s.img_small + s.m_net_st,
                    B.small + (u64)e * s.img_small + s.m_net_bst);
}
