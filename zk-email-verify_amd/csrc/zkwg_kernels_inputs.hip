// zk_gen_inputs -- batched input generation on the device: restates
// generateEmailVerifierInputsFromDKIMResult (packages/helpers/src/input-generators.ts:190-252),
// sha256Pad / generatePartialSHA / findIndexInUint8Array (sha-utils.ts:9-111, incl. the SHA-256 midstate of
// lib/fast-sha256.ts:240-251 cacheState) and toCircomBigIntBytes (binary-format.ts:71-83).
// One wavefront per email: raw canonical header / body bytes + 2048-bit big-endian key and signature
// in, one packed input record out.  The per-email error codes mirror the reference's thrown errors.
#include "zkwg_dev.h"
#include "zkwg_kernels.h"

// 17 x 121-bit limbs (16-byte LE each) of a 256-byte big-endian integer
__device__ inline void zk_be2048_to_limbs(const u8* __restrict__ be, u8* __restrict__ out, u32 lane) {
  for (u32 i = lane; i < 17; i += 64) {
    u64 lo = 0, hi = 0;
    for (u32 bit = 0; bit < 121; bit += 8) {              // byte-wise gather of bits [121 i, 121 i + 121)
      const u32 pos = 121 * i + bit;                      // little-endian bit position
      // 16 bits straddling: value bits [pos, pos+8)
      u32 v = 0;
      for (u32 k = 0; k < 2; ++k) {
        const u32 byte = (pos >> 3) + k;
        if (byte < 256) v |= (u32)be[255 - byte] << (8 * k);
      }
      v = (v >> (pos & 7)) & 0xffu;
      const u32 take = min(8u, 121u - bit);
      v &= (1u << take) - 1u;
      if (bit < 64) { lo |= (u64)v << bit; if (bit + take > 64) hi |= (u64)v >> (64 - bit); }
      else hi |= (u64)v << (bit - 64);
    }
    *(u64*)(out + 16 * i) = lo;
    *(u64*)(out + 16 * i + 8) = hi;
  }
}

__global__ __launch_bounds__(64) void zk_gen_inputs(ZkSched s, ZkDkimBatch D, u8* __restrict__ recs,
                                                   int* __restrict__ gen_status, u32 n) {
  const u32 e = blockIdx.x;
  if (e >= n) return;
  const u32 lane = threadIdx.x;
  u8* rec = recs + (u64)e * s.in_stride;
  const u32 N = s.fr[0].max_bytes;
  const u32 M = s.body ? s.fr[1].max_bytes : 0;
  const u8* hdr = D.headers + (u64)e * D.header_stride;
  const u32 hl = D.header_len[e];
  __shared__ int err;
  __shared__ u32 cut_sh;
  if (lane == 0) err = 0;
  for (u32 i = lane; i < s.in_stride; i += 64) rec[i] = 0;
  __syncthreads();
  // sha256Pad(headers, maxHeadersLength): message | 0x80 | zeros | 64-bit BE bit length, zero padded to max
  const u64 hpad64 = (((u64)hl + 9 + 63) / 64) * 64;      // 64-bit: a huge length must not wrap
  const bool hdr_ok = hpad64 <= N && hl <= D.header_stride;
  const u32 hpad = hdr_ok ? (u32)hpad64 : 0u;
  if (!hdr_ok) { if (lane == 0) err = 1; }   // "Padding to max length did not complete properly!"
  else {
    for (u32 i = lane; i < hl; i += 64) rec[s.in_off[0] + i] = hdr[i];
    if (lane == 0) {
      rec[s.in_off[0] + hl] = 0x80;
      const u64 bits = (u64)hl * 8;
      for (u32 k = 0; k < 8; ++k) rec[s.in_off[0] + hpad - 1 - k] = (u8)(bits >> (8 * k));
      *(u32*)(rec + s.in_off[6]) = hpad;
    }
  }
  zk_be2048_to_limbs(D.pubkey_be + (u64)e * 256, rec + s.in_off[3], lane);
  zk_be2048_to_limbs(D.signature_be + (u64)e * 256, rec + s.in_off[4], lane);
  if (s.body) {
    const u8* body = D.bodies + (u64)e * D.body_stride;
    const u32 bl = D.body_len[e];
    // bodyHashIndex = headers.toString().indexOf(bodyHash)   (-1 -> 0xffffffff)
    if (lane == 0) {
      const u8* bh = D.body_hash_b64 + (u64)e * 44;
      u32 idx = 0xffffffffu;
      for (u32 i = 0; hdr_ok && i + 44 <= hl && idx == 0xffffffffu; ++i) {   // never read past the header slot
        u32 k = 0;
        while (k < 44 && hdr[i + k] == bh[k]) ++k;
        if (k == 44) idx = i;
      }
      *(u32*)(rec + s.in_off[8]) = idx;
    }
    // padded body (virtual): body | 0x80 | zeros | length, length bpad = 64-multiple
    const bool body_ok = bl <= D.body_stride && (u64)bl + 9 + 63 < (1ull << 32);
    const u32 bpad = body_ok ? ((bl + 9 + 63) / 64) * 64 : 0u;
    auto padded = [&](u32 i) -> u32 {
      if (i < bl) return body[i];
      if (i == bl) return 0x80u;
      if (i >= bpad - 8 && i < bpad) return (u32)(((u64)bl * 8) >> (8 * (bpad - 1 - i))) & 0xffu;
      return 0u;
    };
    if (lane == 0) {
      // findIndexInUint8Array (sha-utils.ts:9-24), literally: on a mismatch j restarts at 0 and i advances
      u32 sel_idx = 0;
      if (!body_ok) err = 4;                                  // body longer than its slot: nothing is read from it
      if (D.selector_len && body_ok) {
        const u32 total = max(M, ((bl + 63 + 65) / 64) * 64);  // bodyPadded length (input-generators.ts:219-221)
        u32 i = 0, j = 0;
        sel_idx = 0xffffffffu;
        while (i < total) {
          if (padded(i) == D.selector[j]) { ++j; if (j == D.selector_len) { sel_idx = i - j + 1; break; } }
          else j = 0;
          ++i;
        }
        if (sel_idx == 0xffffffffu) {
          // getAdjustedSelector (input-generators.ts:44-105, 224-227): a selector that is not in the body as it
          // stands may span a "=\r\n" soft line break.  It is then looked up in the body with the soft breaks removed
          // (removeSoftLineBreaks of the PADDED body, :127-158), mapped back, and the body's own bytes from there on
          // -- selector length + 3 of them -- become the selector of the search above.
          const u32 L = D.selector_len;
          bool in_body = false;                              // bodyString.includes(selector): a general substring search
          for (u32 a = 0; !in_body && a + L <= bl; ++a) {
            u32 k = 0;
            while (k < L && body[a + k] == D.selector[k]) ++k;
            in_body = k == L;
          }
          auto sb = [&](u64 k) -> bool { return k + 2 < total && padded((u32)k) == 61u && padded((u32)k + 1) == 13u && padded((u32)k + 2) == 10u; };
          u32 orig = 0xffffffffu;
          for (u32 a = 0; !in_body && a < total && orig == 0xffffffffu; ++a) {
            if (sb(a) || (a >= 1 && sb(a - 1)) || (a >= 2 && sb(a - 2))) continue;   // not a byte of the cleaned content
            u32 k = 0, q = a;
            while (k < L && q < total) {                     // cleanString.indexOf(selector) starting at clean position of `a`
              if (sb(q)) { q += 3; continue; }
              if (padded(q) != D.selector[k]) break;
              ++k; ++q;
            }
            if (k == L) orig = a;
          }
          u8 adj[160];
          u32 al = 0;
          if (orig != 0xffffffffu && L + 3 <= sizeof(adj) && orig < bl) {
            al = min(L + 3, bl - orig);                      // bodyString.slice(originalIndex, originalIndex + selector.length + 3)
            for (u32 k = 0; k < al; ++k) adj[k] = body[orig + k];
            i = 0; j = 0;
            while (i < total) {                              // generatePartialSHA's findIndexInUint8Array with the adjusted selector
              if (padded(i) == adj[j]) { ++j; if (j == al) { sel_idx = i - j + 1; break; } }
              else j = 0;
              ++i;
            }
          }
          if (sel_idx == 0xffffffffu) err = 3;              // "SHA precompute selector ... not found in the body"
        }
      }
      u32 cut = sel_idx == 0xffffffffu ? 0 : (sel_idx / 64) * 64;
      if (!err && bpad - cut > M) err = 2;                  // "Remaining body ... is longer than max"
      cut_sh = cut;
      // partialSha(precomputeText): SHA-256 state after body[0..cut)
      u32 st[8];
      zk_sha256_iv(st);
      if (!err) {
        u8 blk[64];
        for (u32 b = 0; b < cut; b += 64) {
          for (u32 k = 0; k < 64; ++k) blk[k] = (u8)padded(b + k);
          zk_sha256_compress(st, blk);
        }
      }
      for (u32 j = 0; j < 8; ++j) for (u32 k = 0; k < 4; ++k) rec[s.in_off[2] + 4 * j + k] = (u8)(st[j] >> (24 - 8 * k));
      *(u32*)(rec + s.in_off[7]) = err ? 0u : bpad - cut;
    }
    __syncthreads();
    if (!err) {
      const u32 cut = cut_sh;
      for (u32 i = lane; i + cut < bpad && i < M; i += 64) rec[s.in_off[1] + i] = (u8)padded(cut + i);
    }
    if (s.rslb) {
      // decodedEmailBodyIn = removeSoftLineBreaks(bodyRemaining) (input-generators.ts:127-158, 241-244):
      // drop every "=\r\n", keep the order, zero fill.  The pattern cannot overlap itself, so the
      // sequential scan equals a stream compaction: 64 bytes per step, positions from a ballot prefix.
      __syncthreads();
      if (!err) {
        const u8* eb = rec + s.in_off[1];
        u8* dec = rec + s.in_off[11];
        auto sb = [&](int k) -> bool { return k >= 0 && (u32)k + 2 < M && eb[k] == 61 && eb[k + 1] == 13 && eb[k + 2] == 10; };
        u32 base = 0;
        for (u32 i0 = 0; i0 < M; i0 += 64) {
          const int j = (int)(i0 + lane);
          const bool keep = (u32)j < M && !(sb(j) || sb(j - 1) || sb(j - 2));
          const u64 m = __ballot(keep);
          if (keep) dec[base + __popcll(m & ((1ull << lane) - 1ull))] = eb[j];
          base += __popcll(m);
        }
      }
    }
  }
  __syncthreads();
  if (lane == 0) gen_status[e] = err;
}
