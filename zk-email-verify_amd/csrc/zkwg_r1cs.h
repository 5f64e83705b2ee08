// `.r1cs` reader + constraint check core.
//
// What it replaces: circom_tester's `circuit.checkConstraints(witness)`, called right after
// `calculateWitness` by every circuit test of the reference (packages/circuits/tests/
// email-verifier.test.ts:44 and the other *.test.ts): for every constraint of the compiled circuit,
// (A.w) * (B.w) - (C.w) == 0 (mod r).  The `.r1cs` container is the iden3 binary format that circom
// writes and snarkjs/r1csfile read [EXT r1csfile 0.0.47, yarn.lock]: magic "r1cs", u32 version 1,
// u32 nSections; sections (u32 type, u64 size): 1 = header {u32 fieldSize, prime, u32 nWires,
// nPubOut, nPubIn, nPrvIn, u64 nLabels, u32 mConstraints}, 2 = constraints {3 x (u32 nTerms,
// nTerms x (u32 wire, fieldSize-byte LE coefficient))}, 3 = wire -> label map (u64 each).
//
// Device form: CSR over the 3m linear combinations; coefficients in Montgomery form with a class
// byte so that the overwhelmingly common +1 / -1 coefficients cost an add instead of a product.
#pragma once
#include <string.h>
#include <string>
#include <vector>
#include "zkwg_fr.h"
#include "zkwg_par.h"

enum ZkCoefKind : u8 { ZK_COEF_GENERIC = 0, ZK_COEF_ONE = 1, ZK_COEF_MINUS_ONE = 2 };

struct ZkR1csHost {
  u32 n_wires = 0, n_pub_out = 0, n_pub_in = 0, n_prv_in = 0, n_constraints = 0;
  u64 n_labels = 0;
  std::vector<u64> row_ptr;   // 3m + 1 offsets into wire/coef (A_0, B_0, C_0, A_1, ...)
  std::vector<u32> wire;
  std::vector<Fr> coef;       // Montgomery form
  std::vector<u8> kind;
  std::string err;
};

static inline bool zk_r1cs_parse(const u8* p, u64 len, ZkR1csHost& R) {
  auto fail = [&](const char* m) { R.err = m; return false; };
  if (len < 12 || memcmp(p, "r1cs", 4) != 0) return fail("not an .r1cs file (magic)");
  u32 version, nsec;
  memcpy(&version, p + 4, 4); memcpy(&nsec, p + 8, 4);
  if (version != 1) return fail("unsupported .r1cs version");
  const u8 *hdr = nullptr, *cons = nullptr;
  u64 hdr_len = 0, cons_len = 0, pos = 12;
  for (u32 s = 0; s < nsec; ++s) {
    if (pos + 12 > len) return fail("truncated section table");
    u32 type; u64 size;
    memcpy(&type, p + pos, 4); memcpy(&size, p + pos + 4, 8);
    pos += 12;
    if (size > len - pos) return fail("truncated section");
    if (type == 1) { hdr = p + pos; hdr_len = size; }
    else if (type == 2) { cons = p + pos; cons_len = size; }
    pos += size;
  }
  if (!hdr || !cons) return fail("header or constraint section missing");
  if (hdr_len < 4) return fail("short header");
  u32 fs;
  memcpy(&fs, hdr, 4);
  if (fs != 32 || hdr_len < 4 + 32 + 16 + 8 + 4) return fail("field size must be 32 bytes");
  const Fr prime = fr_p();
  if (memcmp(hdr + 4, prime.l, 32) != 0) return fail("prime is not the BN254 scalar field");
  const u8* q = hdr + 36;
  memcpy(&R.n_wires, q, 4); memcpy(&R.n_pub_out, q + 4, 4); memcpy(&R.n_pub_in, q + 8, 4); memcpy(&R.n_prv_in, q + 12, 4);
  memcpy(&R.n_labels, q + 16, 8); memcpy(&R.n_constraints, q + 24, 4);
  // every linear combination takes at least its 4-byte term count: bound the header's claim by
  // the section that is actually there before sizing anything from it
  if (3ull * R.n_constraints * 4 > cons_len) return fail("constraint count exceeds the constraint section");
  // pass 1 (sequential, cheap): where every linear combination starts; pass 2 (threads): wires, coefficients, classes
  const u64 n_lc = 3ull * R.n_constraints;
  R.row_ptr.assign(n_lc + 1, 0);
  std::vector<u64> off(n_lc);
  u64 cp = 0;
  for (u64 lc = 0; lc < n_lc; ++lc) {
    if (cp + 4 > cons_len) return fail("truncated constraint section");
    u32 nt;
    memcpy(&nt, cons + cp, 4);
    cp += 4;
    if ((u64)nt * 36 > cons_len - cp) return fail("truncated linear combination");
    off[lc] = cp;
    R.row_ptr[lc + 1] = R.row_ptr[lc] + nt;
    cp += (u64)nt * 36;
  }
  const u64 total = R.row_ptr[n_lc];
  R.wire.resize(total); R.coef.resize(total); R.kind.resize(total);
  const Fr one_m = fr_R(), minus_one_m = fr_neg(fr_R());
  const Fr one_s = fr_from_u64(1), minus_one_s = fr_neg(fr_from_u64(1));
  const unsigned T = n_lc > (1u << 16) ? zk_host_threads() : 1u;
  std::vector<int> bad(T, 0);
  zk_parallel_chunks(T, [&](unsigned ci, unsigned nc) {
    const u64 lo = n_lc * ci / nc, hi = n_lc * (ci + 1) / nc;
    for (u64 lc = lo; lc < hi; ++lc) {
      const u8* q = cons + off[lc];
      for (u64 t = R.row_ptr[lc]; t < R.row_ptr[lc + 1]; ++t, q += 36) {
        u32 w; Fr v;
        memcpy(&w, q, 4); memcpy(v.l, q + 4, 32);
        if (w >= R.n_wires) { bad[ci] = 1; return; }
        if (fr_geq(v, prime)) { bad[ci] = 2; return; }
        // (nearly every coefficient of a circom system is 1 or -1: no product for those)
        const u8 kd = fr_eq(v, one_s) ? ZK_COEF_ONE : (fr_eq(v, minus_one_s) ? ZK_COEF_MINUS_ONE : ZK_COEF_GENERIC);
        R.wire[t] = w;
        R.coef[t] = kd == ZK_COEF_ONE ? one_m : (kd == ZK_COEF_MINUS_ONE ? minus_one_m : fr_to_mont(v));
        R.kind[t] = kd;
      }
    }
  });
  for (int b : bad) { if (b == 1) return fail("wire index out of range"); if (b == 2) return fail("coefficient not reduced"); }
  return true;
}

// sum_t coef_t * w[wire_t] over one linear combination, standard form (w: standard-form witness, 32 B/slot)
// *canon is cleared when a witness value is not reduced (>= r): such a witness is rejected
// (mont_witness: the witness is in Montgomery form, so is the result; a value 0 or R -- most signals are bits --
// then needs no product: R * c = the stored coefficient)
// (reduce: a value >= r is reduced mod r instead of being skipped -- the prover-side evaluation must not drop terms)
ZK_HD Fr zk_r1cs_lc(const u64* __restrict__ row_ptr, const u32* __restrict__ wire, const Fr* __restrict__ coef,
                    const u8* __restrict__ kind, u64 lc, const Fr* __restrict__ w, bool* canon, bool mont_witness = false,
                    bool reduce = false) {
  Fr acc = fr_zero();
  const Fr unit = fr_R();
  for (u64 t = row_ptr[lc]; t < row_ptr[lc + 1]; ++t) {
    Fr x = w[wire[t]];
    if (fr_geq(x, fr_p())) {
      *canon = false;
      if (!reduce) continue;
      while (fr_geq(x, fr_p())) { u64 bw; x = fr_sub_raw(x, fr_p(), bw); }   // at most 5 rounds: 2^256 < 6 r
    }
    const u8 k = kind[t];
    if (k == ZK_COEF_ONE) acc = fr_add(acc, x);
    else if (k == ZK_COEF_MINUS_ONE) acc = fr_sub(acc, x);
    else if (mont_witness && fr_is_zero(x)) continue;
    else if (mont_witness && fr_eq(x, unit)) acc = fr_add(acc, coef[t]);
    else acc = fr_add(acc, fr_mont_mul(x, coef[t]));   // standard * Montgomery -> standard (Montgomery * Montgomery -> Montgomery)
  }
  return acc;
}
// constraint i holds for witness w?
ZK_HD bool zk_r1cs_check_one(const u64* row_ptr, const u32* wire, const Fr* coef, const u8* kind, u64 i, const Fr* w) {
  bool canon = true;
  const Fr a = zk_r1cs_lc(row_ptr, wire, coef, kind, 3 * i, w, &canon);
  const Fr b = zk_r1cs_lc(row_ptr, wire, coef, kind, 3 * i + 1, w, &canon);
  const Fr c = zk_r1cs_lc(row_ptr, wire, coef, kind, 3 * i + 2, w, &canon);
  // a*b == c  <=>  mont(a, b) == mont(c, 1)   (both sides carry the same R^-1)
  return canon && fr_eq(fr_mont_mul(a, b), fr_mont_mul(c, fr_from_u64(1)));
}
