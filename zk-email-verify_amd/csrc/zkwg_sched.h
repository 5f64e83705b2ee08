// Witness schedule of one circuit: the compact "kept-v1" layout lowered into
//   (1) a static SEGMENT TABLE that tiles the witness [0, W) -- consumed by the single
//       streaming expansion kernel zk_expand (the HBM-write-bound kernel), and
//   (2) the offsets at which the compute kernels deposit their results inside the
//       per-email compact IMAGE (packed bit groups / small ints / field elements).
// Built once per circuit on the host (zkwg_layout.h) and passed to kernels by value.
#pragma once
#include <stdint.h>
#include "zkwg_fr.h"

// Kept signals of one circomlib Sha256compression instance, in layout order:
//   sigmaPlus[48] x (sigma1.xor3.out32, .mid32, sigma0.xor3.out32, .mid32, sum.out34) = 48 x 162
//   t1[64]        x (ch.out32, bigsigma1.xor3.out32, .mid32, sum.out35)               = 64 x 131
//   t2[64]        x (bigsigma0.xor3.out32, .mid32, maj.out32, .mid32, sum.out33)      = 64 x 161
//   suma[64] x 33, sume[64] x 33, fsum[8] x 33
#define ZK_COMP_SLOTS 30952u
#define ZK_SP_SLOTS 162u
#define ZK_T1_SLOTS 131u
#define ZK_T2_SLOTS 161u
// trace groups (one u64 word per 32..35-bit group)
#define ZK_G_SP 0u
#define ZK_G_T1 240u
#define ZK_G_T2 496u
#define ZK_G_SUMA 816u
#define ZK_G_SUME 880u
#define ZK_G_FSUM 944u
#define ZK_TRACE_GROUPS 952u

// ---------------------------------------------------------------- segment table
// Every witness slot belongs to exactly one segment.  `src` indexes the email's image
// (or its input record); bit vectors are LSB-first inside little-endian u64 words.
enum ZkSegType : u32 {
  ZSEG_SMALL = 0,   // slot r = small[src + r]                                  (u32 values)
  ZSEG_FR = 1,      // slot r = fr[src + r]                                     (32-byte values)
  ZSEG_BITS = 2,    // groups of a bits, b words/group: bit (r%a) of bits[src + (r/a)*b ...]
  ZSEG_SHA_SP = 3,  // 162-slot periods over 5 words (32,32,32,32,34 bits)
  ZSEG_SHA_T1 = 4,  // 131-slot periods over 4 words (32,32,32,35)
  ZSEG_SHA_T2 = 5,  // 161-slot periods over 5 words (32,32,32,32,33)
  ZSEG_ISZ = 6,     // IsZero pairs: d = (i32)small[src + r/2]; even: d==0, odd: d^-1 (table)
  ZSEG_SEL = 7,     // ItemAtIndex(a) x 256: idx=(i32)small[src]; digest words small[b..b+8)
  ZSEG_IN8 = 8,     // slot r = input byte  in[src + r]
  ZSEG_IN8BITS = 9, // Num2Bits(8) of input bytes: bit (r&7) of in[src + r/8]
  ZSEG_LIMB = 10,   // slot r = 16-byte LE limb in[src + 16 r]
  ZSEG_LTBITS = 11, // LessThan(a) arrays: v = (i32)small[src] + 2^a - i, i = r/(a+1), bit r%(a+1)
  ZSEG_REGSEL = 12, // SelectRegexReveal(c, 44) per-index comparators: start=(i32)small[src], a=bitLength, reveal bytes small[b..b+c)
  ZSEG_VSHIFT = 13, // VarShiftLeft(a, .).tmp[j][i] = small[b + (i + (shift & (2^(j+1)-1))) % a], shift=small[src]
  ZSEG_B64BITS = 14,// Base64Decode bitsIn: 6 bits of the decoded value of char small[src + r/6]
  ZSEG_B64 = 15,    // Base64Lookup(char small[src + r/68]): 8 mids, 6 x 9 comparator bits, 3 IsZero pairs
  ZSEG_DFA = 16,    // BodyHashRegex DFA circuit arrays: a = ZkDfaKind, b/c = parameters, src = small idx of the per-position words (then class masks, primitive masks)
  ZSEG_IN8MASK = 17,// ByteMask: in[src + r] * in[a + r]   (data byte times mask byte)
  ZSEG_RSLB = 18,   // RemoveSoftLineBreaks byte-derived arrays over in[src ..]: a = ZkRslbKind, b/c = parameters
  ZSEG_HOLE = 19,   // signals this schedule does not produce: left to the linear completion pass (zkwg_full.h)
  // 20: (rounds 2-4: ZSEG_NET, the region of a loaded regex template read word by word from the image; replaced by ZSEG_NETP)
  ZSEG_NETP = 21,   // one periodic run of a loaded regex template's region (zkwg_circom.h finish_region): src = first period descriptor, a = period,
                    // b = the position the descriptors are relative to, c = region index of the run's first slot; slot r = descriptor
                    // src + r % a at position b + r / a (+ the descriptor's own offset) (zkwg_net_core.h ZkNetDec)
  ZSEG_NETQ = 22,   // a long run of the same region with a DENSE table: src = the table's offset in the region's tables, a = period, b = position of
                    // period 0; slot r = table[(fstate, byte of position b + r / a)][r % a] -- no descriptor (zkwg_net_core.h zk_netq_word)
  ZSEG_NTYPES = 23
};

// ZSEG_RSLB kinds (helpers/remove-soft-line-breaks.circom:14-126); enc = the emailBody bytes at src
enum ZkRslbKind : u32 {
  ZRS_PROC = 0,  // processed[r] = (1 - shouldZero[r]) * enc[r]           (b = maxLength)
  ZRS_TSB = 1,   // tempSoftBreak[r] = (enc[r] == '=') * (enc[r+1] == CR)
  ZRS_SB = 2,    // isSoftBreak[r]   = tempSoftBreak[r] * (enc[r+2] == LF)
  ZRS_EQ = 3     // IsEqual([enc[i + b], c]) pairs (isz.out, isz.inv), i = r / 2, isz.in = c - enc[i + b]
};

// ZSEG_DFA kinds (component arrays of the regex circuit, one entry per header position)
enum ZkDfaKind : u32 {
  ZDFA_EQ = 0,   // IsEqual(in[i], b): (isz.out, isz.inv), isz.in = b - in[i]
  ZDFA_LT = 1,   // LessThan(8) Num2Bits(9) of  b + (c ? +in[i] : -in[i])
  ZDFA_RNG = 2,  // AND: b <= in[i] <= c
  ZDFA_CLS = 3,  // MultiOR over the primitive tests in mask c: (is_zero.out, is_zero.inv)
  ZDFA_AND = 4,  // transition gate: from-state b (0: from_zero_enabled[i]), class c
  ZDFA_TMP = 5,  // MultiOR states_tmp[i+1][b]: (is_zero.out, is_zero.inv)
  ZDFA_FZE = 6,  // MultiNOR -> from_zero_enabled[i]: (is_zero.out, is_zero.inv)
  ZDFA_ST = 7,   // MultiOR(2) states[i+1][b]: (is_zero.out, is_zero.inv)
  ZDFA_SUB = 8   // MultiOR over the public transitions at message index i: (is_zero.out, is_zero.inv)
};

struct ZkSeg {
  u64 slot;    // first witness slot
  u32 nslots;  // length
  u32 type;    // ZkSegType
  u32 src;     // source offset (see type)
  u32 a, b, c; // type parameters
  u32 r0;      // index of the segment's first element inside the logical array (0 unless a `.sym`
               // remap split the array: zkwg_build.h zk_remap_segments)
  u32 pad;     // kernel-private: reciprocal of the type's period (zkwg_build.h zk_finish_tables), 0 = divide
};
#define ZK_MONT_LIMBS 51u   // 128-bit limbs of an input record (pubkey, signature, message: 3 x 17)

// segments whose slots are all immediate-valued (zkwg_expand_dec.h: no references), and the period a segment's decoder divides by
ZK_HD bool zk_seg_is_immediate(const ZkSeg& g) {
  switch (g.type) {
    case ZSEG_BITS: case ZSEG_SHA_SP: case ZSEG_SHA_T1: case ZSEG_SHA_T2: case ZSEG_IN8: case ZSEG_IN8MASK:
    case ZSEG_IN8BITS: case ZSEG_LTBITS: case ZSEG_B64BITS: case ZSEG_HOLE: return true;
    case ZSEG_DFA: return g.a != ZDFA_EQ;
    case ZSEG_RSLB: return g.a != ZRS_EQ;
    default: return false;
  }
}
ZK_HD u32 zk_seg_period(const ZkSeg& g) {
  switch (g.type) {
    case ZSEG_BITS: return g.a;
    case ZSEG_SEL: return 3u * g.a;
    case ZSEG_LTBITS: return g.a + 1u;
    case ZSEG_REGSEL: return g.a + 7u;
    case ZSEG_VSHIFT: return g.a;
    case ZSEG_NETP: case ZSEG_NETQ: return g.a;
    default: return 0;
  }
}

// zk_expand works in pieces of 256 K slots (K = 1, 2, 4: 8 K KiB of output), one per workgroup.  One table entry per piece:
// a piece inside a single segment carries that segment's parameters (type < ZSEG_NTYPES) and the logical index of its
// first slot; a piece that straddles segments (type = ZSEG_NTYPES) names the first segment to look at.
struct ZkPortionEntry {
  u32 type, src, a, b, c, magic;
  u32 r_start;
  u32 first_seg;
};

struct ZkShaFrame {      // one Sha256Bytes / Sha256BytesPartial instance
  u32 max_bytes;         // maxByteLength
  u32 nblocks;           // maxBits / 512
  u32 lenbits;           // log2Ceil(maxBits)
  u32 partial;           // 1 = Sha256BytesPartial (preHash input)
  u32 in_data;           // byte offset of the data bytes in an input record
  u32 in_len;            // byte offset of the u32 length
  u32 in_pre;            // byte offset of precomputedSHA[32] (partial only)
  u32 hstate_base;       // first chaining-state index (units of 8 x u32) in scratch
  u32 block_base;        // index of this frame's first block among the email's blocks
  // image offsets
  u32 b_trace;           // bits: nblocks x ZK_TRACE_GROUPS words
  u32 b_lenbits;         // bits: 1 word, the LessEqThan Num2Bits input
  u32 b_digest;          // bits: 4 words, selected hash as one LSB-first 256-bit group (bit k = out[k])
  u32 m_ibi;             // small: inBlockIndex
  u32 m_idx;             // small: (i32) inBlockIndex - 1
  u32 m_digest;          // small: 8 words, state after block idx (the selected hash)
  // EmailVerifier-level values owned by the frame's chain thread (0xffffffff = absent)
  u32 m_len;             // small: the length input
  u32 m_len_m1;          // small: (i32) length - 1   (AssertZeroPadding LessThan inputs)
  u32 b_len;             // bits: 1 word, the length (Num2Bits(log2Ceil(max)))
  u32 azp;               // 1: check length range + zero padding (email-verifier.circom:58-63,116-121)
};

// image offsets of one BigLessThan instance (lib/bigint.circom:16-60)
struct ZkBltLayout {
  u32 b_lt;     // bits: 17 x 2 words (Num2Bits(122) inputs)
  u32 f_eq;     // fr: 34 = 17 x (isz.out, isz.inv)
  u32 m_gates;  // small: ors[16], ands[16], eq_ands[16]
};
// image offsets of one FpMul instance (lib/fp.circom:16-81)
struct ZkFpMulLayout {
  u32 f_main;   // fr: v_ab[33], q[17], r[17], v_pq_r[33]
  u32 b_qr;     // bits: 34 x 2 words (q then r limbs)
  ZkBltLayout blt;
  u32 f_carry;  // fr: carry[33]
  u32 b_carry;  // bits: 32 x 3 words (carry + 2^130)
};
// main = FpMul(n, k), n k <= 62 (tests/test-circuits/fp-mul-test.circom; zkwg_fpmul_core.h)
struct ZkFpgLayout {
  u32 present, n, k;
  u32 in_a, in_b, in_p;  // record offsets of the k 16-byte chunks of a, b, p
  u32 m_out;             // small: out[k]
  u32 f_main;            // fr: v_ab[2k-1], q[k], r[k], v_pq_r[2k-1]
  u32 b_qr;              // bits: q_range_check[k], r_range_check[k] (one word each)
  u32 b_lt;              // bits: r_p_lt_check.lt[k].n2b (n + 1 bits, one word each)
  u32 f_eq;              // fr: r_p_lt_check.eq[k] (isz.out, isz.inv)
  u32 m_gates;           // small: ors, ands, eq_ands [k-1] each
  u32 f_carry;           // fr: tCheck.carry[2k-1]
  u32 b_carry;           // bits: tCheck.carryRangeChecks[2k-2] (one word each)
};
// RSAVerifier65537(121,17) (lib/rsa.circom:13-46)
struct ZkRsaLayout {
  u32 present;                 // 0 = circuit has no RSA block
  u32 in_mod, in_sig, in_msg;  // byte offsets in the input record
  u32 msg_from_digest;         // 1: message = header digest (EmailVerifier), 0: from the record
  u32 m_digest;                // small: header digest words (msg_from_digest)
  u32 b_modbits, b_msgbits;    // bits: 17 x 2 words each
  u32 m_modzero;               // small: 205 popcounts (IsZero inputs)
  u32 b_sigbits;               // bits: 17 x 2 words
  ZkBltLayout blt;             // signature < modulus
  ZkFpMulLayout mul[17];       // doublers[0..15], adder
};

// record field index of the generic-input-path range flags (include/zkwg.h ZKWG_IN_RANGE_FLAGS)
#define ZK_IN_RANGE_FLAGS 12

struct ZkSched {
  u32 main_kind;
  u32 n, k;
  u32 ignore_body;
  u32 nframes;           // SHA frames (0: header, 1: body)
  u32 total_blocks;      // sum of nblocks
  u32 hstates_per_email; // sum of (nblocks+1)
  u32 in_stride;         // bytes per input record
  u32 in_off[13];        // enum zkwg_input_field -> byte offset (12 = ZK_IN_RANGE_FLAGS)
  u32 n_public;
  u32 nsegs;
  u32 img_bits;          // u64 words per email
  u32 img_small;         // u32 words per email
  u32 img_fr;            // Fr elements per email
  u32 inv_half;          // inverse table covers d in [-inv_half, inv_half]
  u64 W;                 // witness length in field elements
  ZkShaFrame fr[2];
  // image offsets of main-level values
  u32 m_one;             // small: constant 1
  u32 m_hdr_len;         // small: emailHeaderLength / paddedInLength
  ZkRsaLayout rsa;
  ZkFpgLayout fpg;       // main = FpMul(n, k) with small parameters
  // EmailVerifier main (email-verifier.circom:42-174)
  u32 f_out;             // fr: pubkeyHash, shaHi, shaLo
  u32 f_pos;             // fr: 420 Poseidon S-box signals
  u32 body;              // 1: body-hash path present (ignoreBodyHashCheck != 1)
  u32 mask_header, mask_body;  // template flags enableHeaderMasking / enableBodyMasking
  u32 m_bh_idx;          // small: bodyHashIndex (also SelectRegexReveal.startIndex / VarShiftLeft.shift)
  u32 m_rev;             // small: bhReveal[max_header]
  u32 m_chars;           // small: bhBase64[44]
  u32 b_shift;           // bits: 1 word, bodyHashIndex (VarShiftLeft.n2b)
  u32 sel_bits;          // log2Ceil(max_header + 43)
  // BodyHashRegex DFA circuit (zkwg v1)
  u32 m_dfa_st;          // small: per position i of in[] (N+2 words): in[i] | st[i]<<8 | nx<<16 | st[i+1]<<24 (nx = delta(st[i],in[i]) from a non-zero state, else 255)
  u32 m_dfa_cm;          // small: per position, class truth mask of in[i]
  u32 m_dfa_pm;          // small: per position, primitive-test truth mask of in[i]
  u32 m_dfa_own;         // small: live_c1[nb], live_t[nb], prev_states0[NP][N], is_reveal0[N]
  u32 m_dfa_acc;         // small: number of positions in the accept state
  // BodyHashRegex loaded from a circom template (zkwg_circom.h): gate list evaluated by zk_net_eval
  u32 net_mode;          // 1: the regex circuit comes from a loaded template (the m_dfa_* fields are unused)
  u32 m_net;             // small: gate values, kept signals first (net_kept), then temporaries
  u32 m_net_out;         // small: the template's scalar output (bhRegexMatch)
  u32 net_kept;          // kept signals (= witness slots of the region)
  u32 net_total;         // kept + temporaries
  u32 net_steps;         // 64-record steps of the gate list
  u32 net_pins;          // zk_net_eval's LDS image: words holding gate values (then the message bytes, a zero, a scratch word)
  u32 net_lds_words;     // zk_net_eval's LDS image: total words
  u32 net_lds_masks;     // zk_net_eval's LDS image: first word of the per-byte mask region (byte-local frontier bits, zkwg_circom.h localize)
  u32 net_mask_words;    // mask words per message byte (0: none)
  u32 net_lanes;         // lanes per email of zk_net_eval (64 / net_lanes emails per wavefront)
  // the recurrences of the template collapsed to scans (zkwg_circom.h chain_pass; 0 positions: none)
  u32 net_chain_end;     // forward chain: bytes [0, net_chain_end)
  u32 net_chain_smax;    // its states (rows of every table)
  u32 net_chain_mw;      // mask words per position served from it (after the net_mask_words byte-local ones)
  u32 m_net_st;          // small: the forward state entering every position, one byte each (zk_net_scan)
  u32 net_bchain_end;    // backward chain: bytes [N - net_bchain_end, N)
  u32 net_bchain_smax, net_bchain_mw, net_bchain_fdim;   // its states, mask words (after the forward chain's), symbols / 256
  u32 m_net_bst;         // small: the backward state entering every position
  u32 m_net_pw;          // small: one position word per message byte, byte | fstate << 8 | bstate << 16 (zk_net_eval's prologue; read by zk_expand)
  // RemoveSoftLineBreaks(max_body) (template flag removeSoftLineBreaks, email-verifier.circom:148-156)
  u32 rslb;              // 1: present
  u32 rs_nch;            // 2 * max_body / 16 Poseidon(16) chunks of PoseidonModular(2 * max_body)
  u32 f_rs_chunk;        // fr: rs_nch chunk digests (not part of the witness)
  u32 f_rs_sum_enc;      // fr: sumEnc[max_body]
  u32 f_rs_rdec;         // fr: rDec[1 .. max_body)
  u32 f_rs_sum_dec;      // fr: sumDec[max_body]
  u32 f_rs_mux;          // fr: muxEnc: mux[0].out, then (c[0], mux.out) for i = 1 .. max_body-1
  u32 f_rs_hash;         // fr: rHasher S-box signals: chunk 0 (612), then per chunk (612 chunk, 243 merge)
  u32 f_rs_final;        // fr: final IsEqual (isz.out, isz.inv)
};
#define ZK_P16_KEPT 612u   // Poseidon(16): 3 * (8 * 17 + 68)
#define ZK_P2_KEPT 243u    // Poseidon(2):  3 * (8 * 3 + 57)
// offset (inside f_rs_hash) of chunk c's Poseidon(16) signals; its merge Poseidon(2) (c >= 1) follows at +612
ZK_HD u32 zk_rs_chunk_off(u32 c) { return c == 0 ? 0u : ZK_P16_KEPT + (c - 1u) * (ZK_P16_KEPT + ZK_P2_KEPT); }

// Raw DKIM results of a batch (device pointers) -- input of zk_gen_inputs; mirrors the fields of
// `DKIMVerificationResult` that generateEmailVerifierInputsFromDKIMResult reads.
struct ZkDkimBatch {
  const u8* headers;       // [n][header_stride] canonical signed header bytes
  const u32* header_len;   // [n]
  const u8* bodies;        // [n][body_stride] canonical body bytes
  const u32* body_len;     // [n]
  const u8* body_hash_b64; // [n][44] bodyHash (base64 text)
  const u8* pubkey_be;     // [n][256] RSA modulus, big-endian
  const u8* signature_be;  // [n][256] signature, big-endian
  const u8* selector;      // shaPrecomputeSelector bytes (shared by the batch) or NULL
  u32 header_stride, body_stride, selector_len;
};

#if defined(__HIPCC__)
// Device pointers of one launch (kernel argument, by value).
struct ZkBufs {
  const u8* in;          // packed input records
  u32* hst;              // SHA chaining states
  u64* bits;             // image: bit groups      [n_emails][img_bits]
  u32* small;            // image: small integers  [n_emails][img_small]
  Fr* frv;               // image: field elements  [n_emails][img_fr]
  const Fr* invtab;      // d^-1 for d in [-inv_half, inv_half]
  const Fr* pos_c;       // Poseidon(9) sparse-round table (zkwg_poseidon_sparse.h layout, t = 10, R_P = 60)
  const Fr* pos_m;       // Poseidon(9) dense tables for the wave-collective small-batch kernel: C[680], M[100] (Montgomery)
  const Fr* pos16;       // Poseidon(16) sparse-round table (zkwg_poseidon_sparse.h), removeSoftLineBreaks only
  const Fr* pos2;        // Poseidon(2)  sparse-round table
  const u32* pos16_l29;  // Poseidon(16) table in 29-bit limb form (zkwg_poseidon29.h): what zk_rslb_chunks reads
  const u32* pos2_l29;   // Poseidon(2) table in the same form: zk_rslb_merge1
  u32 rs_prio;           // zk_rslb_merge1 raises its wavefronts' issue priority (ZKWG_RSLB_MERGE_PRIO=0: off)
  u32* rs_stage;         // zk_rslb_chunks' dense-mix staging: word k of unit u at rs_stage[k * rs_units + u] (153 words per unit)
  u64 rs_units;          // emails x chunks, rounded up to whole wavefronts
  u32* rs_list;          // constant chunks (zkwg_kernels_rslb.hip): units whose 16 bytes are not all zero | units that are, rs_units entries each;
  u32* rs_cnt;           //   their two lengths (device counters, zeroed per batch).  rs_list = nullptr: every unit is hashed
  const Fr* rs_zero;     //   the 612 S-box signals + the digest of Poseidon(16)(0, ..., 0)
  const Fr* invtab_m;    // zk_expand_mont: the inverse table in Montgomery form
  const u32* net_records; // loaded regex template: 16 words per gate in execution order (zkwg_net_core.h)
  const u32* net_counts;  // loaded regex template: gates per step | flags (0x8000: 64-bit path)
  const u32* net_mask_tab; // loaded regex template: 256 x net_mask_words frontier masks by byte value
  const u8* net_cclass;   // forward chain: class of every position
  const u8* net_cdelta;   // [class][state][byte] next state
  const u32* net_cmask;   // [class][state][byte][net_chain_mw] mask words
  const u8* net_bclass;   // backward chain: the same with symbols (forward state, byte)
  const u8* net_bdelta;
  const u32* net_bmask;
  const Fr* rtab;        // zk_expand_mont: v * R mod r for v < 65536 (Montgomery-form output)
  Fr* frm;               // Montgomery-form output: per email, Montgomery copies of its img_fr field elements, then of the record's ZK_MONT_LIMBS limbs
  const ZkSeg* segs;     // segment table
  uint4* wit;            // output witnesses
  u64 wit_stride16;      // distance between consecutive witnesses in 16-byte units (>= 2 W; a caller may pad it)
  int* status;           // per-email status
  u32 n_emails;          // emails covered by the image arrays / this launch
  u32 e_first;           // zk_expand: first email to expand (wit points at its witness)
  u32 xcd_remap;         // zk_expand: 1 = workgroup -> unit mapping that gives each of the 8 XCDs one contiguous range
};

// Arguments of zk_expand (by value; deliberately small: a workgroup lives for one 8 KiB piece)
struct ZkX3 {
  const u8* in; const u64* bits; const u32* small; const Fr* frv; const Fr* invtab;
  const ZkPortionEntry* ent; const ZkSeg* segs;
  uint4* wit;
  const Fr* frm; const Fr* invtab_m; const Fr* rtab;   // Montgomery-form output only
  Fr* frm_w; u32* small_w; Fr* frv_w;                  // writable views (zk_image_to_mont, the O0 row kernels)
  u64 wit_stride16, W;
  u32 in_stride, img_bits, img_small, img_fr, inv_half, m_dfa_cm, m_dfa_pm, m_dfa_st;
  const struct ZkNetDec* netd;   // loaded regex template: how a slot of the region is decoded (device copy; NULL otherwise)
  u32 nportions, nsegs, e_first, n_count, xcd_remap, limb_off;   // nportions: pieces per witness (of 256 K slots)
};
#endif
