/*
 * zkwg.h -- C-ABI of the MI355X-native batched witness generator ("zkwg") for the
 * zk-email `EmailVerifier` circom circuit.
 *
 * This is the drop-in boundary: a plain C interface (pointers
 * and sizes only, no torch / HIP types) that a host binding (N-API addon, ctypes,
 * cgo ...) calls in place of the circom-generated WASM witness calculator.
 *
 * What each entry point replaces in the reference (paths relative to the reference
 * repository root):
 *
 *   zkwg_circuit_create      <- `wasm_tester(circuit.circom, {...})`
 *                                packages/circuits/tests/email-verifier.test.ts:21-31
 *                                (compiling `component main = EmailVerifier(...)`,
 *                                tests/test-circuits/email-verifier-test.circom:5,
 *                                rsa-test.circom:5, sha-test.circom:5) and the
 *                                `${circuitName}.wasm` argument of
 *                                packages/helpers/src/chunked-zkey.ts:80-84
 *   zkwg_calculate_batch     <- `circuit.calculateWitness(input)`
 *                                packages/circuits/tests/email-verifier.test.ts:43
 *                                and the first half of `snarkjs.groth16.fullProve`
 *                                (packages/helpers/src/chunked-zkey.ts:80), batched
 *   zkwg_calculate_batch_device  same, inputs/outputs already resident in HBM
 *   zkwg_pack_input          <- the `CircuitInput` object built by
 *                                packages/helpers/src/input-generators.ts:190-252
 *   zkwg_wtns_size / zkwg_write_wtns <- `snarkjs wtns calculate` / generate_witness.js
 *                                docs/zk-email-docs/UsageGuide/README.md:132-140
 *   zkwg_write_sym           <- the compiler's `.sym` file used by
 *                                circom_tester `assertOut` (email-verifier.test.ts:204)
 *   per-email status codes   <- circom_runtime exception codes; 4 = "Assert Failed"
 *                                (email-verifier.test.ts:73-79)
 *
 * All witness values are integers mod the BN254 scalar prime r
 * (packages/helpers/src/constants.ts:1), stored as 32-byte little-endian,
 * non-Montgomery field elements exactly as in a `.wtns` data section.
 *
 * Threading: the circuit description of a handle is immutable after creation.  The handle also
 * owns launch bookkeeping (timing event rings, the removeSoftLineBreaks merge-chain slots, the
 * host-path staging buffers); the device entry points take an internal lock around it and select
 * the handle's device themselves (the caller's current HIP device is restored on return), so a
 * handle may be shared by threads that launch on different streams.  Ordering between a
 * zkwg_prepare_device and the zkwg_expand_device that reads its scratch buffer is the caller's
 * (same stream, or an event).  zkwg_calculate_batch (host buffers) runs one call at a time per
 * handle.  No global state except zkwg_last_error (thread-local).
 */
#ifndef ZKWG_H
#define ZKWG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2 (round 3): ZKWG_IN_NFIELDS = 13 (the record's former padding word is ZKWG_IN_RANGE_FLAGS: hand-built records must
 * zero it, a non-zero word fails the email); zkwg_expand_device accepts out_stride >= 32 W (multiple of 16);
 * zkwg_scratch_bytes includes the Montgomery-copy area (and, round 5, 612 bytes per 16-byte body chunk of staging for the
 * removeSoftLineBreaks chunk hashes, behind the image arrays); zkwg_segment.pad is kernel-private.
 * Added since without breaking 2: zkwg_expand_host / zkwg_set_host_expand, zkwg_circuit_attach_r1cs / zkwg_expand_abc_device /
 * zkwg_expand_abc_host / zkwg_abc_bytes, ZKWG_MAIN_FP_MUL. */
#define ZKWG_ABI_VERSION 3

/* `component main = ...` choices (the reference's own test mains). */
enum zkwg_main_kind {
  ZKWG_MAIN_EMAIL_VERIFIER = 0, /* EmailVerifier(maxHeader,maxBody,n,k,ignoreBodyHashCheck,0,0,0), public [pubkey] */
  ZKWG_MAIN_SHA256_BYTES = 1,   /* Sha256Bytes(maxHeader), public [paddedIn, paddedInLength] (sha-test.circom) */
  ZKWG_MAIN_RSA_VERIFIER = 2,   /* RSAVerifier65537(n,k), public [modulus] (rsa-test.circom) */
  ZKWG_MAIN_FP_MUL = 3          /* FpMul(n,k), n*k <= 62, 2 <= k <= 17: inputs a, b, p in the pubkey / signature / message slots of the
                                   record (fp-mul-test.circom: FpMul(2,4); tests/fp-mul.test.ts:34-46) */
};

/* Witness layouts.  KEPT_V1 is the compact layout documented in DESIGN.md; SYM is KEPT_V1 re-ordered
 * (and thinned) to the witness indices of a circom `.sym` file, see zkwg_circuit_create_sym. */
enum zkwg_layout { ZKWG_LAYOUT_KEPT_V1 = 0, ZKWG_LAYOUT_SYM = 1 };

typedef struct zkwg_config {
  uint32_t main_kind;                /* enum zkwg_main_kind */
  uint32_t max_header;               /* maxHeadersLength (multiple of 64) */
  uint32_t max_body;                 /* maxBodyLength   (multiple of 64; 0 if unused) */
  uint32_t n;                        /* bits per RSA limb   (121) */
  uint32_t k;                        /* number of RSA limbs (17)  */
  uint32_t ignore_body_hash_check;   /* template flag, email-verifier.circom:42 */
  uint32_t enable_header_masking;    /* template flag: headerMask input, maskedHeader output */
  uint32_t enable_body_masking;      /* template flag: bodyMask input, maskedBody output */
  uint32_t remove_soft_line_breaks;  /* template flag: decodedEmailBodyIn input, RemoveSoftLineBreaks check */
  uint32_t layout;                   /* enum zkwg_layout */
} zkwg_config;

/* Fields of one packed input record (one record per email). */
enum zkwg_input_field {
  ZKWG_IN_HEADER = 0,         /* u8[max_header]  emailHeader / paddedIn            */
  ZKWG_IN_BODY = 1,           /* u8[max_body]    emailBody                         */
  ZKWG_IN_PRECOMPUTED_SHA = 2,/* u8[32]          precomputedSHA                    */
  ZKWG_IN_PUBKEY = 3,         /* 17 x 16-byte LE limbs: pubkey / modulus           */
  ZKWG_IN_SIGNATURE = 4,      /* 17 x 16-byte LE limbs                             */
  ZKWG_IN_MESSAGE = 5,        /* 17 x 16-byte LE limbs (RSA main only)             */
  ZKWG_IN_HEADER_LEN = 6,     /* u32 emailHeaderLength / paddedInLength            */
  ZKWG_IN_BODY_LEN = 7,       /* u32 emailBodyLength                               */
  ZKWG_IN_BODY_HASH_INDEX = 8,/* u32 bodyHashIndex                                 */
  ZKWG_IN_HEADER_MASK = 9,    /* u8[max_header]  headerMask (enable_header_masking)  */
  ZKWG_IN_BODY_MASK = 10,     /* u8[max_body]    bodyMask   (enable_body_masking)    */
  ZKWG_IN_DECODED_BODY = 11,  /* u8[max_body]    decodedEmailBodyIn (remove_soft_line_breaks) */
  ZKWG_IN_RANGE_FLAGS = 12,   /* u32 bit f = an element of field f did not fit its packed slot (generic path)  */
  ZKWG_IN_NFIELDS = 13
};

/* Per-email status: circom_runtime exception codes. */
enum zkwg_status {
  ZKWG_OK = 0,
  ZKWG_ERR_ASSERT_FAILED = 4 /* a `===` / assert failed: "Assert Failed" */
};

/* API return codes (negative = misuse / runtime failure). */
enum zkwg_rc {
  ZKWG_RC_OK = 0,
  ZKWG_RC_BAD_CONFIG = -1,
  ZKWG_RC_BAD_ARG = -2,
  ZKWG_RC_NO_DEVICE = -3,
  ZKWG_RC_HIP_ERROR = -4,
  ZKWG_RC_OOM = -5
};

typedef struct zkwg_circuit zkwg_circuit_t;

int zkwg_abi_version(void);
const char* zkwg_strerror(int rc_or_status);

/* Create / destroy.  `device` is a HIP device ordinal; device < 0 builds a
 * layout-only handle (no GPU needed: sizes, .sym, .wtns and packing still work,
 * zkwg_calculate_* return ZKWG_RC_NO_DEVICE). */
int zkwg_circuit_create(const zkwg_config* cfg, int device, zkwg_circuit_t** out);
/* Same, with the witness ordered as a compiled circuit's `.sym` file says (the file circom writes next
 * to the `.r1cs`/`.wasm`; circom_tester `loadSymbols`, packages/circuits/tests/email-verifier.test.ts:204,
 * and the order `groth16.prove(zkey, wtns)` needs, packages/helpers/src/chunked-zkey.ts:80).  `sym_text` =
 * the file's bytes: lines "labelIdx,witnessIdx,componentIdx,name", witnessIdx -1 = eliminated.  Every
 * signal the file keeps must be one this schedule produces (same qualified name); signals it eliminates
 * are dropped from the output.  `alias_text` (may be NULL): rename rules "ours=theirs", one per line,
 * applied to this library's names first (compiler-generated names of anonymous components).
 * On ZKWG_RC_BAD_CONFIG, zkwg_last_error() says which signal did not match.  cfg->layout is ignored. */
int zkwg_circuit_create_sym(const zkwg_config* cfg, int device, const char* sym_text, uint64_t sym_len,
                            const char* alias_text, uint64_t alias_len, zkwg_circuit_t** out);
/* Complete witnesses for a circuit compiled without full simplification (the reference documents
 * `circom ... --O0`, docs/zk-email-docs/UsageGuide/README.md:56-64): besides the `.sym` file the compiler's
 * `.r1cs` is given.  Every signal the file numbers beyond the ones this library's schedule produces (aliases,
 * constants, linear combinations -- 2.4 M of the 3.1 M signals of EmailVerifier(576,192) at O0) is derived from
 * the LINEAR constraints of the `.r1cs` (triangular elimination at creation).  On the device the file's witness is
 * written in ONE pass from the compact image (csrc/zkwg_o0.h): every wire has an 8-byte descriptor (the bit / byte /
 * image word it copies, or the result of a row), the rows that are real sums are evaluated into extensions of the
 * image by the row kernels at the end of zkwg_prepare_device, and zk_expand3_o0 streams all wires out -- no staging
 * buffer, no gather, launches of one handle need no ordering beyond prepare-before-expand of the same scratch buffer,
 * and Montgomery-form output works (zkwg_expand_montgomery_device).  Rates: DESIGN.md section 16.  Creation fails
 * with zkwg_last_error naming the first signal that is neither produced nor linearly defined (a quadratic
 * signal of a template this schedule does not implement, e.g. zk-regex's real BodyHashRegex).  Multi-
 * dimensional signal names (`a[t][k]`) are accepted by both `.sym` entry points. */
int zkwg_circuit_create_full(const zkwg_config* cfg, int device, const char* sym_text, uint64_t sym_len,
                             const char* alias_text, uint64_t alias_len, const uint8_t* r1cs, uint64_t r1cs_len,
                             zkwg_circuit_t** out);
uint64_t zkwg_linear_rows(const zkwg_circuit_t* c);                       /* derived signals of such a handle */
/* `.sym` layouts: out[s] = witness index of the default (kept-v1) layout's slot s, 0xffffffff if the file
 * eliminated it; returns the number of kept-v1 slots (0 for a handle without a `.sym`). */
uint64_t zkwg_layout_map(const zkwg_circuit_t* c, uint32_t* out, uint64_t cap);
/* BodyHashRegex from the template text.  EmailVerifier takes its regex circuit from the generated file
 * `@zk-email/zk-regex-circom/circuits/common/body_hash_regex.circom` (packages/circuits/email-verifier.circom:5,
 * 126-127), which is not part of the reference tree.  zkwg_circuit_create builds zkwg's own DFA circuit for that
 * regex; this entry builds the schedule FROM a supplied template instead: the file (and what it includes --
 * regex_helpers.circom, circomlib's comparators/gates/bitify; circomlib falls back to the restatement carried by
 * the library) is parsed and elaborated for msg_bytes = max_header, every hint / quadratic signal becomes a gate
 * evaluated on the device (kernel zk_net_eval), linear signals are substituted away.  Signal names follow the
 * compiler's (anonymous components `<T>_<line>_<offset>`), so a `.sym` (and `.r1cs`, see
 * zkwg_circuit_create_full) of the same circuit can be passed along; both may be NULL for the compact layout.
 * Supported subset and limits: zk-email-verify_amd/csrc/zkwg_circom.h. */
typedef struct zkwg_regex_source {
  const char* circom_path;    /* the generated template file                                              */
  const char* include_dirs;   /* ':'-separated directories searched for include "..." (may be NULL)       */
  const char* template_name;  /* NULL: "BodyHashRegex"; instantiated as template_name(max_header)(header) */
} zkwg_regex_source;
int zkwg_circuit_create_regex(const zkwg_config* cfg, int device, const zkwg_regex_source* regex,
                              const char* sym_text, uint64_t sym_len, const char* alias_text, uint64_t alias_len,
                              const uint8_t* r1cs, uint64_t r1cs_len, zkwg_circuit_t** out);
/* out[0..8): kept signals, temporaries, gates, assertion gates, chunks, 64-gate steps, LDS words that hold gate
 * values, gates on the evaluator's 64-bit path (handles created by zkwg_circuit_create_regex) */
int zkwg_regex_info(const zkwg_circuit_t* c, uint64_t out[8]);
int zkwg_linear_complete_host(const zkwg_circuit_t* c, uint8_t* witness); /* layout-only handles: host evaluation */
/* layout-only handles of a fully numbered circuit: the linear plan (copies + rows over the kept-v1 witness) on the host --
 * `out` (32 * zkwg_witness_len bytes) from one compact kept-v1 witness (the default layout of the same configuration) */
int zkwg_o0_gather_host(const zkwg_circuit_t* c, const uint8_t* kept_witness, uint8_t* out);
const char* zkwg_last_error(void);   /* detail of the calling thread's last ZKWG_RC_BAD_CONFIG */
void zkwg_circuit_destroy(zkwg_circuit_t* c);

/* Geometry. */
uint64_t zkwg_witness_len(const zkwg_circuit_t* c);   /* W: field elements per witness */
uint64_t zkwg_witness_bytes(const zkwg_circuit_t* c); /* 32 * W */
uint32_t zkwg_num_public(const zkwg_circuit_t* c);    /* outputs + public inputs: w[1..nPublic] */
uint64_t zkwg_input_stride(const zkwg_circuit_t* c);  /* bytes per packed input record */
uint64_t zkwg_input_offset(const zkwg_circuit_t* c, int field); /* byte offset inside a record */
uint64_t zkwg_scratch_bytes(const zkwg_circuit_t* c, uint64_t n_emails); /* device scratch for a batch */
/* The same without the Montgomery-copy area, which lies at the end of the buffer: enough for every entry point except the ones that
 * produce Montgomery-form values from this buffer (zkwg_expand_montgomery_device, zkwg_expand_abc_device with montgomery = 1) -- those
 * need zkwg_scratch_bytes.  For removeSoftLineBreaks = 1 the image is mostly field elements and this is half the size. */
uint64_t zkwg_scratch_bytes_standard(const zkwg_circuit_t* c, uint64_t n_emails);

/* Pack one email's inputs into a record (host helper; every pointer may be NULL
 * when the main kind does not use that field).  Limbs are 17 x 16-byte LE. */
int zkwg_pack_input(const zkwg_circuit_t* c, uint8_t* record,
                    const uint8_t* header, uint32_t header_len,
                    const uint8_t* body, uint32_t body_len,
                    const uint8_t* precomputed_sha, const uint8_t* pubkey_limbs,
                    const uint8_t* signature_limbs, const uint8_t* message_limbs,
                    uint32_t body_hash_index);
/* Generic input path: `CircuitInput` values are arbitrary field elements
 * (packages/helpers/src/input-generators.ts:6-18; circom_runtime normalises each one mod r and
 * stores it as a full 32-byte element), while the packed record keeps bytes / u32 / 128-bit limbs.
 * zkwg_pack_field takes `count` elements of input field `field` (ZKWG_IN_*) as 32-byte
 * little-endian integers (any 256-bit value; reduced mod r here), starting at element `first` of
 * that field, and writes them into the record.  An element that does not fit its slot keeps its
 * low bits and sets bit `field` of the record's ZKWG_IN_RANGE_FLAGS word: the kernels then fail the
 * email exactly where the circuit's own range check of that signal fails -- Num2Bits(8) of
 * lib/sha.circom:27,60,70, Num2Bits(log2Ceil(max)) of email-verifier.circom:58,116, Num2Bits(n) of
 * lib/rsa.circom:28,118,123, AssertBit of utils/bytes.circom:151-154, the RLC equality of
 * helpers/remove-soft-line-breaks.circom:124 -- status 4, "Assert Failed".  Call after
 * zkwg_pack_input (which clears the flags). */
int zkwg_pack_field(const zkwg_circuit_t* c, uint8_t* record, int field, uint64_t first,
                    const uint8_t* values32, uint64_t count);
/* headerMask / bodyMask of one record (flag variants; either pointer may be NULL). */
int zkwg_pack_masks(const zkwg_circuit_t* c, uint8_t* record, const uint8_t* header_mask, const uint8_t* body_mask);
/* decodedEmailBodyIn of one record (remove_soft_line_breaks = 1): max_body bytes, the output of
 * `removeSoftLineBreaks(bodyRemaining)` (packages/helpers/src/input-generators.ts:127-158, 241-244). */
int zkwg_pack_decoded_body(const zkwg_circuit_t* c, uint8_t* record, const uint8_t* decoded_body);

/* Host-buffer batch: H2D of `packed_inputs` (n_emails records), kernels, D2H of
 * n_emails witnesses (32*W bytes each, `out_stride` bytes apart; out_wtns may be
 * NULL to fetch only `status`).  Processes the batch in tiles of `max_tile` emails
 * (0 = 256, reduced if HBM is short); witness buffers are double-buffered so the
 * D2H copy of tile t overlaps the kernels of tile t+1.  One call at a time per handle. */
int zkwg_calculate_batch(zkwg_circuit_t* c, const uint8_t* packed_inputs, uint64_t n_emails,
                         uint8_t* out_wtns, uint64_t out_stride, int32_t* status,
                         uint64_t max_tile);

/* Batched input generation on the device (the step right before the path): restates
 * generateEmailVerifierInputsFromDKIMResult (packages/helpers/src/input-generators.ts:190-252) with
 * sha256Pad / generatePartialSHA (sha-utils.ts:30-111) and toCircomBigIntBytes (binary-format.ts:71-83).
 * All pointers are device pointers; `selector` is the optional shaPrecomputeSelector (shared by the
 * batch).  d_records receives n packed input records (zkwg_input_stride each); d_gen_status[i]:
 *   0 ok | 1 header does not fit maxHeadersLength | 2 remaining body longer than maxBodyLength
 *   | 3 selector not found in the body | 4 body longer than body_stride.   EmailVerifier main only. */
typedef struct zkwg_dkim_batch {
  const uint8_t* headers;        /* [n][header_stride] canonical signed header bytes */
  const uint32_t* header_len;    /* [n] */
  const uint8_t* bodies;         /* [n][body_stride] canonical body bytes (NULL if ignoreBodyHashCheck) */
  const uint32_t* body_len;      /* [n] */
  const uint8_t* body_hash_b64;  /* [n][44] bodyHash, base64 text */
  const uint8_t* pubkey_be;      /* [n][256] RSA modulus, big-endian */
  const uint8_t* signature_be;   /* [n][256] signature, big-endian */
  const uint8_t* selector;       /* shaPrecomputeSelector bytes or NULL */
  uint32_t header_stride, body_stride, selector_len;
} zkwg_dkim_batch;
int zkwg_generate_inputs_device(zkwg_circuit_t* c, const zkwg_dkim_batch* batch, uint64_t n_emails,
                                void* d_records, void* d_gen_status, void* hip_stream);

/* Device-resident batch below the boundary: for a host without a tensor library (the Node host of the
 * reference) that feeds a device-side consumer.  The records go to the device once; the compute kernels of sub-batch i + 1
 * (`prep` emails, default 1024) run on one internal stream while sub-batch i is expanded tile by tile (`tile` emails, default
 * 512) on another; witnesses are written into a ring of TWO tiles in HBM that is overwritten as the batch proceeds -- the
 * ring is placed where HBM takes the stores fastest (once per handle: the first tile's expansion is timed into spare
 * candidate buffers, zkwg_resident_placement reports the timings; ZKWG_PLACE_RING=0 disables).  After each tile's expansion
 * has been enqueued `consumer` (may be NULL) is called on the calling thread with the tile's device pointer, the distance
 * between consecutive witnesses, the index of its first email, its email count and the HIP stream the expansion was
 * enqueued on: work the consumer enqueues on that stream sees the complete tile and finishes before the ring slot is reused
 * (a GPU prover's entry point; packages/helpers/src/chunked-zkey.ts:80-84 is the call it completes).  What comes back to the
 * host: status[n] and, if `table` is given, n rows of 100 bytes {status i32, pubkeyHash, shaHi, shaLo} -- never the witnesses
 * (56.9 MB each: use zkwg_calculate_batch for that, PCIe-bound).  zkwg_calculate_batch_multi with out_wtns = NULL runs this
 * on every device. */
typedef void (*zkwg_tile_fn)(void* user, int device, const void* d_tile, uint64_t witness_stride, uint64_t first_email,
                             uint64_t count, void* hip_stream);
int zkwg_calculate_batch_resident(zkwg_circuit_t* c, const uint8_t* packed_inputs, uint64_t n_emails, int32_t* status,
                                  uint8_t* table, uint64_t tile, uint64_t prep, zkwg_tile_fn consumer, void* user);
int zkwg_resident_placement(const zkwg_circuit_t* c, float* ms, int cap, int kept[2]);
/* Give the buffers zkwg_calculate_batch_resident keeps in the handle (records, scratch, the two-tile witness ring, statuses) back to the
 * device; the next call allocates and places them again.  For a service that shares the GPU between circuits. */
int zkwg_resident_release(zkwg_circuit_t* c);


/* Host expansion (the delivered-to-host rate; the consumer is snarkjs on the host,
 * packages/helpers/src/chunked-zkey.ts:80-84).  A witness crosses PCIe at 32 bytes per signal (56.9 MB per email) although
 * its information is the 0.45 MB image the prepare kernels leave.  zkwg_expand_host runs the segment decoders of
 * zk_expand on the host (same source, csrc/zkwg_expand_dec.h) over an image that was copied to host memory:
 * `records` = the n packed input records, `scratch_host` = a host copy of the scratch buffer of
 * zkwg_prepare_device(c, ., n, ...) (zkwg_scratch_bytes(c, n) bytes; only the image arrays are read), emails
 * [first, first + count) go to `out` (16-byte aligned), written with non-temporal stores by `threads` host threads.
 * Not available for zkwg_circuit_create_full handles (their row results are computed on the device).
 * zkwg_set_host_expand(c, threads > 0) makes zkwg_calculate_batch use this route: per tile H2D records -> prepare
 * kernels -> D2H of the image, the host expanding tile t while the device prepares tile t + 1 (0 = device expansion + D2H
 * of the witnesses, the default).  Both routes give identical bytes. */
int zkwg_expand_host(const zkwg_circuit_t* c, const uint8_t* records, uint64_t n_emails, const uint8_t* scratch_host,
                     uint64_t first, uint64_t count, uint8_t* out, uint64_t out_stride, int threads);
int zkwg_set_host_expand(zkwg_circuit_t* c, int threads);

/* A HIP stream (returned as void*, NULL on failure) restricted to the compute units whose bit is set in `mask`
 * (`words` x 32 bits, CU i = bit i; hipExtStreamCreateWithCUMask).  Optional tuning aid: the prepare kernels are latency-bound
 * and few; confined to a few CUs they do not take wave slots from zk_expand on the others (bench.py --prep-cus). */
void* zkwg_stream_create_masked(int device, const uint32_t* mask, int words);
void zkwg_stream_destroy(void* stream);

/* Page-locked host memory for `out_wtns` (optional): with it the D2H copy of one tile overlaps the
 * kernels of the next tile. */
void* zkwg_alloc_pinned(uint64_t bytes);
void zkwg_free_pinned(void* p);

/* Device-resident batch: every pointer is a device pointer on the handle's
 * device; `hip_stream` is a hipStream_t passed as void* (NULL = default stream).
 * d_scratch must hold zkwg_scratch_bytes(c, n_emails).  Asynchronous. */
int zkwg_calculate_batch_device(zkwg_circuit_t* c, const void* d_packed_inputs, uint64_t n_emails,
                                void* d_out_wtns, uint64_t out_stride, void* d_status,
                                void* d_scratch, void* hip_stream);

/* The two phases of zkwg_calculate_batch_device, separately launchable (e.g. on different
 * streams, with different granularity):
 *   zkwg_prepare_device  runs every compute kernel for n_emails and leaves their compact
 *                        images in d_scratch (zkwg_scratch_bytes(c, n_emails)) + d_status;
 *   zkwg_expand_device   streams the witnesses of emails [first, first+count) of that prepared
 *                        batch into d_out_wtns (count * 32 W bytes) -- the HBM-write-bound kernel.
 * The images are ~1 % of the witness size, so a whole batch can be prepared at once while its
 * witnesses are expanded tile by tile into a smaller output ring.
 * remove_soft_line_breaks = 1: the serial Poseidon merge chain of a prepared batch runs on an internal
 * side stream (it is latency-bound and would stall the caller's stream for ~0.2 s); d_status and the
 * scratch images are complete once a zkwg_expand_device call on that scratch buffer has been ordered
 * (it waits for the chain), so keep at most 16 scratch buffers in flight per handle. */
int zkwg_prepare_device(zkwg_circuit_t* c, const void* d_packed_inputs, uint64_t n_emails, void* d_status,
                        void* d_scratch, void* hip_stream);
/* (out_stride: distance between consecutive witnesses in d_out_wtns, >= 32 * W and a multiple of 16) */
int zkwg_expand_device(zkwg_circuit_t* c, const void* d_packed_inputs, uint64_t n_emails, const void* d_scratch,
                       uint64_t first, uint64_t count, void* d_out_wtns, uint64_t out_stride, void* hip_stream);

/* Throttle for pipelined use: when zkwg_prepare_device runs concurrently with zkwg_expand_device of
 * the previous batch, cap the resident zk_rsa wavefronts per CU (each holds 164 VGPRs) so that the
 * HBM-bound expand kernel keeps its occupancy.  0 = no cap (default; lowest prepare latency). */
int zkwg_set_prepare_throttle(zkwg_circuit_t* c, int rsa_wavefronts_per_cu);
/* Measurement aid (tools/beside.py): zkwg_prepare_device launches only the kernels whose bit is
 * set -- bit 0 zk_sha_chain, 1 zk_sha_trace, 2 zk_net_eval, 3 zk_misc_ev, 4 zk_rsa, 5 zk_poseidon9*, 6 zk_rslb_chunks,
 * 7 zk_rslb_chain, 8 the row kernels of numbered / constraint-attached handles.  Default 0xffffffff (everything); with a
 * partial mask the images of the batch are NOT complete, so only timing may be taken from such a call. */
int zkwg_set_prepare_mask(zkwg_circuit_t* c, uint32_t kernel_mask);

/* Time the dominant kernel(s) of the last zkwg_calculate_batch_device call with
 * HIP events recorded on the launch stream.  Returns ms in *ms for kernel index
 * `which` (see zkwg_kernel_name); negative rc if timing was not enabled. */
int zkwg_set_timing(zkwg_circuit_t* c, int enable);
int zkwg_last_kernel_ms(zkwg_circuit_t* c, int which, float* ms);
/* Sum of kernel `which`'s durations over the launches recorded since zkwg_set_timing(c, 1)
 * (at most the last 512 launches), and how many launches that sum covers. */
int zkwg_timing_summary(zkwg_circuit_t* c, int which, float* total_ms, uint32_t* launches);
int zkwg_num_kernels(const zkwg_circuit_t* c);
const char* zkwg_kernel_name(const zkwg_circuit_t* c, int which);
/* Witness slots (field elements) written per email by kernel `which`. */
uint64_t zkwg_kernel_slots(const zkwg_circuit_t* c, int which);

/* `.wtns` container: 4-byte magic "wtns", u32 version 2,
 * u32 nSections 2; section 1: u32 n8=32, 32-byte LE prime, u32 nWitness;
 * section 2: nWitness x 32-byte LE values. */
uint64_t zkwg_wtns_size(const zkwg_circuit_t* c);
int zkwg_write_wtns(const zkwg_circuit_t* c, const uint8_t* witness, uint8_t* out, uint64_t out_cap);

/* Symbol table of the layout in force, one line per witness slot: "slot,slot,0,name\n"
 * (the `.sym` line format `labelIdx,varIdx,componentIdx,name`).  Returns bytes
 * needed; writes at most cap bytes. */
uint64_t zkwg_write_sym(const zkwg_circuit_t* c, char* out, uint64_t cap);

/* ---- `.r1cs` reader + constraint check on the device ----------------------------
 * Replaces circom_tester's `await circuit.checkConstraints(witness)`, which every circuit test of the
 * reference calls right after calculateWitness (packages/circuits/tests/email-verifier.test.ts:44,
 * sha.test.ts, rsa.test.ts, ...): for each constraint of the compiled circuit, (A.w)*(B.w) == C.w mod r.
 * `bytes` = the iden3 `.r1cs` file circom wrote (field must be BN254 Fr).  device < 0: parse only. */
typedef struct zkwg_r1cs zkwg_r1cs_t;
int zkwg_r1cs_load(const uint8_t* bytes, uint64_t len, int device, zkwg_r1cs_t** out);
void zkwg_r1cs_destroy(zkwg_r1cs_t* r);
/* out = {nWires, nPubOut, nPubIn, nPrvIn, mConstraints, nLabels} */
int zkwg_r1cs_info(const zkwg_r1cs_t* r, uint64_t out[6]);
/* n witnesses resident in HBM (32-byte LE values, `stride` bytes apart, stride >= 32*nWires):
 * d_first_bad[i] (u64) = index of the first violated constraint of witness i, UINT64_MAX if all hold.
 * A value >= r anywhere in a checked linear combination also counts as a violation. */
int zkwg_check_constraints_device(zkwg_r1cs_t* r, const void* d_witness, uint64_t n, uint64_t stride,
                                  void* d_first_bad, void* hip_stream);
/* Same for witnesses in host memory (staged through the device in tiles). */
int zkwg_check_constraints(zkwg_r1cs_t* r, const uint8_t* witness, uint64_t n, uint64_t stride, uint64_t* first_bad);
/* First stage of a Groth16 prover on the device-resident witness (what snarkjs' `groth16.prove` does first with the
 * witness: the evaluations A.w, B.w, C.w of every constraint; second half of `fullProve`,
 * packages/helpers/src/chunked-zkey.ts:80): d_abc[e] receives 3 * nConstraints field elements -- the A values, then
 * B, then C -- `abc_stride` (>= 96 * nConstraints) bytes apart.  A witness in Montgomery form
 * (zkwg_expand_montgomery_device; pass montgomery = 1) yields evaluations in Montgomery form, a standard-form witness
 * (montgomery = 0) standard-form ones; nothing is converted or copied to the host.  The flag only selects a shortcut
 * (a Montgomery-form 0 or 1 needs no product); the arithmetic is the same.  NTT / MSM are not part of this library. */
int zkwg_r1cs_evaluate_device(zkwg_r1cs_t* r, const void* d_witness, uint64_t n, uint64_t stride, int montgomery,
                              void* d_abc, uint64_t abc_stride, void* hip_stream);

/* ---- prover hand-off -------------------------------------------------------------
 * The step after this path is `groth16.prove(zkey, wtns)` (second half of fullProve,
 * packages/helpers/src/chunked-zkey.ts:80).  A device-resident prover takes the witness where
 * zkwg_expand_device left it; this converts n_values 32-byte field elements in place between the
 * `.wtns` standard form and Montgomery form (x * 2^256 mod r), whichever its NTT / MSM kernels want. */
int zkwg_convert_montgomery_device(void* d_values, uint64_t n_values, int to_montgomery, void* hip_stream);
/* The fused form: zkwg_expand_device that writes every witness value directly as x * 2^256 mod r
 * (0 -> 0, 1 -> R, values below 2^16 from a table, the rest one Montgomery product) -- the witness
 * is still written exactly once and never read back, so handing it to a device prover costs no extra
 * HBM pass.  Same arguments and layout as zkwg_expand_device; bit-identical to zkwg_expand_device
 * followed by zkwg_convert_montgomery_device(.., 1, ..).  Needs the default portion size. */
int zkwg_expand_montgomery_device(zkwg_circuit_t* c, const void* d_packed_inputs, uint64_t n_emails,
                                  const void* d_scratch, uint64_t first, uint64_t count, void* d_out_wtns,
                                  uint64_t out_stride, void* hip_stream);
/* ---- prover stage 2 (SURVEY.md 8f4 "next"): the transforms between A.w | B.w | C.w and the H multi-exponentiation ----------
 * snarkjs groth16_prove.js [EXT, pinned in yarn.lock:7767-7800; ffjavascript F1Field]: A = ifft(A.w), Aodd = A[i] * inc^i,
 * Aodd_T = fft(Aodd) (likewise B, C), P_T = Aodd_T * Bodd_T - Codd_T on the domain of 2^power >= nConstraints + nPublic + 1
 * points, inc = the primitive 2^(power+1)-th root w[power + 1] (w[28] = 5^((r-1)/2^28), w[i] = w[i+1]^2).  Everything is in
 * Montgomery form (x * 2^256 mod r), as zkwg_expand_abc_device(montgomery = 1) writes it and as an MSM wants it.
 * zkwg_ntt_create builds the twiddle and coset tables of one domain size (2 x 32 bytes x 2^log2_n on the device).
 * zkwg_h_evaluations_device: d_abc = n_emails records of A.w | B.w | C.w (n_constraints values each, abc_stride bytes apart,
 * e.g. straight from zkwg_expand_abc_device); d_work = zkwg_ntt_work_bytes(plan, n_emails) bytes of scratch; d_out receives
 * 2^log2_n values per email, out_stride bytes apart.  n_emails <= 21845 per call.  Arithmetic-bound (~12 Montgomery products
 * per element and transform), not HBM-bound: DESIGN.md section 22.
 * zkwg_ntt_transform_device: stand-alone in-place transforms of n_polys arrays (natural order in and out; inverse includes
 * 1 / n) -- ffjavascript's Fr.fft / Fr.ifft on Montgomery-form data. */
typedef struct zkwg_ntt zkwg_ntt_t;
int zkwg_ntt_create(int device, uint32_t log2_n, zkwg_ntt_t** out);
void zkwg_ntt_destroy(zkwg_ntt_t* plan);
uint64_t zkwg_ntt_domain(const zkwg_ntt_t* plan);
uint64_t zkwg_ntt_work_bytes(const zkwg_ntt_t* plan, uint64_t n_emails);
int zkwg_ntt_transform_device(zkwg_ntt_t* plan, void* d_data, uint64_t n_polys, int inverse, void* hip_stream);
int zkwg_h_evaluations_device(zkwg_ntt_t* plan, const void* d_abc, uint64_t abc_stride, uint64_t n_constraints, uint64_t n_emails,
                              void* d_work, void* d_out, uint64_t out_stride, void* hip_stream);

/* Device buffers mapped from physical chunks (HIP virtual-memory API; chunk_bytes = 0: 1 GiB): the allocator for the output ring of
 * zkwg_expand_device and any other large store target.  A buffer from hipMalloc may take zk_expand's stores 12-20 % slower than
 * another of the same size (round 4); a buffer mapped chunk by chunk does not (tools/chunkbench.hip, profiles/r05/r05_e_chunkbench.txt)
 * -- no spare candidates, no transient memory.  The pointer is an ordinary device pointer; free it with zkwg_device_free_chunked. */
int zkwg_device_alloc_chunked(int device, uint64_t bytes, uint64_t chunk_bytes, void** out);
/* the same with `extra` spare candidate chunks: every candidate takes a probe fill of zk_expand's store shape on a mapping of its own,
 * the fastest are kept and mapped back to back, the others released (transient: `extra` chunks).  rates[0 .. *n_rates): GB/s of the
 * candidates in creation order (rates may be NULL). */
int zkwg_device_alloc_chunked_ex(int device, uint64_t bytes, uint64_t chunk_bytes, uint32_t extra, void** out, float* rates, uint32_t cap,
                                 uint32_t* n_rates);
int zkwg_device_free_chunked(void* ptr);

/* ---- prover stage 3: the multi-exponentiations of groth16_prove.js (reference call site:
 * packages/helpers/src/chunked-zkey.ts:80-84, the second half of fullProve) -----------------------------------------------
 * pi_a, pib1, pi_c and resH are  sum_i scalar_i * base_i  over BN254 G1 with the zkey's bases (affine, x | y in Montgomery form,
 * 64 bytes each, the point at infinity all zeros -- sections 5, 6, 8, 9 of the file), pi_b the same over G2 (section 7: x.c0 | x.c1 |
 * y.c0 | y.c1, 128 bytes), with the witness resp. the H evaluations of zkwg_h_evaluations_device as scalars (32 bytes each,
 * standard or Montgomery form).  zkwg_msm_create / _create_g2 upload the bases of one such sum (window_bits = 0: chosen from n);
 * zkwg_msm_create_device takes bases already in device memory (group 1 / 2; not owned).  zkwg_msm_g1_device / _g2_device compute
 * the sum for the n scalars at d_scalars, d_work = zkwg_msm_work_bytes bytes of scratch (256-byte aligned), and return the point as
 * the zkey would store it (affine, Montgomery form, 64 / 128 bytes; zeros = infinity).  ones_apart = 1 for witness scalars (mostly
 * 0 / 1: the bases with scalar 1 are summed by a plain reduction tree instead of all landing in one bucket).  Bucket method, signed
 * windows, XYZZ accumulators: DESIGN.md section 23.  Scalars must be CANONICAL (below the group order r; a Montgomery-form scalar is
 * reduced by its conversion): the signed windows cover 255 bits, a value of 2^255 or more would lose its top carry.
 * zkwg_fixed_base_device: d_out[i] = scalar_i * G for the group's generator (G1: (1, 2); G2: the EIP-197 generator) -- how a key
 * with a KNOWN trapdoor becomes bases (tests, tools/bench_prove.py; a real key comes from its .zkey). */
typedef struct zkwg_msm zkwg_msm_t;
int zkwg_msm_create(int device, const uint8_t* bases, uint64_t n, int window_bits, zkwg_msm_t** out);
int zkwg_msm_create_g2(int device, const uint8_t* bases, uint64_t n, int window_bits, zkwg_msm_t** out);
int zkwg_msm_create_device(int device, int group, const void* d_bases, uint64_t n, int window_bits, zkwg_msm_t** out);
void zkwg_msm_destroy(zkwg_msm_t* plan);
uint64_t zkwg_msm_work_bytes(const zkwg_msm_t* plan);
int zkwg_msm_window_bits(const zkwg_msm_t* plan);
int zkwg_msm_group(const zkwg_msm_t* plan);
int zkwg_msm_g1_device(zkwg_msm_t* plan, const void* d_scalars, int scalars_montgomery, int ones_apart, void* d_work, uint8_t* out_xy,
                       void* hip_stream);
int zkwg_msm_g2_device(zkwg_msm_t* plan, const void* d_scalars, int scalars_montgomery, int ones_apart, void* d_work, uint8_t* out_xy,
                       void* hip_stream);
/* asynchronous form: the sum is left at d_out_xyzz as an accumulator in XYZZ coordinates (Montgomery form; 128 bytes for a G1 plan,
 * 256 for a G2 plan) and nothing is synchronised -- the sums of several proofs can be in flight on several streams, which is what
 * hides the serial tail of each.  zkwg_msm_finish_host: n downloaded accumulators -> n points as the zkey stores them. */
int zkwg_msm_enqueue_device(zkwg_msm_t* plan, const void* d_scalars, int scalars_montgomery, int ones_apart, void* d_work, void* d_out_xyzz,
                            void* hip_stream);
int zkwg_msm_finish_host(int group, const uint8_t* xyzz, uint64_t n, uint8_t* out_points);
/* Round 6: emails are the parallel axis of the sums too.  One launch series sums n_emails scalar vectors (scalar_stride bytes apart,
 * a multiple of 32) against the plan's bases and leaves n_emails accumulators at d_out_xyzz (128 / 256 bytes each, consecutive); d_work:
 * zkwg_msm_work_bytes_batch(plan, n_emails) bytes.  zkwg_msm_create_ex: window_bits (0: from n), slice0 = bucket entries per lane at the
 * first slice level (0: 16; 64 for 2^20 full-size scalars), table_budget = bytes the plan may spend on its K shifted copies of the bases
 * (0: what is free minus 4 GiB; too small: the classic layout, K bucket sets and a Horner pass -- zkwg_msm_precomputed tells).
 * For witness scalars one classification pass serves several plans: zkwg_msm_classify_device writes, for up to three plans whose bases
 * cover the scalars [first[t], first[t] + n_t), the per-email index lists of the scalars that are 1 and of the others that are not 0
 * (bases at infinity left out) into d_lists[t] (zkwg_msm_lists_bytes(plan, n_emails) bytes, 256-byte aligned), and
 * zkwg_msm_enqueue_lists_device sums with them (d_work: n_emails * zkwg_msm_work_bytes(plan) bytes). */
int zkwg_msm_create_ex(int device, int group, const void* bases, int bases_on_device, uint64_t n, int window_bits, int slice0, uint64_t table_budget,
                       zkwg_msm_t** out);
uint64_t zkwg_msm_work_bytes_batch(const zkwg_msm_t* plan, uint64_t n_emails);
uint64_t zkwg_msm_lists_bytes(const zkwg_msm_t* plan, uint64_t n_emails);
uint64_t zkwg_msm_estimate_work_bytes(int group, uint64_t n, int window_bits, int slice0);   /* per email, before a plan exists */
uint64_t zkwg_msm_table_bytes(const zkwg_msm_t* plan);
int zkwg_msm_precomputed(const zkwg_msm_t* plan);
int zkwg_msm_enqueue_batch_device(zkwg_msm_t* plan, const void* d_scalars, uint64_t scalar_stride, uint64_t n_emails, int scalars_montgomery, int ones_apart,
                                  void* d_work, void* d_out_xyzz, void* hip_stream);
int zkwg_msm_classify_device(zkwg_msm_t* const* plans, const uint64_t* first, uint32_t n_plans, const void* d_scalars, uint64_t scalar_stride, uint64_t n_scalars,
                             uint64_t n_emails, int scalars_montgomery, int ones_apart, void* const* d_lists, void* hip_stream);
int zkwg_msm_enqueue_lists_device(zkwg_msm_t* plan, const void* d_scalars, uint64_t scalar_stride, uint64_t n_emails, int scalars_montgomery, const void* d_lists,
                                  int with_ones, void* d_work, void* d_out_xyzz, void* hip_stream);
int zkwg_fixed_base_device(int device, int group, const void* d_scalars, uint64_t n, void* d_out, void* hip_stream);
/* pi_a, pi_b, pi_c from the five sums of one proof and the key's alpha / beta / delta points (groth16_prove.js: pi_a = alpha1 + sum_a +
 * r delta1, pi_b = beta2 + sum_b2 + s delta2, pi_c = sum_c + sum_h + s pi_a + r (beta1 + sum_b1 + s delta1) - r s delta1).  Points in
 * as the zkey stores them (affine, Montgomery form); r, s = the blinding scalars (32 bytes, little-endian, below the group order);
 * points out in STANDARD form, little-endian x | y (G2: x.c0 | x.c1 | y.c0 | y.c1): the integers of snarkjs' proof.json.  Host
 * arithmetic (a dozen group operations).  ZKWG_RC_BAD_ARG for a point that is not on its curve. */
int zkwg_groth16_assemble(const uint8_t* sum_a, const uint8_t* sum_b1, const uint8_t* sum_b2, const uint8_t* sum_c, const uint8_t* sum_h,
                          const uint8_t* vk_alpha1, const uint8_t* vk_beta1, const uint8_t* vk_beta2, const uint8_t* vk_delta1,
                          const uint8_t* vk_delta2, const uint8_t* r32, const uint8_t* s32, uint8_t* pi_a, uint8_t* pi_b, uint8_t* pi_c);

/* ---- the prover as one call (reference: the second half of snarkjs.groth16.fullProve, packages/helpers/src/chunked-zkey.ts:80-84) ----
 * zkwg_prover_create plans everything a circuit's proofs need once: it attaches the constraint system `r1cs` (over the handle's
 * witness layout, WITH the nPublic + 1 rows snarkjs appends to A; n_rows = its constraint count; NULL if already attached) to `c`,
 * builds the transform plan of the key's domain and the five multi-exponentiation plans over the key's bases (host pointers to
 * sections 5-9 of the .zkey, or device pointers with bases_on_device = 1), and allocates the buffers of `slots` proofs in flight:
 * 1-3 contexts of E = zkwg_prover_emails_per_series emails each (round 6: every stage is one launch series for the E emails of a
 * context; the contexts roll, nothing depends on the number of hardware queues).  zkwg_prover_prove_prepared: proofs of emails indices[0 .. n_idx) of a batch prepared with
 * zkwg_prepare_device(c, d_in, n, ..., d_scratch); blinding = n_idx x (r | s), 32-byte little-endian scalars below the group order
 * (random per proof); out_proofs = n_idx x 256 bytes: pi_a (x | y) | pi_b (x.c0 | x.c1 | y.c0 | y.c1) | pi_c (x | y), standard-form
 * little-endian integers -- the numbers of snarkjs' proof.json.  zkwg_prover_prove_batch: the same from n packed input records on the
 * host (status[i] = circom_runtime code of email i; the proof bytes of a failed email are zero). */
typedef struct zkwg_prover zkwg_prover_t;
typedef struct zkwg_proving_key {
  uint64_t n_wires, n_public, log2_domain;
  const void *a, *b1, *b2, *c, *h;            /* bases: n_wires G1, n_wires G1, n_wires G2, n_wires - n_public - 1 G1, 2^log2_domain G1 */
  int bases_on_device;                        /* 0: host memory (uploaded here), 1: device memory (not owned) */
  uint8_t alpha1[64], beta1[64], beta2[128], delta1[64], delta2[128];   /* as the zkey's header stores them */
} zkwg_proving_key;
int zkwg_prover_create(zkwg_circuit_t* c, int device, const uint8_t* r1cs, uint64_t r1cs_len, uint64_t n_rows, const zkwg_proving_key* key,
                       uint32_t slots, zkwg_prover_t** out);
/* The prover from the zkey ALONE, as `groth16.prove(zkey, wtns)` / `fullProve(input, wasm, zkey)` take it (chunked-zkey.ts:80-84): the
 * rows of A and B (the nPublic + 1 public rows included) come from section 4 of the file, C.w = A.w o B.w as snarkjs' buildABC1 forms
 * it, the bases from sections 5-9; `zkey` is only read during the call.  The handle's witness layout must be the key's (nVars wires). */
int zkwg_prover_create_zkey(zkwg_circuit_t* c, int device, const uint8_t* zkey, uint64_t zkey_len, uint32_t slots, zkwg_prover_t** out);
void zkwg_prover_destroy(zkwg_prover_t* p);
uint32_t zkwg_prover_emails_per_series(const zkwg_prover_t* p);
uint32_t zkwg_prover_contexts(const zkwg_prover_t* p);
int zkwg_prover_prove_prepared(zkwg_prover_t* p, const void* d_in, uint64_t n, const void* d_scratch, const uint64_t* indices, uint64_t n_idx,
                               const uint8_t* blinding, uint8_t* out_proofs);
int zkwg_prover_prove_batch(zkwg_prover_t* p, const uint8_t* packed, uint64_t n, const uint8_t* blinding, int32_t* status, uint8_t* out_proofs);

/* The same prover stage without a 32-byte witness in between: the constraint system `r1cs` (its wires = the handle's
 * witness layout: built-in, `.sym`, or -- since ABI 3 -- a fully numbered handle of zkwg_circuit_create_full, whose system is
 * the compiler's own `.r1cs`, the file a zkey is keyed to: every wire of every combination is substituted by the kept-v1
 * signal(s) it derives from) is attached to the handle once -- BEFORE the
 * first zkwg_scratch_bytes / zkwg_prepare_device call that should serve it, the compact image grows by the results of the
 * combinations that are genuine sums -- and zkwg_expand_abc_device then writes A.w | B.w | C.w (3 * nConstraints values,
 * `abc_stride` >= zkwg_abc_bytes apart) of emails [first, first + count) straight from the prepared image: a combination
 * that is one wire is written like that wire, sums of bits and small integers are 64-bit integer rows, only the rest is
 * arithmetic mod r.  Bit-identical to zkwg_r1cs_evaluate_device on the expanded witness (montgomery = 1: to that of the
 * Montgomery-form witness).  Replaces the head of snarkjs' groth16.prove (packages/helpers/src/chunked-zkey.ts:80). */
int zkwg_circuit_attach_r1cs(zkwg_circuit_t* c, const uint8_t* r1cs, uint64_t len);
uint64_t zkwg_abc_bytes(const zkwg_circuit_t* c);
int zkwg_expand_abc_device(zkwg_circuit_t* c, const void* d_packed_inputs, uint64_t n_emails, const void* d_scratch,
                           uint64_t first, uint64_t count, int montgomery, void* d_abc, uint64_t abc_stride, void* hip_stream);
/* The same values (standard form) written by the host from a host copy of the scratch buffer, like zkwg_expand_host for the
 * witness: rows_on_host = 0 -- the image was prepared by the device with the system already attached (only the image crosses
 * PCIe); rows_on_host = 1 -- the row tables are evaluated on the host too (scratch_host is written; the complete path of a
 * layout-only handle, used by the CPU tests).  out: 16-byte aligned, out_stride >= zkwg_abc_bytes.
 * (removeSoftLineBreaks handles: the row kernels follow the merge chain on its side stream, so the image is complete once that
 * stream's work is -- i.e. after any expansion call on the scratch buffer has been ordered, or a device synchronisation.) */
int zkwg_expand_abc_host(const zkwg_circuit_t* c, const uint8_t* packed_inputs, uint64_t n_emails, uint8_t* scratch_host,
                         uint64_t first, uint64_t count, int rows_on_host, uint8_t* out, uint64_t out_stride);
/* Layout-only handles of a fully numbered circuit (zkwg_circuit_create_full with device < 0): the complete witness of emails
 * [first, first + count) from a host copy of their images, evaluated through the same descriptor / row tables the device
 * kernels read (scratch_host is written: the row results).  A test hook like zkwg_o0_gather_host; device handles keep these
 * tables on the device only (BAD_CONFIG). */
int zkwg_expand_full_host(const zkwg_circuit_t* c, const uint8_t* packed_inputs, uint64_t n_emails, uint8_t* scratch_host,
                          uint64_t first, uint64_t count, uint8_t* out, uint64_t out_stride);

/* ---- the compact image as a device-side interchange format --------------------------
 * zkwg_prepare_device leaves, per email, a compact IMAGE in the scratch buffer (~0.45 MB instead of the 57 MB
 * witness of EmailVerifier(1024,1536)): `bits` (u64 words of LSB-first bit groups), `small` (u32 values) and
 * `fr` (32-byte standard-form field elements).  Together with the email's packed input record and the circuit's
 * static SEGMENT TABLE it determines every witness slot: zk_expand is nothing but the evaluation of that table.
 * A device consumer that can evaluate the table itself (an MSM that treats bit / byte scalars specially, a
 * constraint checker ...) therefore never needs the 32-byte-per-signal witness in HBM.
 *
 * Scratch layout for a batch of n emails (all offsets in bytes, 256-byte aligned; zkwg_image_layout):
 *   hstates  u32[n][hstate_words]   SHA-256 chaining states (internal to the prepare kernels)
 *   bits     u64[n][bits_words]
 *   small    u32[n][small_words]
 *   fr       u8 [n][fr_elems][32]
 * Segment semantics (slot = witness index, r = r0 + (slot - first_slot), rec = the email's input record):
 *   see enum zkwg_segment_type; the per-type comments state how slot r derives from image / record.
 *   Inverses of small integers d (|d| <= zkwg_inverse_table_half) are d^-1 mod r (0 for d = 0).
 * The table follows the handle's layout (kept-v1 or a `.sym` order); ZKWG_SEG_HOLE only occurs for
 * zkwg_circuit_create_full handles (slots derived afterwards from the `.r1cs`). */
enum zkwg_segment_type {
  ZKWG_SEG_SMALL = 0,    /* small[src + r]                                                            */
  ZKWG_SEG_FR = 1,       /* fr[src + r]                                                               */
  ZKWG_SEG_BITS = 2,     /* groups of a bits in b words: bit (r % a) of bits[src + (r / a) * b ...]    */
  ZKWG_SEG_SHA_SP = 3,   /* 162-slot periods over 5 words (32,32,32,32,34 bits): SigmaPlus instances  */
  ZKWG_SEG_SHA_T1 = 4,   /* 131-slot periods over 4 words (32,32,32,35): T1 instances                 */
  ZKWG_SEG_SHA_T2 = 5,   /* 161-slot periods over 5 words (32,32,32,32,33): T2 instances              */
  ZKWG_SEG_ISZ = 6,      /* IsZero pairs: d = (i32)small[src + r/2]; even r: d == 0, odd r: d^-1       */
  ZKWG_SEG_SEL = 7,      /* 256 x ItemAtIndex(a): idx = (i32)small[src], digest words small[b..b+8)    */
  ZKWG_SEG_IN8 = 8,      /* rec[src + r]                                                              */
  ZKWG_SEG_IN8BITS = 9,  /* bit (r & 7) of rec[src + r/8]                                             */
  ZKWG_SEG_LIMB = 10,    /* 16-byte LE limb rec[src + 16 r]                                           */
  ZKWG_SEG_LTBITS = 11,  /* bits of (i32)small[src] + 2^a - i, i = r / (a+1), bit r % (a+1)            */
  ZKWG_SEG_REGSEL = 12,  /* SelectRegexReveal comparators (see csrc/zkwg_sched.h)                      */
  ZKWG_SEG_VSHIFT = 13,  /* VarShiftLeft.tmp[j][i] = small[b + (i + (small[src] & (2^(j+1)-1))) % a]    */
  ZKWG_SEG_B64BITS = 14, /* 6 bits of the Base64 value of char small[src + r/6]                        */
  ZKWG_SEG_B64 = 15,     /* Base64Lookup internals of char small[src + r/68]                           */
  ZKWG_SEG_DFA = 16,     /* BodyHashRegex DFA circuit arrays (a = kind, b/c = parameters)              */
  ZKWG_SEG_IN8MASK = 17, /* rec[src + r] * rec[a + r]                                                  */
  ZKWG_SEG_RSLB = 18,    /* RemoveSoftLineBreaks byte-derived arrays over rec[src ..]                  */
  ZKWG_SEG_HOLE = 19,    /* not produced by the schedule (zkwg_circuit_create_full: derived later)     */
  ZKWG_SEG_NET = 20      /* loaded regex template: slot r = small[src + r] decoded: 31-bit signed integer, bit 31 = inverse of it */
};
typedef struct zkwg_segment {
  uint64_t slot;     /* first witness slot */
  uint32_t nslots;
  uint32_t type;     /* enum zkwg_segment_type */
  uint32_t src, a, b, c;
  uint32_t r0;       /* index of the first element inside the logical array */
  uint32_t pad;
} zkwg_segment;
typedef struct zkwg_image_layout_t {
  uint64_t hstate_words, bits_words, small_words, fr_elems;   /* per email */
  uint64_t off_hstates, off_bits, off_small, off_fr;          /* byte offsets in the scratch buffer of n emails */
  uint64_t total_bytes;                                       /* = zkwg_scratch_bytes(c, n) */
} zkwg_image_layout_t;
int zkwg_image_layout(const zkwg_circuit_t* c, uint64_t n_emails, zkwg_image_layout_t* out);
uint64_t zkwg_segment_table(const zkwg_circuit_t* c, zkwg_segment* out, uint64_t cap);   /* returns the segment count */
uint32_t zkwg_inverse_table_half(const zkwg_circuit_t* c);

/* ---- multi-GPU (BASELINE.json configs[3]) -------------------------------------------
 * Emails are independent: the batch is cut into n_dev contiguous shards (zkwg_shard_range), shard i runs on
 * devices[i] through its own handle and host thread, and its witnesses reach `out_wtns` over that GPU's own
 * PCIe link -- no data-path collective.  The only exchange is the 100-byte/email result table
 * {status i32 LE, pubkeyHash, shaHi, shaLo (32-byte LE each)} gathered on devices[0] over RCCL / xGMI
 * (ncclGroupStart + ncclSend / ncclRecv) and copied to `table` (n x 100 bytes, may be NULL; then RCCL is
 * not touched).  For the other mains the 96 bytes are w[1..3] of the witness.  librccl.so is opened with
 * dlopen, and only when n_dev > 1. */
typedef struct zkwg_multi zkwg_multi_t;
void zkwg_shard_range(uint64_t n_emails, int n_shards, int i, uint64_t* first, uint64_t* count);
int zkwg_multi_create(const zkwg_config* cfg, const int* devices, int n_dev, zkwg_multi_t** out);
void zkwg_multi_destroy(zkwg_multi_t* m);
int zkwg_multi_devices(const zkwg_multi_t* m);
zkwg_circuit_t* zkwg_multi_circuit(zkwg_multi_t* m, int i);   /* the handle of shard i (geometry, packing) */
int zkwg_calculate_batch_multi(zkwg_multi_t* m, const uint8_t* packed_inputs, uint64_t n_emails,
                               uint8_t* out_wtns, uint64_t out_stride, int32_t* status, uint8_t* table,
                               uint64_t max_tile);

#ifdef __cplusplus
}
#endif
#endif /* ZKWG_H */
