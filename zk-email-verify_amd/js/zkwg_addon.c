/*
 * zkwg_addon.c -- thin N-API addon over the C-ABI of libzkwg.so (include/zkwg.h).
 *
 * This is the binding a maintainer of the reference adds so that the Node/TypeScript host
 * keeps its `calculateWitness` / `snarkjs.wtns.calculate` call sites
 * (packages/circuits/tests/email-verifier.test.ts:43, packages/helpers/src/chunked-zkey.ts:80)
 * while the witness is computed on the GPU.  N-API >= 6 (BigInt words), node >= 12.22.
 * All heavy work runs on a libuv worker thread (napi_create_async_work).
 */
#include <node_api.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include "../../include/zkwg.h"

#define NAPI_OK(call) do { if ((call) != napi_ok) { napi_throw_error(env, NULL, "zkwg addon: N-API call failed: " #call); return NULL; } } while (0)

static void circuit_finalize(napi_env env, void* data, void* hint) { zkwg_circuit_destroy((zkwg_circuit_t*)data); }

static int get_u32(napi_env env, napi_value obj, const char* key, uint32_t* out, uint32_t dflt) {
  napi_value v; bool has = false;
  *out = dflt;
  if (napi_has_named_property(env, obj, key, &has) != napi_ok || !has) return 0;
  if (napi_get_named_property(env, obj, key, &v) != napi_ok) return -1;
  return napi_get_value_uint32(env, v, out) == napi_ok ? 0 : -1;
}

/* string property -> malloc'ed UTF-8 copy (NULL when absent / not a string) */
static void get_str(napi_env env, napi_value obj, const char* key, char** out, size_t* len) {
  napi_value v; bool has = false; napi_valuetype t;
  *out = NULL; *len = 0;
  if (napi_has_named_property(env, obj, key, &has) != napi_ok || !has) return;
  if (napi_get_named_property(env, obj, key, &v) != napi_ok || napi_typeof(env, v, &t) != napi_ok || t != napi_string) return;
  size_t n = 0;
  if (napi_get_value_string_utf8(env, v, NULL, 0, &n) != napi_ok) return;
  char* buf = (char*)malloc(n + 1);
  if (!buf) return;
  if (napi_get_value_string_utf8(env, v, buf, n + 1, &n) != napi_ok) { free(buf); return; }
  *out = buf; *len = n;
}

/* createCircuit({mainKind,maxHeader,maxBody,n,k,ignoreBodyHashCheck}, device) -> External */
static napi_value CreateCircuit(napi_env env, napi_callback_info info) {
  size_t argc = 2; napi_value argv[2];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  zkwg_config cfg; memset(&cfg, 0, sizeof(cfg));
  get_u32(env, argv[0], "mainKind", &cfg.main_kind, 0);
  get_u32(env, argv[0], "maxHeader", &cfg.max_header, 1024);
  get_u32(env, argv[0], "maxBody", &cfg.max_body, 1536);
  get_u32(env, argv[0], "n", &cfg.n, 121);
  get_u32(env, argv[0], "k", &cfg.k, 17);
  get_u32(env, argv[0], "ignoreBodyHashCheck", &cfg.ignore_body_hash_check, 0);
  get_u32(env, argv[0], "enableHeaderMasking", &cfg.enable_header_masking, 0);
  get_u32(env, argv[0], "enableBodyMasking", &cfg.enable_body_masking, 0);
  get_u32(env, argv[0], "removeSoftLineBreaks", &cfg.remove_soft_line_breaks, 0);
  int32_t device = 0;
  if (argc > 1) napi_get_value_int32(env, argv[1], &device);
  zkwg_circuit_t* c = NULL;
  /* optional `sym` (text of the compiled circuit's .sym file) and `symAlias` (rename rules) */
  char* sym = NULL; size_t sym_len = 0; char* alias = NULL; size_t alias_len = 0;
  get_str(env, argv[0], "sym", &sym, &sym_len);
  get_str(env, argv[0], "symAlias", &alias, &alias_len);
  /* optional `r1cs` (Buffer with the compiled circuit's .r1cs): complete witness of an O0 / O1 build */
  void* r1cs = NULL; size_t r1cs_len = 0;
  {
    bool has = false; napi_value v; bool isbuf = false;
    if (napi_has_named_property(env, argv[0], "r1cs", &has) == napi_ok && has &&
        napi_get_named_property(env, argv[0], "r1cs", &v) == napi_ok && napi_is_buffer(env, v, &isbuf) == napi_ok && isbuf)
      napi_get_buffer_info(env, v, &r1cs, &r1cs_len);
  }
  /* optional `regex` (path of a zk-regex style body_hash_regex.circom), `regexIncludeDirs` (':'-separated),
   * `regexTemplate`: BodyHashRegex is compiled from the template text (zkwg_circuit_create_regex) */
  char* regex = NULL; size_t regex_len = 0; char* rdirs = NULL; size_t rdirs_len = 0; char* rtmpl = NULL; size_t rtmpl_len = 0;
  get_str(env, argv[0], "regex", &regex, &regex_len);
  get_str(env, argv[0], "regexIncludeDirs", &rdirs, &rdirs_len);
  get_str(env, argv[0], "regexTemplate", &rtmpl, &rtmpl_len);
  int rc;
  if (regex) {
    zkwg_regex_source src = {regex, rdirs, rtmpl};
    rc = zkwg_circuit_create_regex(&cfg, device, &src, sym, sym_len, alias, alias_len, (const uint8_t*)r1cs, r1cs_len, &c);
  } else
    rc = sym ? (r1cs ? zkwg_circuit_create_full(&cfg, device, sym, sym_len, alias, alias_len, (const uint8_t*)r1cs, r1cs_len, &c)
                     : zkwg_circuit_create_sym(&cfg, device, sym, sym_len, alias, alias_len, &c))
             : zkwg_circuit_create(&cfg, device, &c);
  free(sym); free(alias); free(regex); free(rdirs); free(rtmpl);
  if (rc != ZKWG_RC_OK) {
    char msg[512];
    snprintf(msg, sizeof(msg), "%s%s%s", zkwg_strerror(rc), rc == ZKWG_RC_BAD_CONFIG ? ": " : "", rc == ZKWG_RC_BAD_CONFIG ? zkwg_last_error() : "");
    napi_throw_error(env, NULL, msg);
    return NULL;
  }
  napi_value ext;
  NAPI_OK(napi_create_external(env, c, circuit_finalize, NULL, &ext));
  return ext;
}

static zkwg_circuit_t* unwrap(napi_env env, napi_value v) {
  void* p = NULL;
  if (napi_get_value_external(env, v, &p) != napi_ok) { napi_throw_type_error(env, NULL, "zkwg: circuit handle expected"); return NULL; }
  return (zkwg_circuit_t*)p;
}

/* info(circuit) -> {witnessLen, witnessBytes, numPublic, inputStride, wtnsSize, offsets: [9]} */
static napi_value Info(napi_env env, napi_callback_info info) {
  size_t argc = 1; napi_value argv[1];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  zkwg_circuit_t* c = unwrap(env, argv[0]);
  if (!c) return NULL;
  napi_value obj, v, arr;
  NAPI_OK(napi_create_object(env, &obj));
  NAPI_OK(napi_create_double(env, (double)zkwg_witness_len(c), &v)); NAPI_OK(napi_set_named_property(env, obj, "witnessLen", v));
  NAPI_OK(napi_create_double(env, (double)zkwg_witness_bytes(c), &v)); NAPI_OK(napi_set_named_property(env, obj, "witnessBytes", v));
  NAPI_OK(napi_create_uint32(env, zkwg_num_public(c), &v)); NAPI_OK(napi_set_named_property(env, obj, "numPublic", v));
  NAPI_OK(napi_create_double(env, (double)zkwg_input_stride(c), &v)); NAPI_OK(napi_set_named_property(env, obj, "inputStride", v));
  NAPI_OK(napi_create_double(env, (double)zkwg_wtns_size(c), &v)); NAPI_OK(napi_set_named_property(env, obj, "wtnsSize", v));
  NAPI_OK(napi_create_array_with_length(env, ZKWG_IN_NFIELDS, &arr));
  for (int f = 0; f < ZKWG_IN_NFIELDS; ++f) {
    NAPI_OK(napi_create_double(env, (double)zkwg_input_offset(c, f), &v));
    NAPI_OK(napi_set_element(env, arr, f, v));
  }
  NAPI_OK(napi_set_named_property(env, obj, "offsets", arr));
  return obj;
}

/* ---- async batch calculation -------------------------------------------------------------- */
typedef struct {
  napi_async_work work;
  napi_deferred deferred;
  napi_ref in_ref;
  zkwg_circuit_t* c;
  zkwg_multi_t* multi;  /* non-NULL: shard over its devices (zkwg_calculate_batch_multi), c = shard 0's handle */
  uint8_t* table;       /* multi: n x 100 bytes gathered result table (malloc'ed) */
  const uint8_t* in;
  uint64_t n;
  uint8_t* out;      /* malloc'ed: n * witness_bytes (NULL if want_witness == 0) */
  int32_t* status;   /* malloc'ed */
  int want_witness;
  int as_wtns;       /* 1: wrap each witness into a .wtns container */
  int rc;
} calc_job;

static void calc_execute(napi_env env, void* data) {
  calc_job* j = (calc_job*)data;
  const uint64_t wb = zkwg_witness_bytes(j->c);
  j->status = (int32_t*)calloc(j->n, sizeof(int32_t));
  if (j->want_witness) j->out = (uint8_t*)malloc(j->n * wb);
  if (!j->status || (j->want_witness && !j->out)) { j->rc = ZKWG_RC_OOM; return; }
  if (j->multi) {
    j->table = (uint8_t*)malloc(j->n * 100);
    if (!j->table) { j->rc = ZKWG_RC_OOM; return; }
    j->rc = zkwg_calculate_batch_multi(j->multi, j->in, j->n, j->out, wb, j->status, j->table, 0);
  } else {
    j->rc = zkwg_calculate_batch(j->c, j->in, j->n, j->out, wb, j->status, 0);
  }
}

static void free_cb(napi_env env, void* data, void* hint) { free(data); }

static void calc_complete(napi_env env, napi_status st, void* data) {
  calc_job* j = (calc_job*)data;
  napi_value result = NULL, err = NULL, msg;
  if (j->rc != ZKWG_RC_OK) {
    napi_create_string_utf8(env, zkwg_strerror(j->rc), NAPI_AUTO_LENGTH, &msg);
    napi_create_error(env, NULL, msg, &err);
  } else {
    napi_value obj, statusArr, ab, wit;
    napi_create_object(env, &obj);
    void* sdata;
    napi_create_arraybuffer(env, j->n * sizeof(int32_t), &sdata, &ab);
    memcpy(sdata, j->status, j->n * sizeof(int32_t));
    napi_create_typedarray(env, napi_int32_array, j->n, ab, 0, &statusArr);
    napi_set_named_property(env, obj, "status", statusArr);
    if (j->want_witness) {
      const uint64_t wb = zkwg_witness_bytes(j->c);
      if (j->as_wtns && j->n == 1) {
        const uint64_t ws = zkwg_wtns_size(j->c);
        uint8_t* w = (uint8_t*)malloc(ws);
        zkwg_write_wtns(j->c, j->out, w, ws);
        napi_create_external_buffer(env, ws, w, free_cb, NULL, &wit);
        free(j->out);
      } else {
        napi_create_external_buffer(env, j->n * wb, j->out, free_cb, NULL, &wit);
      }
      j->out = NULL;
      napi_set_named_property(env, obj, "witness", wit);
    }
    if (j->table) {
      napi_value tb;
      napi_create_external_buffer(env, j->n * 100, j->table, free_cb, NULL, &tb);
      j->table = NULL;
      napi_set_named_property(env, obj, "table", tb);
    }
    result = obj;
  }
  if (err) napi_reject_deferred(env, j->deferred, err); else napi_resolve_deferred(env, j->deferred, result);
  napi_delete_reference(env, j->in_ref);
  napi_delete_async_work(env, j->work);
  free(j->status); free(j->out); free(j->table); free(j);
}

/* calculateBatch(circuit, recordsBuffer, wantWitness, asWtns) -> Promise<{status: Int32Array, witness?: Buffer}> */
static napi_value CalculateBatch(napi_env env, napi_callback_info info) {
  size_t argc = 4; napi_value argv[4];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  zkwg_circuit_t* c = unwrap(env, argv[0]);
  if (!c) return NULL;
  void* data; size_t len;
  NAPI_OK(napi_get_buffer_info(env, argv[1], &data, &len));
  const uint64_t stride = zkwg_input_stride(c);
  if (len == 0 || len % stride) { napi_throw_range_error(env, NULL, "zkwg: records buffer is not a whole number of input records"); return NULL; }
  calc_job* j = (calc_job*)calloc(1, sizeof(calc_job));
  j->c = c; j->in = (const uint8_t*)data; j->n = len / stride;
  bool b = true;
  if (argc > 2) napi_get_value_bool(env, argv[2], &b);
  j->want_witness = b ? 1 : 0;
  b = false;
  if (argc > 3) napi_get_value_bool(env, argv[3], &b);
  j->as_wtns = b ? 1 : 0;
  napi_value promise, name;
  NAPI_OK(napi_create_promise(env, &j->deferred, &promise));
  NAPI_OK(napi_create_reference(env, argv[1], 1, &j->in_ref));   /* keep the input buffer alive */
  NAPI_OK(napi_create_string_utf8(env, "zkwg.calculateBatch", NAPI_AUTO_LENGTH, &name));
  NAPI_OK(napi_create_async_work(env, NULL, name, calc_execute, calc_complete, j, &j->work));
  NAPI_OK(napi_queue_async_work(env, j->work));
  return promise;
}

/* ---- multi-GPU (zkwg_multi_*) ------------------------------------------------------------------ */
static void multi_finalize(napi_env env, void* data, void* hint) { zkwg_multi_destroy((zkwg_multi_t*)data); }

/* createMulti(opts, devices: number[]) -> external */
static napi_value CreateMulti(napi_env env, napi_callback_info info) {
  size_t argc = 2; napi_value argv[2];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  zkwg_config cfg; memset(&cfg, 0, sizeof(cfg));
  get_u32(env, argv[0], "mainKind", &cfg.main_kind, 0);
  get_u32(env, argv[0], "maxHeader", &cfg.max_header, 1024);
  get_u32(env, argv[0], "maxBody", &cfg.max_body, 1536);
  get_u32(env, argv[0], "n", &cfg.n, 121);
  get_u32(env, argv[0], "k", &cfg.k, 17);
  get_u32(env, argv[0], "ignoreBodyHashCheck", &cfg.ignore_body_hash_check, 0);
  get_u32(env, argv[0], "enableHeaderMasking", &cfg.enable_header_masking, 0);
  get_u32(env, argv[0], "enableBodyMasking", &cfg.enable_body_masking, 0);
  get_u32(env, argv[0], "removeSoftLineBreaks", &cfg.remove_soft_line_breaks, 0);
  uint32_t nd = 0;
  NAPI_OK(napi_get_array_length(env, argv[1], &nd));
  if (nd == 0 || nd > 64) { napi_throw_range_error(env, NULL, "zkwg: devices must list 1..64 GPUs"); return NULL; }
  int devs[64];
  for (uint32_t i = 0; i < nd; ++i) {
    napi_value v; int32_t d = 0;
    NAPI_OK(napi_get_element(env, argv[1], i, &v));
    NAPI_OK(napi_get_value_int32(env, v, &d));
    devs[i] = d;
  }
  zkwg_multi_t* m = NULL;
  int rc = zkwg_multi_create(&cfg, devs, (int)nd, &m);
  if (rc != ZKWG_RC_OK) {
    char msg[512];
    snprintf(msg, sizeof(msg), "%s%s%s", zkwg_strerror(rc), rc == ZKWG_RC_BAD_CONFIG ? ": " : "", rc == ZKWG_RC_BAD_CONFIG ? zkwg_last_error() : "");
    napi_throw_error(env, NULL, msg);
    return NULL;
  }
  napi_value ext;
  NAPI_OK(napi_create_external(env, m, multi_finalize, NULL, &ext));
  return ext;
}

/* calculateBatchMulti(multi, recordsBuffer, wantWitness) -> Promise<{status, witness?, table}> */
static napi_value CalculateBatchMulti(napi_env env, napi_callback_info info) {
  size_t argc = 3; napi_value argv[3];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  void* p = NULL;
  if (napi_get_value_external(env, argv[0], &p) != napi_ok || !p) { napi_throw_type_error(env, NULL, "zkwg: multi handle expected"); return NULL; }
  zkwg_multi_t* m = (zkwg_multi_t*)p;
  zkwg_circuit_t* c = zkwg_multi_circuit(m, 0);
  void* data; size_t len;
  NAPI_OK(napi_get_buffer_info(env, argv[1], &data, &len));
  const uint64_t stride = zkwg_input_stride(c);
  if (len == 0 || len % stride) { napi_throw_range_error(env, NULL, "zkwg: records buffer is not a whole number of input records"); return NULL; }
  calc_job* j = (calc_job*)calloc(1, sizeof(calc_job));
  j->c = c; j->multi = m; j->in = (const uint8_t*)data; j->n = len / stride;
  bool b = true;
  if (argc > 2) napi_get_value_bool(env, argv[2], &b);
  j->want_witness = b ? 1 : 0;
  napi_value promise, name;
  NAPI_OK(napi_create_promise(env, &j->deferred, &promise));
  NAPI_OK(napi_create_reference(env, argv[1], 1, &j->in_ref));
  NAPI_OK(napi_create_string_utf8(env, "zkwg.calculateBatchMulti", NAPI_AUTO_LENGTH, &name));
  NAPI_OK(napi_create_async_work(env, NULL, name, calc_execute, calc_complete, j, &j->work));
  NAPI_OK(napi_queue_async_work(env, j->work));
  return promise;
}
/* multiInfo(multi) -> same object as info() for shard 0's handle, plus nDevices */
static napi_value MultiCircuit0(napi_env env, napi_callback_info info) {
  size_t argc = 1; napi_value argv[1];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  void* p = NULL;
  if (napi_get_value_external(env, argv[0], &p) != napi_ok || !p) { napi_throw_type_error(env, NULL, "zkwg: multi handle expected"); return NULL; }
  napi_value v;
  NAPI_OK(napi_create_int32(env, zkwg_multi_devices((zkwg_multi_t*)p), &v));
  return v;
}

/* witnessToBigInts(buffer) -> bigint[]   (32-byte LE words -> napi_create_bigint_words) */
static napi_value WitnessToBigInts(napi_env env, napi_callback_info info) {
  size_t argc = 1; napi_value argv[1];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  void* data; size_t len;
  NAPI_OK(napi_get_buffer_info(env, argv[0], &data, &len));
  size_t n = len / 32;
  napi_value arr;
  NAPI_OK(napi_create_array_with_length(env, n, &arr));
  for (size_t i = 0; i < n; ++i) {
    uint64_t words[4];
    memcpy(words, (const uint8_t*)data + 32 * i, 32);
    napi_value v;
    NAPI_OK(napi_create_bigint_words(env, 0, 4, words, &v));
    NAPI_OK(napi_set_element(env, arr, i, v));
  }
  return arr;
}

/* packField(circuit, record: Buffer, field, first, values32: Buffer) -- generic 32-byte-per-signal input
 * path of include/zkwg.h (zkwg_pack_field): values that do not fit their packed slot raise the range flag */
static napi_value PackField(napi_env env, napi_callback_info info) {
  size_t argc = 5; napi_value argv[5];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  zkwg_circuit_t* c = unwrap(env, argv[0]);
  if (!c) return NULL;
  void* rec = NULL; size_t rec_len = 0; void* vals = NULL; size_t vals_len = 0;
  int32_t field = 0; double first = 0;
  NAPI_OK(napi_get_buffer_info(env, argv[1], &rec, &rec_len));
  NAPI_OK(napi_get_value_int32(env, argv[2], &field));
  NAPI_OK(napi_get_value_double(env, argv[3], &first));
  NAPI_OK(napi_get_buffer_info(env, argv[4], &vals, &vals_len));
  if (rec_len < zkwg_input_stride(c) || vals_len % 32 != 0) { napi_throw_range_error(env, NULL, "zkwg: packField buffer sizes"); return NULL; }
  int rc = zkwg_pack_field(c, (uint8_t*)rec, field, (uint64_t)first, (const uint8_t*)vals, vals_len / 32);
  if (rc != ZKWG_RC_OK) { napi_throw_error(env, NULL, zkwg_strerror(rc)); return NULL; }
  return NULL;
}

// setHostExpand(circuit, threads): threads > 0 -> calculateBatch expands the witnesses on the host from the downloaded
// image (zkwg_set_host_expand: 0.45 MB per email over PCIe instead of 56.9 MB); 0 -> on the device (default)
static napi_value SetHostExpand(napi_env env, napi_callback_info info) {
  size_t argc = 2; napi_value argv[2];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  zkwg_circuit_t* c = unwrap(env, argv[0]);
  if (!c) return NULL;
  int32_t threads = 0;
  NAPI_OK(napi_get_value_int32(env, argv[1], &threads));
  int rc = zkwg_set_host_expand(c, threads);
  if (rc != ZKWG_RC_OK) { napi_throw_error(env, NULL, zkwg_strerror(rc)); return NULL; }
  return NULL;
}

static napi_value StrError(napi_env env, napi_callback_info info) {
  size_t argc = 1; napi_value argv[1]; int32_t code = 0; napi_value s;
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  napi_get_value_int32(env, argv[0], &code);
  NAPI_OK(napi_create_string_utf8(env, zkwg_strerror(code), NAPI_AUTO_LENGTH, &s));
  return s;
}

/* ---- .r1cs / checkConstraints ------------------------------------------------------------- */
static void r1cs_finalize(napi_env env, void* data, void* hint) { (void)env; (void)hint; zkwg_r1cs_destroy((zkwg_r1cs_t*)data); }
/* r1csLoad(fileBytes: Buffer, device) -> {handle, nWires, nPubOut, nPubIn, nPrvIn, nConstraints, nLabels} */
static napi_value R1csLoad(napi_env env, napi_callback_info info) {
  size_t argc = 2; napi_value argv[2];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  void* data; size_t len;
  NAPI_OK(napi_get_buffer_info(env, argv[0], &data, &len));
  int32_t device = 0;
  if (argc > 1) napi_get_value_int32(env, argv[1], &device);
  zkwg_r1cs_t* r = NULL;
  int rc = zkwg_r1cs_load((const uint8_t*)data, len, device, &r);
  if (rc != ZKWG_RC_OK) { napi_throw_error(env, NULL, rc == ZKWG_RC_BAD_CONFIG ? "zkwg: malformed or unsupported .r1cs file" : zkwg_strerror(rc)); return NULL; }
  uint64_t inf[6];
  zkwg_r1cs_info(r, inf);
  napi_value obj, ext, v;
  NAPI_OK(napi_create_object(env, &obj));
  NAPI_OK(napi_create_external(env, r, r1cs_finalize, NULL, &ext));
  NAPI_OK(napi_set_named_property(env, obj, "handle", ext));
  const char* keys[6] = {"nWires", "nPubOut", "nPubIn", "nPrvIn", "nConstraints", "nLabels"};
  for (int i = 0; i < 6; ++i) { NAPI_OK(napi_create_double(env, (double)inf[i], &v)); NAPI_OK(napi_set_named_property(env, obj, keys[i], v)); }
  return obj;
}
/* r1csCheck(handle, witnesses: Buffer, n, strideBytes) -> number[] (first violated constraint, -1 = all hold) */
static napi_value R1csCheck(napi_env env, napi_callback_info info) {
  size_t argc = 4; napi_value argv[4];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  void* p = NULL;
  if (napi_get_value_external(env, argv[0], &p) != napi_ok || !p) { napi_throw_type_error(env, NULL, "zkwg: r1cs handle expected"); return NULL; }
  void* data; size_t len;
  NAPI_OK(napi_get_buffer_info(env, argv[1], &data, &len));
  int64_t n = 0, stride = 0;
  napi_get_value_int64(env, argv[2], &n); napi_get_value_int64(env, argv[3], &stride);
  if (n <= 0 || stride <= 0 || (uint64_t)n * (uint64_t)stride > len) { napi_throw_range_error(env, NULL, "zkwg: witness buffer too small"); return NULL; }
  uint64_t* bad = (uint64_t*)malloc((size_t)n * 8);
  int rc = zkwg_check_constraints((zkwg_r1cs_t*)p, (const uint8_t*)data, (uint64_t)n, (uint64_t)stride, bad);
  if (rc != ZKWG_RC_OK) { free(bad); napi_throw_error(env, NULL, zkwg_strerror(rc)); return NULL; }
  napi_value arr, v;
  NAPI_OK(napi_create_array_with_length(env, (size_t)n, &arr));
  for (int64_t i = 0; i < n; ++i) {
    NAPI_OK(napi_create_double(env, bad[i] == UINT64_MAX ? -1.0 : (double)bad[i], &v));
    NAPI_OK(napi_set_element(env, arr, (uint32_t)i, v));
  }
  free(bad);
  return arr;
}

/* ---- groth16.prove (zkwg_prover_*, the second half of fullProve: packages/helpers/src/chunked-zkey.ts:80-84) ----------------- */
static void prover_finalize(napi_env env, void* data, void* hint) { (void)env; (void)hint; zkwg_prover_destroy((zkwg_prover_t*)data); }
/* createProver(circuit, device, r1cs: Buffer, nRows, key: {nWires, nPublic, log2Domain, a, b1, b2, c, h, alpha1, beta1, beta2, delta1, delta2 (Buffers)}, slots) -> external */
static napi_value CreateProver(napi_env env, napi_callback_info info) {
  size_t argc = 6; napi_value argv[6];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  zkwg_circuit_t* c = unwrap(env, argv[0]);
  if (!c) return NULL;
  int32_t device = 0; uint32_t slots = 8; int64_t n_rows = 0;
  napi_get_value_int32(env, argv[1], &device);
  void* r1cs; size_t r1cs_len;
  NAPI_OK(napi_get_buffer_info(env, argv[2], &r1cs, &r1cs_len));
  napi_get_value_int64(env, argv[3], &n_rows);
  if (argc > 5) napi_get_value_uint32(env, argv[5], &slots);
  zkwg_proving_key key;
  memset(&key, 0, sizeof key);
  uint32_t u;
  if (get_u32(env, argv[4], "nWires", &u, 0)) return NULL; key.n_wires = u;
  if (get_u32(env, argv[4], "nPublic", &u, 0)) return NULL; key.n_public = u;
  if (get_u32(env, argv[4], "log2Domain", &u, 0)) return NULL; key.log2_domain = u;
  const char* names[10] = {"a", "b1", "b2", "c", "h", "alpha1", "beta1", "beta2", "delta1", "delta2"};
  const size_t unit[10] = {64, 64, 128, 64, 64, 64, 64, 128, 64, 128};
  const size_t count[10] = {key.n_wires, key.n_wires, key.n_wires, key.n_wires - key.n_public - 1, (size_t)1 << key.log2_domain, 1, 1, 1, 1, 1};
  const void* ptr[10];
  for (int i = 0; i < 10; ++i) {
    napi_value v; void* d; size_t len;
    if (napi_get_named_property(env, argv[4], names[i], &v) != napi_ok || napi_get_buffer_info(env, v, &d, &len) != napi_ok || len != unit[i] * count[i]) {
      napi_throw_type_error(env, NULL, "zkwg: the proving key needs Buffers a, b1, b2, c, h (sections 5-9 of the .zkey) and alpha1, beta1, beta2, delta1, delta2 of the right sizes");
      return NULL;
    }
    ptr[i] = d;
  }
  key.a = ptr[0]; key.b1 = ptr[1]; key.b2 = ptr[2]; key.c = ptr[3]; key.h = ptr[4];
  memcpy(key.alpha1, ptr[5], 64); memcpy(key.beta1, ptr[6], 64); memcpy(key.beta2, ptr[7], 128); memcpy(key.delta1, ptr[8], 64); memcpy(key.delta2, ptr[9], 128);
  zkwg_prover_t* pv = NULL;
  int rc = zkwg_prover_create(c, device, (const uint8_t*)r1cs, r1cs_len, (uint64_t)n_rows, &key, slots, &pv);
  if (rc != ZKWG_RC_OK) { napi_throw_error(env, NULL, zkwg_strerror(rc)); return NULL; }
  napi_value ext;
  NAPI_OK(napi_create_external(env, pv, prover_finalize, NULL, &ext));
  return ext;
}
/* createProverZkey(circuit, device, zkey: Buffer, slots) -> external: the prover from the zkey alone (zkwg_prover_create_zkey) --
 * what groth16.prove(zkey, wtns) / fullProve(input, wasm, zkey) take (packages/helpers/src/chunked-zkey.ts:80-84) */
static napi_value CreateProverZkey(napi_env env, napi_callback_info info) {
  size_t argc = 4; napi_value argv[4];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  zkwg_circuit_t* c = unwrap(env, argv[0]);
  if (!c) return NULL;
  int32_t device = 0; uint32_t slots = 8;
  napi_get_value_int32(env, argv[1], &device);
  void* zkey; size_t zkey_len;
  NAPI_OK(napi_get_buffer_info(env, argv[2], &zkey, &zkey_len));
  if (argc > 3) napi_get_value_uint32(env, argv[3], &slots);
  zkwg_prover_t* pv = NULL;
  int rc = zkwg_prover_create_zkey(c, device, (const uint8_t*)zkey, zkey_len, slots, &pv);
  if (rc != ZKWG_RC_OK) { napi_throw_error(env, NULL, zkwg_strerror(rc)); return NULL; }
  napi_value ext;
  NAPI_OK(napi_create_external(env, pv, prover_finalize, NULL, &ext));
  return ext;
}
typedef struct {
  zkwg_prover_t* p; const uint8_t* in; const uint8_t* blinding; uint64_t n;
  int32_t* status; uint8_t* proofs; int rc;
  napi_deferred deferred; napi_async_work work; napi_ref in_ref, bl_ref;
} prove_job;
static void prove_execute(napi_env env, void* data) {
  prove_job* j = (prove_job*)data; (void)env;
  j->rc = zkwg_prover_prove_batch(j->p, j->in, j->n, j->blinding, j->status, j->proofs);
}
static void prove_complete(napi_env env, napi_status st, void* data) {
  prove_job* j = (prove_job*)data; (void)st;
  if (j->rc != ZKWG_RC_OK) {
    napi_value msg, err;
    napi_create_string_utf8(env, zkwg_strerror(j->rc), NAPI_AUTO_LENGTH, &msg);
    napi_create_error(env, NULL, msg, &err);
    napi_reject_deferred(env, j->deferred, err);
  } else {
    napi_value obj, ab, ta, buf; void* p;
    napi_create_object(env, &obj);
    napi_create_arraybuffer(env, j->n * 4, &p, &ab); memcpy(p, j->status, j->n * 4);
    napi_create_typedarray(env, napi_int32_array, j->n, ab, 0, &ta);
    napi_set_named_property(env, obj, "status", ta);
    napi_create_buffer_copy(env, j->n * 256, j->proofs, NULL, &buf);
    napi_set_named_property(env, obj, "proofs", buf);
    napi_resolve_deferred(env, j->deferred, obj);
  }
  napi_delete_reference(env, j->in_ref); napi_delete_reference(env, j->bl_ref);
  napi_delete_async_work(env, j->work);
  free(j->status); free(j->proofs); free(j);
}
/* proveBatch(prover, circuit, recordsBuffer, blindingBuffer (64 bytes per email: r | s)) -> Promise<{status: Int32Array, proofs: Buffer (256 bytes per email)}> */
static napi_value ProveBatch(napi_env env, napi_callback_info info) {
  size_t argc = 4; napi_value argv[4];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  void* pv = NULL;
  if (napi_get_value_external(env, argv[0], &pv) != napi_ok || !pv) { napi_throw_type_error(env, NULL, "zkwg: prover handle expected"); return NULL; }
  zkwg_circuit_t* c = unwrap(env, argv[1]);
  if (!c) return NULL;
  void *data, *bl; size_t len, bl_len;
  NAPI_OK(napi_get_buffer_info(env, argv[2], &data, &len));
  NAPI_OK(napi_get_buffer_info(env, argv[3], &bl, &bl_len));
  const uint64_t stride = zkwg_input_stride(c);
  if (len == 0 || len % stride || bl_len != (len / stride) * 64) { napi_throw_range_error(env, NULL, "zkwg: records / blinding buffers do not match"); return NULL; }
  prove_job* j = (prove_job*)calloc(1, sizeof(prove_job));
  j->p = (zkwg_prover_t*)pv; j->in = (const uint8_t*)data; j->blinding = (const uint8_t*)bl; j->n = len / stride;
  j->status = (int32_t*)calloc(j->n, 4); j->proofs = (uint8_t*)calloc(j->n, 256);
  napi_value promise, name;
  NAPI_OK(napi_create_promise(env, &j->deferred, &promise));
  NAPI_OK(napi_create_reference(env, argv[2], 1, &j->in_ref));
  NAPI_OK(napi_create_reference(env, argv[3], 1, &j->bl_ref));
  NAPI_OK(napi_create_string_utf8(env, "zkwg.proveBatch", NAPI_AUTO_LENGTH, &name));
  NAPI_OK(napi_create_async_work(env, NULL, name, prove_execute, prove_complete, j, &j->work));
  NAPI_OK(napi_queue_async_work(env, j->work));
  return promise;
}

/* symText(circuit) -> the layout's symbol table in `.sym` line format (zkwg_write_sym) */
static napi_value SymText(napi_env env, napi_callback_info info) {
  size_t argc = 1; napi_value argv[1];
  NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
  zkwg_circuit_t* c = unwrap(env, argv[0]);
  if (!c) return NULL;
  uint64_t need = zkwg_write_sym(c, NULL, 0);
  char* buf = (char*)malloc(need + 1);
  if (!buf) { napi_throw_error(env, NULL, "zkwg: out of memory"); return NULL; }
  zkwg_write_sym(c, buf, need);
  napi_value out;
  napi_status st = napi_create_string_utf8(env, buf, need, &out);
  free(buf);
  NAPI_OK(st);
  return out;
}

static napi_value Init(napi_env env, napi_value exports) {
  napi_property_descriptor d[] = {
      {"createCircuit", NULL, CreateCircuit, NULL, NULL, NULL, napi_default, NULL},
      {"info", NULL, Info, NULL, NULL, NULL, napi_default, NULL},
      {"calculateBatch", NULL, CalculateBatch, NULL, NULL, NULL, napi_default, NULL},
      {"witnessToBigInts", NULL, WitnessToBigInts, NULL, NULL, NULL, napi_default, NULL},
      {"strerror", NULL, StrError, NULL, NULL, NULL, napi_default, NULL},
      {"packField", NULL, PackField, NULL, NULL, NULL, napi_default, NULL},
      {"setHostExpand", NULL, SetHostExpand, NULL, NULL, NULL, napi_default, NULL},
      {"createMulti", NULL, CreateMulti, NULL, NULL, NULL, napi_default, NULL},
      {"calculateBatchMulti", NULL, CalculateBatchMulti, NULL, NULL, NULL, napi_default, NULL},
      {"multiDevices", NULL, MultiCircuit0, NULL, NULL, NULL, napi_default, NULL},
      {"symText", NULL, SymText, NULL, NULL, NULL, napi_default, NULL},
      {"r1csLoad", NULL, R1csLoad, NULL, NULL, NULL, napi_default, NULL},
      {"r1csCheck", NULL, R1csCheck, NULL, NULL, NULL, napi_default, NULL},
      {"createProver", NULL, CreateProver, NULL, NULL, NULL, napi_default, NULL},
      {"createProverZkey", NULL, CreateProverZkey, NULL, NULL, NULL, napi_default, NULL},
      {"proveBatch", NULL, ProveBatch, NULL, NULL, NULL, napi_default, NULL},
  };
  napi_define_properties(env, exports, sizeof(d) / sizeof(d[0]), d);
  return exports;
}
NAPI_MODULE(zkwg_addon, Init)
