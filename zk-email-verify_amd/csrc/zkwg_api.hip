// C-ABI of libzkwg.so (see include/zkwg.h): schedule construction, launch sequence,
// host<->device staging, .wtns / .sym writers.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include <mutex>
#include <map>
#include <thread>
#include <dlfcn.h>
#include <algorithm>
#include <chrono>
#include "../../include/zkwg.h"
#include "zkwg_kernels.h"
#include "zkwg_layout.h"
#include "zkwg_build.h"
#include "zkwg_poseidon_sparse.h"
#include "zkwg_poseidon29.h"
#include "zkwg_net_host.h"
#include "zkwg_full.h"
#include "zkwg_o0.h"
#include "zkwg_o0_dec.h"
#include "zkwg_expand_dec.h"
#include <atomic>
#include <emmintrin.h>

#define ZK_MAX_KERNELS 9
#define ZK_O0_SHORT_ROW 8   // host evaluation of the plan (tests): rows up to this many terms are listed first
#define ZK_RS_SLOTS 16
#define ZK_POS_STREAMS 4
#define ZK_POS_RING 64
#define ZK_EV_RING 512   // launches whose HIP events are kept for zkwg_timing_summary

struct zkwg_circuit;
extern "C" int zk_rows_copy_launch(const u8* src, u64 stride, u8* dst, u32 count, hipStream_t st);   // zkwg_kernels_handoff.hip
static void rp_free_ring(zkwg_circuit* c);   // the resident pipeline's output ring: chunk-mapped (zkwg_vmm.hip) or plain
struct zkwg_circuit {
  zkwg_config cfg;
  ZkSched s;
  int device;
  Fr* d_invtab;
  std::vector<std::string> sym_names;  // layout SYM: witness index -> name
  // linear completion plan of a fully numbered (O0 / O1) circuit (zkwg_full.h); null for compact layouts
  u64 lin_rows;
  std::vector<u32> kept_dst;   // `.sym` layouts: kept-v1 slot -> witness index (0xffffffff = dropped by the file)
  ZkLinPlan lin_host;   // kept for layout-only handles (tests evaluate it on the host)
  // fully numbered circuits (`--O0` / `--O1`): zk_expand_o0 writes every wire of the file straight from the image (zkwg_o0.h)
  u64 full_W;                      // wires of the `.sym` / `.r1cs` (0: not a fully numbered circuit)
  std::vector<u32> o0_short; u64 n_o0_short;
  std::vector<u32> o0_desc, o0_src, o0_long;   // host evaluation (layout-only handles): wire -> kept-v1 slot | 0xfffffffe (row);  row terms as kept-v1 slots;  the rows
  ZkO0Tables o0t;                  // per-wire descriptors + small / field rows (host copy; moved to the device for device handles)
  ZkO0Dev o0d{};                   // device pointers of the same
  u64 abc_m = 0;                   // constraints of the attached system (zkwg_circuit_attach_r1cs; 0: none)
  ZkO0Tables abct;                 // its 3 m linear combinations as descriptors + rows over the image, like o0t
  ZkO0Dev abcd{};
  Fr* d_invtab_m; // fused Montgomery output: inverse table in Montgomery form (built with d_rtab)
  // BodyHashRegex loaded from a circom template (zkwg_circuit_create_regex): gate list on the device
  zkc::Net net;
  bool has_net;
  u32* d_net_records; u32* d_net_counts; u32* d_net_mask_tab; u32* d_net_pd; u32* d_net_tabs;
  ZkNetDec* d_netd;   // how zk_expand decodes the region (device copy); h_netd: the same with host pointers (host expansion)
  ZkNetDec h_netd;
  u8* d_net_cclass; u8* d_net_cdelta; u32* d_net_cmask; u8* d_net_bclass; u8* d_net_bdelta; u32* d_net_bmask;
  Fr* d_rtab;     // fused Montgomery output: v * R mod r for v < 65536 (built on first use)
  Fr* d_pos;      // Poseidon(9): sparse-round table (zk_build_poseidon_sparse(10, 60))
  u32 pos2_off;
  u32 pos_dense_off;
  Fr* d_pos_rs;   // removeSoftLineBreaks: Poseidon(16) then Poseidon(2) sparse-round tables
  u32* d_pos_l29; // removeSoftLineBreaks: Poseidon(16) table in 29-bit limb form (zkwg_poseidon29.h)
  ZkSeg* d_segs;
  ZkPortionEntry* d_ent;   // zk_expand: one entry per piece of 256 K slots (zkwg_build.h zk_build_entries)
  u32 n_ent;
  std::vector<Fr> invtab_host;   // zkwg_expand_host: the inverse table on the host
  int host_expand_threads;       // > 0: zkwg_calculate_batch expands on the host (zkwg_set_host_expand)
  u8* hx_img[2]; u64 hx_bytes;   // pinned staging of downloaded images (host expansion)
  int o0_emails_per_wg;          // emails per workgroup of zk_expand3_o0 (ZKWG_O0_EMAILS_PER_WG, default 16)
  int o0_pipe;                   // zk_expand3_o0 variant (ZKWG_O0_PIPE): 0 plain, 1 one email ahead, 2 (default) batches of 4 emails + short paths for uniform pieces, 3 double-buffered batches
  int rslb_v;              // zk_rslb_chunks' evaluator variant (ZKWG_RSLB_V = 0..3, zkwg_poseidon29.h)
  u32 pos2_l29_off = 0;    // words: the Poseidon(2) limb table behind the Poseidon(16) one in d_pos_l29
  int rs_merge_lanes = 1;  // zk_rslb_merge1 (one lane per email, limb form) | 4: zk_rslb_merge (ZKWG_RSLB_MERGE_LANES)
  Fr* d_rs_zero = nullptr; // removeSoftLineBreaks: signals + digest of the all-zero chunk (constant chunks, zkwg_kernels_rslb.hip); null: off
  int x3_k, x3_k_o0;       // slots per thread of zk_expand3 (kept-v1 / sym layouts) and zk_expand3_o0: 1, 2 or 4 (ZKWG_X3_K, ZKWG_X3_K_O0)
  std::vector<ZkSeg> segs;
  hipStream_t own_stream, copy_stream;
  // removeSoftLineBreaks: the serial merge chain (zk_rslb_chain, ~16 waves per 1024 emails, latency-bound)
  // runs on a side stream so that the caller's stream can go on with the next batch; zkwg_expand_device
  // waits for the chain of the scratch buffer it reads.
  hipStream_t side_stream[ZK_RS_SLOTS];   // one per scratch buffer in flight: chains of different batches overlap
  hipEvent_t rs_dep[ZK_RS_SLOTS], rs_done[ZK_RS_SLOTS];
  const void* rs_scr[ZK_RS_SLOTS];
  int rs_next, rs_sync, rs_nside;
  // zk_poseidon9 (one lane per email: ~3.5 ms of latency, a handful of wavefronts) can run beside the
  // other prepare kernels on a low-priority side stream, joined at the end of prepare (off by default).
  hipStream_t pos_stream[ZK_POS_STREAMS];
  hipEvent_t pos_dep[ZK_POS_RING], pos_done[ZK_POS_RING];
  u64 pos_calls;
  int pos_side;   // 1: fork zk_poseidon9 onto a side stream (ZKWG_POS_SIDE=1); default 0 = caller's stream
  int pos_lane;   // 1: the lane-per-email zk_poseidon9 of round 2 instead of zk_poseidon9_g16 (ZKWG_POS_LANE=1)
  int pos_wave_below;   // batches below this many emails use the wavefront-per-email kernel (ZKWG_POS_WAVE_BELOW, default 1024)
  u32 xcd_remap;  // zk_expand workgroup -> portion mapping (ZKWG_XCD_REMAP)
  // host-buffer path: cached device staging buffers (double-buffered witnesses)
  std::mutex hb_mutex;
  // launch bookkeeping (event rings, merge-chain slots, counters) of the device entry points
  std::mutex dev_mutex;
  u8 *hb_in, *hb_out[2], *hb_scr;
  int* hb_status[2];
  u64 hb_tile;
  hipEvent_t hb_done[2], hb_copied[2];
  int timing;
  int n_kernels;
  const char* kname[ZK_MAX_KERNELS];
  u64 kslots[ZK_MAX_KERNELS];
  hipEvent_t ev[ZK_EV_RING][2];                    // zk_expand launches: start, stop
  hipEvent_t pev[ZK_EV_RING][ZK_MAX_KERNELS + 1];  // prepare launches: boundaries between kernels
  u64 launches;   // expand launches recorded since timing was enabled
  u64 prep_launches;
  bool ev_valid, prep_valid;
  int rsa_wgs_per_cu;
  u32 prep_mask = 0xffffffffu;   // zkwg_set_prepare_mask (measurement aid)
  // device-resident pipeline (zkwg_calculate_batch_resident): buffers cached in the handle, under hb_mutex
  u8* rp_in = nullptr; u64 rp_in_cap = 0;
  u8* rp_scr[2] = {nullptr, nullptr}; u64 rp_scr_bytes = 0;
  u8* rp_out[2] = {nullptr, nullptr}; u64 rp_tile_bytes = 0;
  int* rp_status = nullptr; u64 rp_n_cap = 0;
  hipStream_t rp_exp = nullptr;
  hipEvent_t rp_prep_done[2] = {nullptr, nullptr}, rp_exp_done[2] = {nullptr, nullptr};
  float rp_place_ms[8] = {0}; int rp_place_n = 0, rp_place_kept[2] = {-1, -1};   // (round 4's candidate timings: unused since the ring is chunked)
  bool rp_chunked = false;     // the ring's tiles come from zkwg_device_alloc_chunked
};

// inverse table: entry (d + half) holds d^{-1} mod r in standard form, d in [-half, half]
// Device entry points may be called from a thread whose current HIP device is not the handle's.
struct ZkDeviceGuard {
  int prev = -1;
  bool ok = true;
  explicit ZkDeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != dev) ok = hipSetDevice(dev) == hipSuccess; else prev = -1;
  }
  ~ZkDeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

static void build_inv_table(u32 inv_half, std::vector<Fr>& tab) {
  const long long half = (long long)inv_half;
  tab.assign(2 * (u64)inv_half + 1, fr_zero());
  std::vector<Fr> inv(half + 1, fr_zero());  // Montgomery form
  if (half >= 1) inv[1] = fr_R();
  // inv[i] = -(r / i) * inv[r mod i]  (mod r); r / i and r % i by long division on 4 limbs
  for (long long i = 2; i <= half; ++i) {
    const u64 p[4] = {ZK_P0, ZK_P1, ZK_P2, ZK_P3};
    u64 q[4];
    unsigned __int128 rem = 0;
    for (int j = 3; j >= 0; --j) {
      unsigned __int128 cur = (rem << 64) | p[j];
      q[j] = (u64)(cur / (u64)i);
      rem = cur % (u64)i;
    }
    Fr qf{{q[0], q[1], q[2], q[3]}};
    Fr t = fr_mont_mul(fr_to_mont(qf), inv[(u64)rem]);
    inv[i] = fr_neg(t);
  }
  for (long long d = 1; d <= half; ++d) {
    Fr v = fr_from_mont(inv[d]);
    tab[(u64)(half + d)] = v;
    tab[(u64)(half - d)] = fr_neg(v);
  }
}

static u64 align256(u64 x) { return (x + 255) & ~255ull; }
// host expansion helpers (zkwg_expand_host below)
static inline void zk_host_put(u8* dst, u32 code, const ZkRefSrc& R) {
  uint4 lo, hi;
  if (!(code >> 31)) { lo = make_uint4(code, 0, 0, 0); hi = make_uint4(0, 0, 0, 0); }
  else { lo = zk_ref_half(code, 0u, R); hi = zk_ref_half(code, 1u, R); }
  _mm_stream_si128((__m128i*)dst, _mm_set_epi32((int)lo.w, (int)lo.z, (int)lo.y, (int)lo.x));
  _mm_stream_si128((__m128i*)(dst + 16), _mm_set_epi32((int)hi.w, (int)hi.z, (int)hi.y, (int)hi.x));
}
template <class DEC>
static void zk_host_segment(const ZkSeg& sg, const ZkCtx& cx, const ZkRefSrc& R, u32 r0, u32 n, u8* dst) {
  const DEC dec(sg, cx);
  for (u32 i = 0; i < n; ++i) zk_host_put(dst + 32ull * i, dec(r0 + i), R);
}

extern "C" int zk_misc_init_tables(void);

// descriptor + row tables of a numbered layout (zkwg_o0.h) onto the device; the host copy keeps its counters only
static bool upload_o0(zkwg_circuit* c, ZkO0Tables& T, ZkO0Dev& O, u64 W, bool keep_host = false) {
  bool ok = true;
  memset(&O, 0, sizeof(O));
  auto up = [&](const void* src, size_t bytes, void** dst) {
    *dst = nullptr;
    if (!ok) return;
    ok = hipMalloc(dst, std::max<size_t>(bytes, 16)) == hipSuccess && (bytes == 0 || hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice) == hipSuccess);
  };
  up(T.desc.data(), T.desc.size() * 4, (void**)&O.desc);
  up(T.aff.data(), T.aff.size() * 4, (void**)&O.aff);
  up(T.s_ptr.data(), T.s_ptr.size() * 8, (void**)&O.s_ptr);
  up(T.s_term.data(), T.s_term.size() * 4, (void**)&O.s_term);
  up(T.s_coef.data(), T.s_coef.size() * 4, (void**)&O.s_coef);
  up(T.s_out.data(), T.s_out.size() * 4, (void**)&O.s_out);
  {
    // small rows: the groups of one row (a thread each, a wavefront for the long ones) and the chains (a wavefront each)
    std::vector<u32> single, lng, chains;
    for (size_t g = 0; g + 1 < T.s_group.size(); ++g) {
      const u32 a = T.s_group[g], b = T.s_group[g + 1];
      if (b - a == 1) (T.s_ptr[a + 1] - T.s_ptr[a] > ZK_ROW_LONG ? lng : single).push_back(a); else { chains.push_back(a); chains.push_back(b - a); }
    }
    up(single.data(), single.size() * 4, (void**)&O.s_single);
    up(lng.data(), lng.size() * 4, (void**)&O.s_long);
    up(chains.data(), chains.size() * 4, (void**)&O.s_chains);
    O.n_small_single = (u32)single.size(); O.n_small_long = (u32)lng.size(); O.n_small_chains = (u32)(chains.size() / 2);
  }
  up(T.f_ptr.data(), T.f_ptr.size() * 8, (void**)&O.f_ptr);
  up(T.f_term.data(), T.f_term.size() * 4, (void**)&O.f_term);
  up(T.f_coef.data(), T.f_coef.size() * sizeof(Fr), (void**)&O.f_coef);
  up(T.f_coefm.data(), T.f_coefm.size() * sizeof(Fr), (void**)&O.f_coefm);
  up(T.f_kind.data(), T.f_kind.size(), (void**)&O.f_kind);
  up(T.gen_seg.data(), T.gen_seg.size() * 4, (void**)&O.gen_seg);
  up(T.gen_r.data(), T.gen_r.size() * 4, (void**)&O.gen_r);
  O.n_gen = (u32)T.gen_seg.size(); O.gen_base = T.gen_base;
  O.W = W; O.nportions = (u32)((W + 256u * c->x3_k_o0 - 1) / (256u * c->x3_k_o0)); O.small_base = T.small_base; O.fr_base = T.fr_base;
  O.emails_per_wg = (u32)c->o0_emails_per_wg;
  O.n_fr_groups = (u32)T.n_fr();   // (field rows are never chained: one group each)
  if (ok && !keep_host) { ZkO0Tables keep; keep.small_base = T.small_base; keep.fr_base = T.fr_base; keep.gen_base = T.gen_base; keep.n_alias = T.n_alias; keep.n_const = T.n_const; std::swap(T, keep); }
  return ok;
}
static void free_o0(ZkO0Dev& O) {
  hipFree((void*)O.desc); hipFree((void*)O.aff); hipFree((void*)O.s_ptr); hipFree((void*)O.s_term); hipFree((void*)O.s_coef); hipFree((void*)O.s_out); hipFree((void*)O.s_single); hipFree((void*)O.s_long); hipFree((void*)O.s_chains);
  hipFree((void*)O.f_ptr); hipFree((void*)O.f_term); hipFree((void*)O.f_coef); hipFree((void*)O.f_coefm); hipFree((void*)O.f_kind); hipFree((void*)O.gen_seg); hipFree((void*)O.gen_r);
  memset(&O, 0, sizeof(O));
}
extern "C" {

int zkwg_abi_version(void) { return ZKWG_ABI_VERSION; }

const char* zkwg_strerror(int rc) {
  switch (rc) {
    case ZKWG_RC_OK: return "ok";
    case ZKWG_ERR_ASSERT_FAILED: return "Error: Assert Failed.";
    case 1: return "Signal not found";
    case 2: return "Too many signals set";
    case 3: return "Signal already set";
    case 5: return "Not enough memory";
    case 6: return "Input signal array access exceeds the size";
    case ZKWG_RC_BAD_CONFIG: return "zkwg: unsupported circuit configuration";
    case ZKWG_RC_BAD_ARG: return "zkwg: bad argument";
    case ZKWG_RC_NO_DEVICE: return "zkwg: no HIP device (layout-only handle or HIP unavailable)";
    case ZKWG_RC_HIP_ERROR: return "zkwg: HIP runtime error";
    case ZKWG_RC_OOM: return "zkwg: out of device memory";
    default: return "zkwg: unknown code";
  }
}

static thread_local std::string g_last_error;
const char* zkwg_last_error(void) { return g_last_error.c_str(); }

static int create_impl(const zkwg_config* cfg_in, int device, const char* sym_text, uint64_t sym_len,
                       const char* alias_text, uint64_t alias_len, zkwg_circuit_t** out,
                       const uint8_t* r1cs = nullptr, uint64_t r1cs_len = 0, const zkwg_regex_source* regex = nullptr) {
  if (!cfg_in || !out) return ZKWG_RC_BAD_ARG;
  zkwg_config cfg_copy = *cfg_in;
  cfg_copy.layout = ZKWG_LAYOUT_KEPT_V1;
  const zkwg_config* cfg = &cfg_copy;
  zkwg_circuit* c = new zkwg_circuit();
  c->cfg = *cfg;
  c->device = -1;
  // tuning knobs (DESIGN.md "zk_expand geometry"): slots per workgroup and threads per workgroup
  auto pick_k = [](const char* name, int dflt, bool k8 = false) { const char* v = getenv(name); const int k = v ? atoi(v) : dflt; return (k == 1 || k == 2 || k == 4 || (k8 && k == 8)) ? k : dflt; };
  { const char* v = getenv("ZKWG_RSLB_V"); c->rslb_v = v ? (atoi(v) & 7) : 6; }
  { const char* v = getenv("ZKWG_RSLB_MERGE_LANES"); c->rs_merge_lanes = v && atoi(v) == 4 ? 4 : 1; }
  c->x3_k = pick_k("ZKWG_X3_K", 4, true);
  c->x3_k_o0 = pick_k("ZKWG_X3_K_O0", 1);   // (round 4: 8 KiB pieces, 53 VGPRs = 8 wavefronts per SIMD, software-pipelined over the group's emails)
  c->o0_emails_per_wg = getenv("ZKWG_O0_EMAILS_PER_WG") ? std::max(1, atoi(getenv("ZKWG_O0_EMAILS_PER_WG"))) : 16;
  c->o0_pipe = getenv("ZKWG_O0_PIPE") ? atoi(getenv("ZKWG_O0_PIPE")) : 2;
  c->xcd_remap = getenv("ZKWG_XCD_REMAP") ? (u32)atoi(getenv("ZKWG_XCD_REMAP")) : 1u;
  c->rsa_wgs_per_cu = 0;
  if (const char* v = getenv("ZKWG_RSA_WGS_PER_CU")) c->rsa_wgs_per_cu = atoi(v);
  c->has_net = false;
  if (regex) {
    // the regex circuit is compiled from the supplied template text (zkwg_circom.h)
    if (!regex->circom_path || cfg->main_kind != ZKWG_MAIN_EMAIL_VERIFIER || cfg->ignore_body_hash_check) {
      g_last_error = "a regex template needs an EmailVerifier configuration with the body-hash check";
      delete c;
      return ZKWG_RC_BAD_CONFIG;
    }
    std::string err;
    if (!zkc::load(regex->circom_path, regex->include_dirs ? regex->include_dirs : "",
                   regex->template_name ? regex->template_name : "BodyHashRegex", {(zkc::i64)cfg->max_header}, c->net, err)) {
      g_last_error = "regex template: " + err;
      delete c;
      return ZKWG_RC_BAD_CONFIG;
    }
    // the scan tables against the plain gate list on a few messages, once per handle (zkwg_net_host.h; ZKWG_NET_SELFCHECK=0 skips it)
    if (!getenv("ZKWG_NET_SELFCHECK") || atoi(getenv("ZKWG_NET_SELFCHECK"))) {
      if (!zkc::self_check(regex->circom_path, regex->include_dirs ? regex->include_dirs : "",
                           regex->template_name ? regex->template_name : "BodyHashRegex", {(zkc::i64)cfg->max_header}, c->net, err)) {
        g_last_error = "regex template: " + err;
        delete c;
        return ZKWG_RC_BAD_CONFIG;
      }
    }
    c->has_net = true;
  }
  const zkc::Net* net = c->has_net ? &c->net : nullptr;
  const bool dbg_t = getenv("ZKWG_DEBUG_TIMING") != nullptr;
  auto t_last = std::chrono::steady_clock::now();
  auto phase = [&](const char* what) {
    if (!dbg_t) return;
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[zkwg] create: %-28s %7.2f s\n", what, std::chrono::duration<double>(now - t_last).count());
    t_last = now;
  };
  if (!build_sched(*cfg, c->s, c->segs, net)) { g_last_error = "unsupported circuit configuration"; delete c; return ZKWG_RC_BAD_CONFIG; }
  if (c->has_net) {   // how a slot of the template's region is decoded, with host pointers (host expansion, layout-only handles); the device copy follows
    ZkNetDec& D = c->h_netd;
    D.pd = c->net.pd.data(); D.tab = c->net.tabs.data();
    D.offF = c->net.offF; D.offB = c->net.offB; D.nL = c->net.nL; D.nF = c->net.nF; D.nB = c->net.nB; D.b_fdim = c->net.bchain.fdim;
    D.m_net = c->s.m_net; D.m_net_pw = c->s.m_net_pw; D.n_in = c->net.n_in;
  }
  if (getenv("ZKWG_DEBUG_SKIP_INV") && atoi(getenv("ZKWG_DEBUG_SKIP_INV")) && c->s.rsa.present) c->s.rsa.present = 2;  // profiling only
  if (sym_text) {
    ZkSymLayout L;
    L.allow_holes = r1cs != nullptr;
    // a compact `.sym` (no .r1cs): zk_expand writes the file's order directly (segments remapped).  A fully numbered
    // circuit (.r1cs given): every wire of the file gets a descriptor over the compact image or a linear row over
    // kept-v1 slots (zkwg_full.h -> zkwg_o0.h); zk_expand3_o0 writes the file's witness in one pass
    phase("schedule");
    // the two files are independent until the linear plan: the .r1cs is parsed on its own thread(s) while the .sym is laid out
    // (both are threaded inside; their serial stretches hide behind each other)
    ZkR1csHost R;
    bool r1cs_ok = true;
    std::thread r1cs_thread;
    std::exception_ptr r1cs_err, sym_err;      // neither thread may leave an exception unjoined: both are rethrown below, joined
    if (r1cs) r1cs_thread = std::thread([&] { try { r1cs_ok = zk_r1cs_parse(r1cs, r1cs_len, R); } catch (...) { r1cs_err = std::current_exception(); } });
    bool sym_ok = false;
    try { sym_ok = zk_sym_layout(c->s, sym_text, sym_len, alias_text, alias_len, L, net); } catch (...) { sym_err = std::current_exception(); }
    if (r1cs_thread.joinable()) r1cs_thread.join();
    if (sym_err || r1cs_err) { delete c; std::rethrow_exception(sym_err ? sym_err : r1cs_err); }      // -> create_guarded's catch (ZKWG_RC_OOM / BAD_CONFIG)
    if (!sym_ok || (!r1cs && !zk_remap_segments(c->s, c->segs, L))) {
      g_last_error = L.err.empty() ? std::string(".sym layout does not tile the witness") : L.err;
      delete c;
      return ZKWG_RC_BAD_CONFIG;
    }
    if (r1cs) {
      // every signal the schedule does not produce must follow from the circuit's own linear constraints
      std::string err;
      std::vector<u8> produced(L.W, 0);
      for (u64 i = 0; i < L.W; ++i) produced[i] = L.hole[i] ? 0 : 1;
      phase(".sym layout | .r1cs parse");
      if (!r1cs_ok) err = "the .r1cs file could not be parsed";
      else if (R.n_wires != L.W) err = "the .r1cs has " + std::to_string(R.n_wires) + " wires, the .sym file numbers " + std::to_string(L.W);
      else zk_linear_plan(R, produced, c->lin_host, err);
      phase("linear plan");
      if (err.empty()) {
        // wire -> where its value comes from: a kept-v1 slot (bit 31 clear) or a linear row over kept-v1 slots
        const ZkLinPlan& Pn = c->lin_host;
        std::vector<u32> inv(L.W, 0xffffffffu);
        for (u64 slot = 0; slot < L.dst.size(); ++slot) if (L.dst[slot] != 0xffffffffu) inv[L.dst[slot]] = (u32)slot;
        c->o0_desc.assign(L.W, 0xffffffffu);
        for (u64 w = 0; w < L.W; ++w) if (!L.hole[w]) c->o0_desc[w] = inv[w];
        c->o0_src.resize(Pn.src.size());
        for (size_t t = 0; t < Pn.src.size() && err.empty(); ++t) {
          c->o0_src[t] = inv[Pn.src[t]];
          if (c->o0_src[t] == 0xffffffffu) err = "internal: a linear row reads a wire the schedule does not produce";
        }
        for (u64 r = 0; r < Pn.n_rows(); ++r) {
          const u64 a = Pn.row_ptr[r], b = Pn.row_ptr[r + 1];
          if (b - a == 1 && Pn.kind[a] == ZK_COEF_ONE) c->o0_desc[Pn.dst[r]] = c->o0_src[a];   // alias: the wire copies its source's descriptor
          else { c->o0_desc[Pn.dst[r]] = 0xfffffffeu; (b - a > ZK_O0_SHORT_ROW ? c->o0_long : c->o0_short).push_back((u32)r); }   // zk_o0_rows
        }
        c->n_o0_short = c->o0_short.size();
        c->o0_long.insert(c->o0_long.begin(), c->o0_short.begin(), c->o0_short.end());   // [short rows | long rows]
        std::vector<u32>().swap(c->o0_short);
        for (u64 w = 0; w < L.W && err.empty(); ++w)
          if (c->o0_desc[w] == 0xffffffffu || (!(c->o0_desc[w] >> 31) && c->o0_desc[w] >= c->s.W)) err = "internal: wire " + std::to_string(w) + " has no source";
        if (Pn.n_rows() >= 0x7fffffffull || c->s.W >= 0x7fffffffull) err = "circuit too large for the O0 gather table";
        c->full_W = L.W;
        c->lin_rows = Pn.n_rows();
        phase("wire table");
        if (err.empty()) zk_o0_build(c->s, c->segs, Pn, c->o0_desc, c->o0_src, c->o0t, err);
        phase("descriptor / row tables");
        if (getenv("ZKWG_DEBUG_PLAN"))
          fprintf(stderr, "[zkwg] O0 tables: %llu wires, %llu alias rows, %llu constant rows, %llu small rows (%llu terms in %llu groups), %llu field rows (%llu terms in %llu groups); %llu terms before chaining\n",
                  (unsigned long long)L.W, (unsigned long long)c->o0t.n_alias, (unsigned long long)c->o0t.n_const, (unsigned long long)c->o0t.n_small(),
                  (unsigned long long)c->o0t.s_coef.size(), (unsigned long long)(c->o0t.s_group.size() - 1), (unsigned long long)c->o0t.n_fr(),
                  (unsigned long long)c->o0t.f_kind.size(), (unsigned long long)(c->o0t.f_group.size() - 1), (unsigned long long)c->o0t.terms_before_chaining);
        if (getenv("ZKWG_DEBUG_PLAN")) {
          u64 hist[8] = {0}, terms = 0, longest = 0;
          for (u64 r = 0; r < Pn.n_rows(); ++r) {
            const u64 n = Pn.row_ptr[r + 1] - Pn.row_ptr[r];
            terms += n; longest = std::max(longest, n);
            ++hist[n == 0 ? 0 : n == 1 ? 1 : n == 2 ? 2 : n <= 4 ? 3 : n <= 16 ? 4 : n <= 64 ? 5 : n <= 256 ? 6 : 7];
          }
          std::map<std::string, std::pair<u64, u64>> by;   // template-ish key -> (rows, terms) of the long rows
          for (u64 r = 0; r < Pn.n_rows(); ++r) {
            const u64 n = Pn.row_ptr[r + 1] - Pn.row_ptr[r];
            if (n <= 8) continue;
            std::string nm = L.names[Pn.dst[r]], key;
            for (char ch : nm) if (!isdigit((unsigned char)ch)) key += ch;
            by[key].first++; by[key].second += n;
          }
          std::vector<std::pair<u64, std::string>> top;
          for (auto& kv : by) top.emplace_back(kv.second.second, kv.first + " rows=" + std::to_string(kv.second.first));
          std::sort(top.rbegin(), top.rend());
          for (size_t i = 0; i < top.size() && i < 14; ++i) fprintf(stderr, "[zkwg]   %llu terms: %s\n", (unsigned long long)top[i].first, top[i].second.c_str());
          fprintf(stderr, "[zkwg] O0 plan: %llu rows, %llu terms, longest %llu; rows with 0/1/2/3-4/5-16/17-64/65-256/>256 terms: %llu %llu %llu %llu %llu %llu %llu %llu\n",
                  (unsigned long long)Pn.n_rows(), (unsigned long long)terms, (unsigned long long)longest, (unsigned long long)hist[0], (unsigned long long)hist[1],
                  (unsigned long long)hist[2], (unsigned long long)hist[3], (unsigned long long)hist[4], (unsigned long long)hist[5], (unsigned long long)hist[6], (unsigned long long)hist[7]);
        }
      }
      if (!err.empty()) { g_last_error = err; delete c; return ZKWG_RC_BAD_CONFIG; }
    }
    c->sym_names.swap(L.names);
    c->kept_dst.swap(L.dst);
    c->cfg.layout = ZKWG_LAYOUT_SYM;
  }
  // kernel table (launch order)
  {
    int k = 0;
    c->kname[k++] = "zk_sha_chain";
    c->kname[k++] = "zk_sha_trace";
    if (c->s.net_mode) c->kname[k++] = "zk_net_eval";
    c->kname[k++] = "zk_misc_ev";
    c->kname[k++] = "zk_rsa";
    c->kname[k++] = "zk_poseidon9";
    if (c->s.rslb) { c->kname[k++] = "zk_rslb_chunks"; c->kname[k++] = "zk_rslb_chain"; }
    c->n_kernels = k + 1;
    for (int i = 0; i < k; ++i) c->kslots[i] = 0;
  }
  build_inv_table(c->s.inv_half, c->invtab_host);   // (also the host expansion's table)
  c->kname[c->n_kernels - 1] = "zk_expand"; c->kslots[c->n_kernels - 1] = c->full_W ? c->full_W : c->s.W;   // (zk_expand3_o0 for a fully numbered circuit)
  if (device >= 0) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device >= ndev) { delete c; return ZKWG_RC_NO_DEVICE; }
    if (hipSetDevice(device) != hipSuccess) { delete c; return ZKWG_RC_HIP_ERROR; }
    c->device = device;
    if (zk_misc_init_tables() != 0) { delete c; return ZKWG_RC_HIP_ERROR; }
    std::vector<Fr>& tab = c->invtab_host;
    bool ok = hipMalloc((void**)&c->d_invtab, tab.size() * sizeof(Fr)) == hipSuccess &&
              hipMalloc((void**)&c->d_segs, c->segs.size() * sizeof(ZkSeg)) == hipSuccess;
    {
      std::vector<ZkPortionEntry> ent;
      zk_build_entries(c->s.W, c->segs, ent, 256u * (u32)c->x3_k);
      c->n_ent = (u32)ent.size();
      ok = ok && hipMalloc((void**)&c->d_ent, ent.size() * sizeof(ZkPortionEntry)) == hipSuccess &&
           hipMemcpy(c->d_ent, ent.data(), ent.size() * sizeof(ZkPortionEntry), hipMemcpyHostToDevice) == hipSuccess;
    }
    ok = ok && hipMemcpy(c->d_invtab, tab.data(), tab.size() * sizeof(Fr), hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(c->d_segs, c->segs.data(), c->segs.size() * sizeof(ZkSeg), hipMemcpyHostToDevice) == hipSuccess;
    if (ok && c->has_net) {
      ok = hipMalloc((void**)&c->d_net_records, c->net.records.size() * 4) == hipSuccess &&
           hipMemcpy(c->d_net_records, c->net.records.data(), c->net.records.size() * 4, hipMemcpyHostToDevice) == hipSuccess &&
           hipMalloc((void**)&c->d_net_counts, c->net.step_count.size() * 4 + 64) == hipSuccess &&
           hipMemcpy(c->d_net_counts, c->net.step_count.data(), c->net.step_count.size() * 4, hipMemcpyHostToDevice) == hipSuccess;
      auto upw = [&](const std::vector<u32>& v, u32** dst) {
        if (!ok) return;
        ok = hipMalloc((void**)dst, std::max<size_t>(v.size(), 4) * 4) == hipSuccess && (v.empty() || hipMemcpy(*dst, v.data(), v.size() * 4, hipMemcpyHostToDevice) == hipSuccess);
      };
      upw(c->net.mask_tab, &c->d_net_mask_tab); upw(c->net.pd, &c->d_net_pd); upw(c->net.tabs, &c->d_net_tabs);
      upw(c->net.chain.mask, &c->d_net_cmask);
      upw(c->net.bchain.mask, &c->d_net_bmask);
      {
        ZkNetDec D = c->h_netd;
        D.pd = c->d_net_pd; D.tab = c->d_net_tabs;
        ok = ok && hipMalloc((void**)&c->d_netd, sizeof(ZkNetDec)) == hipSuccess && hipMemcpy(c->d_netd, &D, sizeof(D), hipMemcpyHostToDevice) == hipSuccess;
      }
      auto upb = [&](const std::vector<u8>& v, u8** dst) {
        if (!ok) return;
        ok = hipMalloc((void**)dst, std::max<size_t>(v.size(), 16)) == hipSuccess && (v.empty() || hipMemcpy(*dst, v.data(), v.size(), hipMemcpyHostToDevice) == hipSuccess);
      };
      upb(c->net.chain.cls, &c->d_net_cclass); upb(c->net.chain.delta, &c->d_net_cdelta);
      upb(c->net.bchain.cls, &c->d_net_bclass); upb(c->net.bchain.delta, &c->d_net_bdelta);
      if (ok) for (zkc::ChainTab* t : {&c->net.chain, &c->net.bchain}) { std::vector<u32>().swap(t->tab); std::vector<u32>().swap(t->mask); }
      if (ok) std::vector<u32>().swap(c->net.fn_tab);
      if (ok) std::vector<u32>().swap(c->net.records);
    }
    if (ok && c->full_W) {
      // numbered circuit: descriptors + row tables on the device (the host copies of the plan are dropped)
      ok = upload_o0(c, c->o0t, c->o0d, c->full_W);
      if (ok) {
        ZkLinPlan empty; std::swap(c->lin_host, empty);
        std::vector<u32>().swap(c->o0_desc); std::vector<u32>().swap(c->o0_src); std::vector<u32>().swap(c->o0_long);
      }
    }
    if (ok && c->s.main_kind == ZKWG_MAIN_EMAIL_VERIFIER) {
      std::vector<Fr> C, M, t10;
      build_poseidon_constants(10, 8, 60, C, M);
      ok = zk_build_poseidon_sparse(10, 60, C, M, t10);
      c->pos_dense_off = (u32)t10.size();
      t10.insert(t10.end(), C.begin(), C.end());     // dense tables for zk_poseidon9_wave
      t10.insert(t10.end(), M.begin(), M.end());
      {
        // additive constants of the sparse rounds in Montgomery form (zk_poseidon9_g16 keeps its state in Montgomery form)
        const u32 T = 10, RP = 60;
        const size_t c_part = 4 * T + T * T, c_last = c_part + RP * T + RP * (2 * T - 1) + T * T;
        std::vector<Fr> cm;
        for (u32 i = 0; i < 4 * T; ++i) cm.push_back(fr_to_mont(t10[i]));
        for (u32 i = 0; i < RP * T; ++i) cm.push_back(fr_to_mont(t10[c_part + i]));
        for (u32 i = 0; i < 4 * T; ++i) cm.push_back(fr_to_mont(t10[c_last + i]));
        t10.insert(t10.end(), cm.begin(), cm.end());
      }
      ok = ok && hipMalloc((void**)&c->d_pos, t10.size() * sizeof(Fr)) == hipSuccess &&
           hipMemcpy(c->d_pos, t10.data(), t10.size() * sizeof(Fr), hipMemcpyHostToDevice) == hipSuccess;
    }
    if (ok && c->s.rslb) {
      std::vector<Fr> C, M, t16, t2;
      build_poseidon_constants(17, 8, 68, C, M);
      ok = zk_build_poseidon_sparse(17, 68, C, M, t16);
      build_poseidon_constants(3, 8, 57, C, M);
      ok = ok && zk_build_poseidon_sparse(3, 57, C, M, t2);
      std::vector<u32> l29, l29_2;
      if (ok) { zk_build_poseidon29(17, 68, t16, l29); zk_build_poseidon29(3, 57, t2, l29_2); }
      const size_t n16 = l29.size();
      c->pos2_l29_off = (u32)n16;
      l29.insert(l29.end(), l29_2.begin(), l29_2.end());
      ok = ok && hipMalloc((void**)&c->d_pos_l29, l29.size() * sizeof(u32)) == hipSuccess &&
           hipMemcpy(c->d_pos_l29, l29.data(), l29.size() * sizeof(u32), hipMemcpyHostToDevice) == hipSuccess;
      { const char* v = getenv("ZKWG_RSLB_CONST_CHUNKS");
        if (ok && !(v && atoi(v) == 0)) {
          // Poseidon(16)(0, ..., 0): the same evaluator on the host (element j, limb l at st[l * 17 + j])
          std::vector<u32> st0(9 * 17, 0u);
          std::vector<Fr> z(ZK_P16_KEPT + 1);
          z[ZK_P16_KEPT] = zk_poseidon29<17, 0>(st0.data(), 1, 17, l29.data(), 68, z.data());
          ok = hipMalloc((void**)&c->d_rs_zero, z.size() * sizeof(Fr)) == hipSuccess &&
               hipMemcpy(c->d_rs_zero, z.data(), z.size() * sizeof(Fr), hipMemcpyHostToDevice) == hipSuccess;
        } }
      c->pos2_off = (u32)t16.size();
      t16.insert(t16.end(), t2.begin(), t2.end());
      ok = ok && hipMalloc((void**)&c->d_pos_rs, t16.size() * sizeof(Fr)) == hipSuccess &&
           hipMemcpy(c->d_pos_rs, t16.data(), t16.size() * sizeof(Fr), hipMemcpyHostToDevice) == hipSuccess;
    }
    if (!ok) {
      hipFree(c->d_pos); hipFree(c->d_pos_rs); hipFree(c->d_pos_l29); hipFree(c->d_rs_zero);
      hipFree(c->d_invtab); hipFree(c->d_segs); hipFree(c->d_ent);
      delete c;
      return ZKWG_RC_OOM;
    }
    hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking);
    if (c->s.rslb) {
      // lowest priority: ROCm keeps separate hardware-queue pools per stream priority, so the side streams
      // never share a queue with the caller's (normal / high priority) prepare and expand streams -- a shared
      // queue would put the next batch's kernels behind a 0.2 s chain.  The chain has 1 wave per 64 emails;
      // dispatch priority does not slow it down once resident.
      int prio_lo = 0, prio_hi = 0;
      hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
      for (int i = 0; i < ZK_RS_SLOTS; ++i) {
        hipStreamCreateWithPriority(&c->side_stream[i], hipStreamNonBlocking, prio_lo);
        hipEventCreateWithFlags(&c->rs_dep[i], hipEventDisableTiming);
        hipEventCreateWithFlags(&c->rs_done[i], hipEventDisableTiming);
        c->rs_scr[i] = nullptr;
      }
      c->rs_next = 0;
      c->rs_sync = getenv("ZKWG_RSLB_SYNC") ? atoi(getenv("ZKWG_RSLB_SYNC")) : 0;
      c->rs_nside = getenv("ZKWG_RSLB_SIDE_STREAMS") ? std::min(ZK_RS_SLOTS, std::max(1, atoi(getenv("ZKWG_RSLB_SIDE_STREAMS")))) : 4;
    }
    {
      int prio_lo = 0, prio_hi = 0;
      hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
      for (int i = 0; i < ZK_POS_STREAMS; ++i) hipStreamCreateWithPriority(&c->pos_stream[i], hipStreamNonBlocking, prio_lo);
      for (int i = 0; i < ZK_POS_RING; ++i) {
        hipEventCreateWithFlags(&c->pos_dep[i], hipEventDisableTiming);
        hipEventCreateWithFlags(&c->pos_done[i], hipEventDisableTiming);
      }
      c->pos_calls = 0;
      c->pos_side = getenv("ZKWG_POS_SIDE") ? atoi(getenv("ZKWG_POS_SIDE")) : 0;   // measured: inline is faster in the pipeline
      c->pos_lane = getenv("ZKWG_POS_LANE") ? atoi(getenv("ZKWG_POS_LANE")) : 0;
      c->pos_wave_below = getenv("ZKWG_POS_WAVE_BELOW") ? atoi(getenv("ZKWG_POS_WAVE_BELOW")) : 1024;
    }
    for (int i = 0; i < 2; ++i) { hipEventCreateWithFlags(&c->hb_done[i], hipEventDisableTiming); hipEventCreateWithFlags(&c->hb_copied[i], hipEventDisableTiming); }
    for (int r = 0; r < ZK_EV_RING; ++r) {
      for (int i = 0; i < 2; ++i) hipEventCreate(&c->ev[r][i]);
      for (int i = 0; i <= ZK_MAX_KERNELS; ++i) hipEventCreate(&c->pev[r][i]);
    }
  }
  *out = c;
  return ZKWG_RC_OK;
}

// no C++ exception may cross the C ABI (ctypes / N-API callers would abort)
static int create_guarded(const zkwg_config* cfg, int device, const char* sym_text, uint64_t sym_len,
                          const char* alias_text, uint64_t alias_len, zkwg_circuit_t** out,
                          const uint8_t* r1cs = nullptr, uint64_t r1cs_len = 0, const zkwg_regex_source* regex = nullptr) {
  try {
    return create_impl(cfg, device, sym_text, sym_len, alias_text, alias_len, out, r1cs, r1cs_len, regex);
  } catch (const std::bad_alloc&) {
    g_last_error = "out of host memory while building the circuit";
    return ZKWG_RC_OOM;
  } catch (const std::exception& e) {
    g_last_error = std::string("circuit construction failed: ") + e.what();
    return ZKWG_RC_BAD_CONFIG;
  }
}
int zkwg_circuit_create(const zkwg_config* cfg, int device, zkwg_circuit_t** out) {
  if (cfg && cfg->layout != ZKWG_LAYOUT_KEPT_V1) { g_last_error = "layout SYM needs zkwg_circuit_create_sym"; return ZKWG_RC_BAD_CONFIG; }
  return create_guarded(cfg, device, nullptr, 0, nullptr, 0, out);
}
int zkwg_circuit_create_sym(const zkwg_config* cfg, int device, const char* sym_text, uint64_t sym_len,
                            const char* alias_text, uint64_t alias_len, zkwg_circuit_t** out) {
  if (!sym_text) return ZKWG_RC_BAD_ARG;
  return create_guarded(cfg, device, sym_text, sym_len, alias_text, alias_len, out);
}

int zkwg_circuit_create_full(const zkwg_config* cfg, int device, const char* sym_text, uint64_t sym_len,
                             const char* alias_text, uint64_t alias_len, const uint8_t* r1cs, uint64_t r1cs_len,
                             zkwg_circuit_t** out) {
  if (!sym_text || !r1cs) return ZKWG_RC_BAD_ARG;
  return create_guarded(cfg, device, sym_text, sym_len, alias_text, alias_len, out, r1cs, r1cs_len);
}
int zkwg_circuit_create_regex(const zkwg_config* cfg, int device, const zkwg_regex_source* regex,
                              const char* sym_text, uint64_t sym_len, const char* alias_text, uint64_t alias_len,
                              const uint8_t* r1cs, uint64_t r1cs_len, zkwg_circuit_t** out) {
  if (!regex || !regex->circom_path) return ZKWG_RC_BAD_ARG;
  if (r1cs && !sym_text) return ZKWG_RC_BAD_ARG;
  return create_guarded(cfg, device, sym_text, sym_len, alias_text, alias_len, out, r1cs, r1cs_len, regex);
}
int zkwg_regex_info(const zkwg_circuit_t* c, uint64_t out[8]) {
  if (!c || !out) return ZKWG_RC_BAD_ARG;
  if (!c->has_net) return ZKWG_RC_BAD_CONFIG;
  const zkc::Net& n = c->net;
  out[0] = n.n_kept; out[1] = n.n_temp; out[2] = n.n_gates; out[3] = n.n_asserts;
  out[4] = n.n_chunks; out[5] = n.n_steps; out[6] = n.n_pins; out[7] = n.n_general;
  return ZKWG_RC_OK;
}
// layout-only handles (device < 0): evaluate the linear completion of one host witness in place (tests)
int zkwg_linear_complete_host(const zkwg_circuit_t* c, uint8_t* witness) {
  if (!c || !witness) return ZKWG_RC_BAD_ARG;
  const ZkLinPlan& Pn = c->lin_host;
  if (c->lin_rows && !Pn.n_rows()) return ZKWG_RC_BAD_CONFIG;   // a device handle keeps its plan on the device only
  for (u64 r = 0; r < Pn.n_rows(); ++r)
    ((Fr*)witness)[Pn.dst[r]] = zk_linear_row(Pn.row_ptr.data(), Pn.src.data(), Pn.coef.data(), Pn.kind.data(), r, (const Fr*)witness);
  return ZKWG_RC_OK;
}
// layout-only handles: the linear plan of a fully numbered circuit (copies + rows over the kept-v1 witness) on the host -- every
// wire of the file from one compact kept-v1 witness through the wire table (tests)
int zkwg_o0_gather_host(const zkwg_circuit_t* c, const uint8_t* kept_witness, uint8_t* out) {
  if (!c || !kept_witness || !out) return ZKWG_RC_BAD_ARG;
  if (!c->full_W || c->o0_desc.size() != c->full_W) return ZKWG_RC_BAD_CONFIG;
  const ZkLinPlan& Pn = c->lin_host;
  const Fr* kw = (const Fr*)kept_witness;
  Fr* o = (Fr*)out;
  for (u64 w = 0; w < c->full_W; ++w) {
    const u32 d = c->o0_desc[w];
    if (d != 0xfffffffeu) o[w] = kw[d];
  }
  for (u32 r : c->o0_long)   // [short rows | long rows]
    o[Pn.dst[r]] = zk_linear_row(Pn.row_ptr.data(), c->o0_src.data(), Pn.coef.data(), Pn.kind.data(), r, kw);
  return ZKWG_RC_OK;
}
// scratch buffer of an n-email batch: [SHA chaining states | bits | small | fr (+256) | removeSoftLineBreaks: zk_rslb_chunks' dense-mix
// staging, 153 words per 16-byte chunk, word-major (zkwg_poseidon29.h) | its two unit lists + counters | Montgomery copies (fr + limbs)].
// The Montgomery copies come LAST (round 6): a caller that never asks this buffer for a Montgomery-form output allocates
// zkwg_scratch_bytes_standard -- half the bytes where the image is mostly field elements (removeSoftLineBreaks: 22 instead of 45 GB per
// 4,096 emails, i.e. twice as many prepared batches in flight in the same HBM).
struct ZkScratchLayout { u64 off_hst, off_bits, off_small, off_fr, off_img_end, off_frm, off_rs_stage, off_rs_list, rs_units, total_std, total; };
static ZkScratchLayout scratch_layout(const ZkSched& s, u64 n) {
  ZkScratchLayout L;
  u64 off = 0;
  L.off_hst = off; off += align256(n * (u64)s.hstates_per_email * 32);
  L.off_bits = off; off += align256(n * (u64)s.img_bits * 8);
  L.off_small = off; off += align256(n * (u64)s.img_small * 4);
  L.off_fr = off; off += align256(n * (u64)s.img_fr * 32) + 256;
  L.off_img_end = off;
  L.rs_units = s.rslb ? (n * (u64)s.rs_nch + 63) / 64 * 64 : 0;
  L.off_rs_stage = off; off += align256(L.rs_units * 153 * 4);
  L.off_rs_list = off; off += s.rslb ? align256(L.rs_units * 2 * 4) + 256 : 0;      // constant chunks: two unit lists | their two counters
  L.total_std = off;
  L.off_frm = off; off += align256(n * (u64)(s.img_fr + ZK_MONT_LIMBS) * 32);
  L.total = off;
  return L;
}
int zkwg_image_layout(const zkwg_circuit_t* c, uint64_t n, zkwg_image_layout_t* o) {
  if (!c || !o) return ZKWG_RC_BAD_ARG;
  const ZkSched& s = c->s;
  o->hstate_words = (u64)s.hstates_per_email * 8; o->bits_words = s.img_bits; o->small_words = s.img_small; o->fr_elems = s.img_fr;
  const ZkScratchLayout L = scratch_layout(s, n);
  o->off_hstates = L.off_hst; o->off_bits = L.off_bits; o->off_small = L.off_small; o->off_fr = L.off_fr;
  o->total_bytes = L.total;
  return ZKWG_RC_OK;
}
uint64_t zkwg_segment_table(const zkwg_circuit_t* c, zkwg_segment* out, uint64_t cap) {
  if (!c) return 0;
  static_assert(sizeof(zkwg_segment) == sizeof(ZkSeg), "public segment struct must mirror ZkSeg");
  if (out) memcpy(out, c->segs.data(), std::min<u64>(cap, c->segs.size()) * sizeof(ZkSeg));
  return c->segs.size();
}
uint32_t zkwg_inverse_table_half(const zkwg_circuit_t* c) { return c ? c->s.inv_half : 0; }
uint64_t zkwg_layout_map(const zkwg_circuit_t* c, uint32_t* out, uint64_t cap) {
  if (!c) return 0;
  if (out) for (u64 i = 0; i < c->kept_dst.size() && i < cap; ++i) out[i] = c->kept_dst[i];
  return c->kept_dst.size();
}
uint64_t zkwg_linear_rows(const zkwg_circuit_t* c) { return c ? (c->lin_rows ? c->lin_rows : c->lin_host.n_rows()) : 0; }

void zkwg_circuit_destroy(zkwg_circuit_t* c) {
  if (!c) return;
  if (c->device >= 0) {
    hipSetDevice(c->device);
    hipFree(c->d_invtab); hipFree(c->d_segs); hipFree(c->d_ent); hipFree(c->d_pos); hipFree(c->d_pos_rs); hipFree(c->d_pos_l29); hipFree(c->d_rs_zero); hipFree(c->d_rtab); hipFree(c->d_invtab_m);
    free_o0(c->o0d); free_o0(c->abcd);
    hipFree(c->d_net_records); hipFree(c->d_net_counts); hipFree(c->d_net_mask_tab); hipFree(c->d_net_pd); hipFree(c->d_net_tabs); hipFree(c->d_netd);
    hipFree(c->d_net_cclass); hipFree(c->d_net_cdelta); hipFree(c->d_net_cmask);
    hipFree(c->d_net_bclass); hipFree(c->d_net_bdelta); hipFree(c->d_net_bmask);
    hipFree(c->hb_in); hipFree(c->hb_out[0]); hipFree(c->hb_out[1]); hipFree(c->hb_scr); hipFree(c->hb_status[0]); hipFree(c->hb_status[1]);
    hipFree(c->rp_in); hipFree(c->rp_scr[0]); hipFree(c->rp_scr[1]); rp_free_ring(c); hipFree(c->rp_status);
    if (c->rp_exp) hipStreamDestroy(c->rp_exp);
    for (int i = 0; i < 2; ++i) { if (c->rp_prep_done[i]) hipEventDestroy(c->rp_prep_done[i]); if (c->rp_exp_done[i]) hipEventDestroy(c->rp_exp_done[i]); }
    for (int i = 0; i < 2; ++i) { hipEventDestroy(c->hb_done[i]); hipEventDestroy(c->hb_copied[i]); if (c->hx_img[i]) hipHostFree(c->hx_img[i]); }
    hipStreamDestroy(c->copy_stream);
    hipStreamDestroy(c->own_stream);
    for (int i = 0; i < ZK_POS_STREAMS; ++i) { hipStreamSynchronize(c->pos_stream[i]); hipStreamDestroy(c->pos_stream[i]); }
    for (int i = 0; i < ZK_POS_RING; ++i) { hipEventDestroy(c->pos_dep[i]); hipEventDestroy(c->pos_done[i]); }
    if (c->s.rslb) {
      for (int i = 0; i < ZK_RS_SLOTS; ++i) {
        hipStreamSynchronize(c->side_stream[i]);
        hipEventDestroy(c->rs_dep[i]); hipEventDestroy(c->rs_done[i]);
        hipStreamDestroy(c->side_stream[i]);
      }
    }
    for (int r = 0; r < ZK_EV_RING; ++r) {
      for (int i = 0; i < 2; ++i) hipEventDestroy(c->ev[r][i]);
      for (int i = 0; i <= ZK_MAX_KERNELS; ++i) hipEventDestroy(c->pev[r][i]);
    }
  }
  delete c;
}

static inline u64 out_W(const zkwg_circuit* c) { return c->full_W ? c->full_W : c->s.W; }   // witness length the caller sees
uint64_t zkwg_witness_len(const zkwg_circuit_t* c) { return out_W(c); }
uint64_t zkwg_witness_bytes(const zkwg_circuit_t* c) { return out_W(c) * 32; }
uint32_t zkwg_num_public(const zkwg_circuit_t* c) { return c->s.n_public; }
uint64_t zkwg_input_stride(const zkwg_circuit_t* c) { return c->s.in_stride; }
uint64_t zkwg_input_offset(const zkwg_circuit_t* c, int field) {
  if (field < 0 || field >= ZKWG_IN_NFIELDS) return (uint64_t)-1;
  return c->s.in_off[field];
}
uint64_t zkwg_scratch_bytes(const zkwg_circuit_t* c, uint64_t n) { return scratch_layout(c->s, n).total; }
uint64_t zkwg_scratch_bytes_standard(const zkwg_circuit_t* c, uint64_t n) { return scratch_layout(c->s, n).total_std; }

int zkwg_pack_input(const zkwg_circuit_t* c, uint8_t* rec, const uint8_t* header, uint32_t header_len,
                    const uint8_t* body, uint32_t body_len, const uint8_t* pre, const uint8_t* pubkey,
                    const uint8_t* sig, const uint8_t* msg, uint32_t bh_index) {
  if (!c || !rec) return ZKWG_RC_BAD_ARG;
  const ZkSched& s = c->s;
  memset(rec, 0, s.in_stride);
  if (header) memcpy(rec + s.in_off[ZKWG_IN_HEADER], header, c->cfg.max_header);
  if (body) memcpy(rec + s.in_off[ZKWG_IN_BODY], body, c->cfg.max_body);
  if (pre) memcpy(rec + s.in_off[ZKWG_IN_PRECOMPUTED_SHA], pre, 32);
  if (pubkey) memcpy(rec + s.in_off[ZKWG_IN_PUBKEY], pubkey, 17 * 16);
  if (sig) memcpy(rec + s.in_off[ZKWG_IN_SIGNATURE], sig, 17 * 16);
  if (msg) memcpy(rec + s.in_off[ZKWG_IN_MESSAGE], msg, 17 * 16);
  memcpy(rec + s.in_off[ZKWG_IN_HEADER_LEN], &header_len, 4);
  memcpy(rec + s.in_off[ZKWG_IN_BODY_LEN], &body_len, 4);
  memcpy(rec + s.in_off[ZKWG_IN_BODY_HASH_INDEX], &bh_index, 4);
  return ZKWG_RC_OK;
}

int zkwg_pack_field(const zkwg_circuit_t* c, uint8_t* rec, int field, uint64_t first, const uint8_t* values32,
                    uint64_t count) {
  if (!c || !rec || !values32 || field < 0 || field >= ZKWG_IN_RANGE_FLAGS) return ZKWG_RC_BAD_ARG;
  const ZkSched& s = c->s;
  u64 cap = 0;
  u32 width = 1;   // bytes per packed element
  switch (field) {
    case ZKWG_IN_HEADER: cap = c->cfg.max_header; break;
    case ZKWG_IN_BODY: cap = c->cfg.max_body; break;
    case ZKWG_IN_PRECOMPUTED_SHA: cap = 32; break;
    case ZKWG_IN_PUBKEY: case ZKWG_IN_SIGNATURE: case ZKWG_IN_MESSAGE: cap = 17; width = 16; break;
    case ZKWG_IN_HEADER_LEN: case ZKWG_IN_BODY_LEN: case ZKWG_IN_BODY_HASH_INDEX: cap = 1; width = 4; break;
    case ZKWG_IN_HEADER_MASK: cap = s.mask_header ? c->cfg.max_header : 0; break;
    case ZKWG_IN_BODY_MASK: cap = s.mask_body ? c->cfg.max_body : 0; break;
    case ZKWG_IN_DECODED_BODY: cap = s.rslb ? c->cfg.max_body : 0; break;
  }
  if (first > cap || count > cap - first) return ZKWG_RC_BAD_ARG;
  u32 flags;
  memcpy(&flags, rec + s.in_off[ZKWG_IN_RANGE_FLAGS], 4);
  const Fr p = fr_p();
  for (u64 i = 0; i < count; ++i) {
    Fr v;
    memcpy(v.l, values32 + 32 * i, 32);
    while (fr_geq(v, p)) { u64 bw; v = fr_sub_raw(v, p, bw); }   // circom_runtime: normalize(BigInt(v), prime)
    u8* dst = rec + s.in_off[field] + (first + i) * width;
    bool fits;
    if (width == 1) { fits = !(v.l[0] >> 8) && !v.l[1] && !v.l[2] && !v.l[3]; dst[0] = (u8)v.l[0]; }
    else if (width == 4) { fits = !(v.l[0] >> 32) && !v.l[1] && !v.l[2] && !v.l[3]; const u32 w = (u32)v.l[0]; memcpy(dst, &w, 4); }
    else { fits = !v.l[2] && !v.l[3]; memcpy(dst, v.l, 16); }
    if (!fits) flags |= 1u << field;
  }
  memcpy(rec + s.in_off[ZKWG_IN_RANGE_FLAGS], &flags, 4);
  return ZKWG_RC_OK;
}

int zkwg_pack_masks(const zkwg_circuit_t* c, uint8_t* rec, const uint8_t* header_mask, const uint8_t* body_mask) {
  if (!c || !rec) return ZKWG_RC_BAD_ARG;
  if (header_mask && c->s.mask_header) memcpy(rec + c->s.in_off[ZKWG_IN_HEADER_MASK], header_mask, c->cfg.max_header);
  if (body_mask && c->s.mask_body) memcpy(rec + c->s.in_off[ZKWG_IN_BODY_MASK], body_mask, c->cfg.max_body);
  return ZKWG_RC_OK;
}
int zkwg_pack_decoded_body(const zkwg_circuit_t* c, uint8_t* rec, const uint8_t* decoded_body) {
  if (!c || !rec || !decoded_body || !c->s.rslb) return ZKWG_RC_BAD_ARG;
  memcpy(rec + c->s.in_off[ZKWG_IN_DECODED_BODY], decoded_body, c->cfg.max_body);
  return ZKWG_RC_OK;
}
int zkwg_set_prepare_throttle(zkwg_circuit_t* c, int rsa_wavefronts_per_cu) {
  if (!c || rsa_wavefronts_per_cu < 0) return ZKWG_RC_BAD_ARG;
  c->rsa_wgs_per_cu = rsa_wavefronts_per_cu;
  return ZKWG_RC_OK;
}
int zkwg_set_prepare_mask(zkwg_circuit_t* c, uint32_t kernel_mask) {
  if (!c) return ZKWG_RC_BAD_ARG;
  std::lock_guard<std::mutex> lock(c->dev_mutex);
  c->prep_mask = kernel_mask;
  return ZKWG_RC_OK;
}
int zkwg_set_timing(zkwg_circuit_t* c, int enable) {
  if (!c) return ZKWG_RC_BAD_ARG;
  std::lock_guard<std::mutex> lock(c->dev_mutex);
  c->timing = enable;
  c->ev_valid = false;
  c->prep_valid = false;
  c->launches = 0;
  c->prep_launches = 0;
  return ZKWG_RC_OK;
}
int zkwg_num_kernels(const zkwg_circuit_t* c) { return c->n_kernels; }
const char* zkwg_kernel_name(const zkwg_circuit_t* c, int which) {
  return (which >= 0 && which < c->n_kernels) ? c->kname[which] : "";
}
uint64_t zkwg_kernel_slots(const zkwg_circuit_t* c, int which) {
  return (which >= 0 && which < c->n_kernels) ? c->kslots[which] : 0;
}
// events of kernel `which` in the k-th most recent recorded launch (k = 0: last); false if none
static bool timing_events(zkwg_circuit* c, int which, u64 k, hipEvent_t& a, hipEvent_t& b) {
  if (which == c->n_kernels - 1) {
    if (!c->ev_valid || k >= c->launches || k >= ZK_EV_RING) return false;
    hipEvent_t* ev = c->ev[(c->launches - 1 - k) % ZK_EV_RING];
    a = ev[0]; b = ev[1];
  } else {
    if (!c->prep_valid || k >= c->prep_launches || k >= ZK_EV_RING) return false;
    hipEvent_t* ev = c->pev[(c->prep_launches - 1 - k) % ZK_EV_RING];
    a = ev[which]; b = ev[which + 1];
  }
  return true;
}
int zkwg_last_kernel_ms(zkwg_circuit_t* c, int which, float* ms) {
  if (!c || !ms || which < 0 || which >= c->n_kernels || !c->timing) return ZKWG_RC_BAD_ARG;
  hipEvent_t a, b;
  if (!timing_events(c, which, 0, a, b)) return ZKWG_RC_BAD_ARG;
  if (hipEventSynchronize(b) != hipSuccess) return ZKWG_RC_HIP_ERROR;
  if (hipEventElapsedTime(ms, a, b) != hipSuccess) return ZKWG_RC_HIP_ERROR;
  return ZKWG_RC_OK;
}
int zkwg_timing_summary(zkwg_circuit_t* c, int which, float* total_ms, uint32_t* launches) {
  if (!c || !total_ms || !launches || which < 0 || which >= c->n_kernels || !c->timing) return ZKWG_RC_BAD_ARG;
  float tot = 0.f;
  u64 k = 0;
  hipEvent_t a, b;
  for (; timing_events(c, which, k, a, b); ++k) {
    float ms = 0.f;
    if (hipEventSynchronize(b) != hipSuccess) return ZKWG_RC_HIP_ERROR;
    if (hipEventElapsedTime(&ms, a, b) != hipSuccess) return ZKWG_RC_HIP_ERROR;
    tot += ms;
  }
  *total_ms = tot;
  *launches = (uint32_t)k;
  return ZKWG_RC_OK;
}

static void rp_free_ring(zkwg_circuit* c) {
  for (int k = 0; k < 2; ++k) {
    if (!c->rp_out[k]) continue;
    if (c->rp_chunked) zkwg_device_free_chunked(c->rp_out[k]); else hipFree(c->rp_out[k]);
    c->rp_out[k] = nullptr;
  }
}
static void fill_bufs(const zkwg_circuit* c, ZkBufs& B, const void* d_in, u64 n, void* d_scratch) {
  const ZkSched& s = c->s;
  u8* scr = (u8*)d_scratch;
  const ZkScratchLayout L = scratch_layout(s, n);
  B.in = (const u8*)d_in;
  B.hst = (u32*)(scr + L.off_hst);
  B.bits = (u64*)(scr + L.off_bits);
  B.small = (u32*)(scr + L.off_small);
  B.frv = (Fr*)(scr + L.off_fr);
  B.frm = (Fr*)(scr + L.off_frm);
  B.invtab = c->d_invtab;
  B.pos_c = c->d_pos;
  B.pos_m = c->d_pos ? c->d_pos + c->pos_dense_off : nullptr;
  B.rtab = c->d_rtab;
  B.invtab_m = c->d_invtab_m;
  B.net_records = c->d_net_records;
  B.net_counts = c->d_net_counts;
  B.net_mask_tab = c->d_net_mask_tab;
  B.net_cclass = c->d_net_cclass; B.net_cdelta = c->d_net_cdelta; B.net_cmask = c->d_net_cmask;
  B.net_bclass = c->d_net_bclass; B.net_bdelta = c->d_net_bdelta; B.net_bmask = c->d_net_bmask;
  B.pos16 = c->d_pos_rs;
  B.pos16_l29 = c->d_pos_l29;
  B.pos2_l29 = c->d_pos_l29 ? c->d_pos_l29 + c->pos2_l29_off : nullptr;
  { static const u32 prio = getenv("ZKWG_RSLB_MERGE_PRIO") ? (u32)atoi(getenv("ZKWG_RSLB_MERGE_PRIO")) : 1u; B.rs_prio = prio; }
  B.rs_stage = (u32*)(scr + L.off_rs_stage);
  B.rs_units = L.rs_units;
  B.rs_zero = c->d_rs_zero;
  B.rs_list = c->d_rs_zero ? (u32*)(scr + L.off_rs_list) : nullptr;
  B.rs_cnt = c->d_rs_zero ? (u32*)(scr + L.off_rs_list + align256(L.rs_units * 2 * 4)) : nullptr;
  B.pos2 = c->d_pos_rs ? c->d_pos_rs + c->pos2_off : nullptr;
  B.segs = c->d_segs;
  B.wit = nullptr;
  B.wit_stride16 = 0;
  B.status = nullptr;
  B.n_emails = (u32)n;
  B.e_first = 0;
  B.xcd_remap = c->xcd_remap;
}

static void fill_x3(const zkwg_circuit* c, const ZkBufs& B, ZkX3& A) {
  const ZkSched& s = c->s;
  A.in = B.in; A.bits = B.bits; A.small = B.small; A.frv = B.frv; A.invtab = B.invtab;
  A.ent = c->d_ent; A.segs = c->d_segs; A.wit = B.wit;
  A.frm = B.frm; A.invtab_m = B.invtab_m; A.rtab = B.rtab;
  A.frm_w = B.frm; A.small_w = B.small; A.frv_w = B.frv;
  A.wit_stride16 = B.wit_stride16; A.W = s.W;
  A.in_stride = s.in_stride; A.img_bits = s.img_bits; A.img_small = s.img_small; A.img_fr = s.img_fr; A.inv_half = s.inv_half;
  A.m_dfa_cm = s.m_dfa_cm; A.m_dfa_pm = s.m_dfa_pm; A.m_dfa_st = s.m_dfa_st;
  A.netd = c->d_netd;
  A.nportions = c->n_ent; A.nsegs = s.nsegs; A.e_first = B.e_first; A.n_count = B.n_emails - B.e_first;
  A.xcd_remap = c->xcd_remap; A.limb_off = s.in_off[ZKWG_IN_PUBKEY];
}

// the linear rows of a numbered (`--O0`) circuit that are real sums, for emails [B.e_first, B.n_emails): results go into the image
// extensions (zkwg_o0.h).  Launched at the end of zkwg_prepare_device -- so that in a two-stream pipeline they overlap the
// previous sub-batch's expansion -- on the caller's stream, or behind the removeSoftLineBreaks chain on its side stream (the
// chain writes field elements the rows read; the expansion waits for that stream anyway).
static void launch_o0_rows(const zkwg_circuit* c, const ZkO0Dev& O, const ZkBufs& B, hipStream_t st) {
  ZkX3 A;
  fill_x3(c, B, A);
  const u32 total = B.n_emails - B.e_first;
  for (u32 off = 0; off < total; off += 32768u) {   // the kernels index emails with blockIdx.y
    const u32 cnt = std::min(32768u, total - off);
    A.e_first = B.e_first + off; A.n_count = cnt;
    // zk_o0_generic first: the rows read the codes it leaves
    if (O.n_gen) hipLaunchKernelGGL(zk_o0_generic, dim3((O.n_gen + 255) / 256, cnt), dim3(256), 0, st, A, O);
    if (O.n_small_single) hipLaunchKernelGGL(zk_o0_rows_small, dim3((O.n_small_single + 255) / 256, (cnt + ZK_ROW_EMAILS - 1) / ZK_ROW_EMAILS), dim3(256), 0, st, A, O);
    if (O.n_small_long) hipLaunchKernelGGL(zk_o0_rows_small_long, dim3((O.n_small_long + 3) / 4, (cnt + ZK_ROW_EMAILS - 1) / ZK_ROW_EMAILS), dim3(256), 0, st, A, O);
    if (O.n_small_chains) hipLaunchKernelGGL(zk_o0_chains_small, dim3(O.n_small_chains, cnt), dim3(64), 0, st, A, O);
    if (O.n_fr_groups) hipLaunchKernelGGL(zk_o0_rows_fr, dim3(O.n_fr_groups, (cnt + 63u) / 64u), dim3(64), 0, st, A, O);   // a wavefront = one row x 64 emails
  }
}

int zkwg_prepare_device(zkwg_circuit_t* c, const void* d_in, uint64_t n, void* d_status, void* d_scratch,
                        void* hip_stream) {
  if (!c || !d_in || !d_status || !d_scratch) return ZKWG_RC_BAD_ARG;
  if (c->device < 0) return ZKWG_RC_NO_DEVICE;
  if (n == 0) return ZKWG_RC_OK;
  const ZkSched& s = c->s;
  if (n > 0x3fffffffull) return ZKWG_RC_BAD_ARG;
  if (((uintptr_t)d_scratch & 255) || ((uintptr_t)d_in & 15)) return ZKWG_RC_BAD_ARG;
  std::lock_guard<std::mutex> lock(c->dev_mutex);
  ZkDeviceGuard dg(c->device);
  if (!dg.ok) return ZKWG_RC_HIP_ERROR;
  hipStream_t st = (hipStream_t)hip_stream;
  const u32 ne = (u32)n;
  ZkBufs B;
  fill_bufs(c, B, d_in, n, d_scratch);
  B.status = (int*)d_status;
  const bool tm = c->timing != 0;
  if (s.rslb && !c->rs_sync)   // a merge chain of an earlier batch may still be reading this scratch buffer
    for (int i = 0; i < ZK_RS_SLOTS; ++i)
      if (c->rs_scr[i] == d_scratch) hipStreamWaitEvent(st, c->rs_done[i], 0);
  if (hipMemsetAsync(B.status, 0, n * sizeof(int), st) != hipSuccess) return ZKWG_RC_HIP_ERROR;
  int ki = 0;
  hipEvent_t* evs = c->pev[c->prep_launches % ZK_EV_RING];
  const u32 pm = c->prep_mask;
  const bool pos9 = s.rsa.present && s.main_kind == ZKWG_MAIN_EMAIL_VERIFIER && (pm & 32u);
  int pos_slot = -1;
  if (pos9 && c->pos_side) {
    // fork: zk_poseidon9 only reads the input record; it overlaps the SHA / regex / RSA kernels (with timing
    // on, its entry measures the time the caller's stream waits at the join, i.e. the exposed latency)
    pos_slot = (int)(c->pos_calls % ZK_POS_RING);
    hipStream_t ps = c->pos_stream[c->pos_calls % ZK_POS_STREAMS];
    c->pos_calls++;
    hipEventRecord(c->pos_dep[pos_slot], st);
    hipStreamWaitEvent(ps, c->pos_dep[pos_slot], 0);
    if ((int)ne < c->pos_wave_below) hipLaunchKernelGGL(zk_poseidon9_wave, dim3(ne), dim3(64), 0, ps, s, B);
    else if (c->pos_lane) hipLaunchKernelGGL(zk_poseidon9, dim3((ne + 63) / 64), dim3(64), 0, ps, s, B);
    else hipLaunchKernelGGL(zk_poseidon9_g16, dim3((ne + 3) / 4), dim3(64), 0, ps, s, B);
    hipEventRecord(c->pos_done[pos_slot], ps);
  }
  if (tm) hipEventRecord(evs[ki], st);
  if (s.nframes && (pm & 1u)) {
    u32 threads = ne * s.nframes;
    hipLaunchKernelGGL(zk_sha_chain, dim3((threads + 63) / 64), dim3(64), 0, st, s, B);
  }
  if (tm) hipEventRecord(evs[++ki], st);
  if (s.nframes && (pm & 2u)) {
    u64 units = (u64)ne * s.total_blocks;
    hipLaunchKernelGGL(zk_sha_trace, dim3((u32)((units + 63) / 64)), dim3(64), 0, st, s, B);
  }
  if (tm) hipEventRecord(evs[++ki], st);
  if (s.net_mode) {
    // the regex circuit of a loaded template: gate list, one wavefront per email (LDS: value cache + message bytes)
    const u32 ew = 64u / std::max(16u, s.net_lanes);   // emails per wavefront
    if (4u * s.net_lds_words * ew + 16 > 48u * 1024u)   // (gfx950: 160 KB of LDS per CU; the default per-workgroup cap is lower)
      hipFuncSetAttribute((const void*)zk_net_eval, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(4u * s.net_lds_words * ew + 16));
    if (pm & 4u) {
      if (s.net_chain_end) hipLaunchKernelGGL(zk_net_scan, dim3((ne + 63) / 64), dim3(64), 0, st, s, B);   // chain states first: the list's mask words need them
      hipLaunchKernelGGL(zk_net_eval, dim3((ne + ew - 1) / ew), dim3(64), 4u * s.net_lds_words * ew + 16, st, s, B);
    }
    if (tm) hipEventRecord(evs[++ki], st);
  }
  if (s.body && (pm & 8u)) hipLaunchKernelGGL(zk_misc_ev, dim3(ne), dim3(64), 7 * s.fr[0].max_bytes + 64 + ZK_DFA_STATES * 256 + 16, st, s, B);
  if (tm) hipEventRecord(evs[++ki], st);
  if (s.rsa.present && (pm & 16u)) {
    // optional throttle: pad the workgroup's LDS claim so that only `rsa_wgs_per_cu` RSA wavefronts
    // (164 VGPRs each) are resident per CU, leaving registers/slots to a concurrently running zk_expand
    u32 dyn = 0;
    if (c->rsa_wgs_per_cu > 0) {
      const u32 per = (160u * 1024u) / (u32)c->rsa_wgs_per_cu;
      dyn = per > 14u * 1024u ? std::min(per - 14u * 1024u, 50u * 1024u) : 0u;
    }
    hipLaunchKernelGGL(zk_rsa, dim3(ne), dim3(64), dyn, st, s, B);
  }
  if (s.fpg.present && (pm & 16u)) hipLaunchKernelGGL(zk_fpmul_small, dim3((ne + 63) / 64), dim3(64), 0, st, s, B);   // (timed in zk_rsa's slot)
  if (tm) hipEventRecord(evs[++ki], st);
  if (pos9 && pos_slot < 0) {
    // one lane per email once the batch supplies >= 16 wavefronts of them; one wavefront per email below
    if ((int)ne < c->pos_wave_below) hipLaunchKernelGGL(zk_poseidon9_wave, dim3(ne), dim3(64), 0, st, s, B);
    else if (c->pos_lane) hipLaunchKernelGGL(zk_poseidon9, dim3((ne + 63) / 64), dim3(64), 0, st, s, B);
    else hipLaunchKernelGGL(zk_poseidon9_g16, dim3((ne + 3) / 4), dim3(64), 0, st, s, B);
  }
  if (pos_slot >= 0) hipStreamWaitEvent(st, c->pos_done[pos_slot], 0);   // join
  if (s.rslb) {
    // removeSoftLineBreaks: chunk hashes (one lane per 16-byte chunk), then the serial merge chain + scans
    if (tm) hipEventRecord(evs[++ki], st);
    const u64 units = (u64)ne * s.rs_nch;
    if (pm & 64u) {
      const dim3 g((u32)((units + 63) / 64));
      if (B.rs_list) {     // constant chunks: split the units, copy the constant ones' signals, hash the others (the grid covers the worst case)
        hipMemsetAsync(B.rs_cnt, 0, 8, st);
        hipLaunchKernelGGL(zk_rslb_classify, dim3((u32)((units + 255) / 256)), dim3(256), 0, st, s, B);
        hipLaunchKernelGGL(zk_rslb_fill_const, dim3(8192), dim3(64), 0, st, s, B);
      }
      switch (c->rslb_v) {
        case 1: hipLaunchKernelGGL(zk_rslb_chunks_v1, g, dim3(64), 0, st, s, B); break;
        case 2: hipLaunchKernelGGL(zk_rslb_chunks_v2, g, dim3(64), 0, st, s, B); break;
        case 3: hipLaunchKernelGGL(zk_rslb_chunks_v3, g, dim3(64), 0, st, s, B); break;
        case 4: hipLaunchKernelGGL(zk_rslb_chunks_v4, g, dim3(64), 0, st, s, B); break;
        case 5: hipLaunchKernelGGL(zk_rslb_chunks_v5, g, dim3(64), 0, st, s, B); break;
        case 6: hipLaunchKernelGGL(zk_rslb_chunks_v6, g, dim3(64), 0, st, s, B); break;
        case 7: hipLaunchKernelGGL(zk_rslb_chunks_v7, g, dim3(64), 0, st, s, B); break;
        default: hipLaunchKernelGGL(zk_rslb_chunks_v0, g, dim3(64), 0, st, s, B); break;
      }
    }
    if (tm) hipEventRecord(evs[++ki], st);
    if (c->rs_sync) {
      if (pm & 128u) { if (c->rs_merge_lanes == 1) hipLaunchKernelGGL(zk_rslb_merge1, dim3((ne + 63u) / 64u), dim3(64), 0, st, s, B); else hipLaunchKernelGGL(zk_rslb_merge, dim3((ne + 64u / ZK_RS_MERGE_LANES - 1u) / (64u / ZK_RS_MERGE_LANES)), dim3(64), 0, st, s, B); hipLaunchKernelGGL(zk_rslb_scan, dim3((ne + 63) / 64), dim3(64), 0, st, s, B); }
    } else {
      int slot = -1;
      for (int i = 0; i < ZK_RS_SLOTS; ++i) if (c->rs_scr[i] == d_scratch) slot = i;
      if (slot < 0) { slot = c->rs_next; c->rs_next = (c->rs_next + 1) % ZK_RS_SLOTS; c->rs_scr[slot] = d_scratch; }
      hipEventRecord(c->rs_dep[slot], st);
      // four side streams by default (ZKWG_RSLB_SIDE_STREAMS): a batch's chain then starts when its chunk hashes are done instead of
      // behind the chain of the batch before last (round 6, same box: 53.3 k witnesses/s against 51.9 k with two,
      // profiles/r06/r06_n_rslb_variants.json).  They are lowest-priority streams: their hardware queues come from a pool of their own.
      hipStream_t ss = c->side_stream[slot % c->rs_nside];
      hipStreamWaitEvent(ss, c->rs_dep[slot], 0);
      if (pm & 128u) { if (c->rs_merge_lanes == 1) hipLaunchKernelGGL(zk_rslb_merge1, dim3((ne + 63u) / 64u), dim3(64), 0, ss, s, B); else hipLaunchKernelGGL(zk_rslb_merge, dim3((ne + 64u / ZK_RS_MERGE_LANES - 1u) / (64u / ZK_RS_MERGE_LANES)), dim3(64), 0, ss, s, B); hipLaunchKernelGGL(zk_rslb_scan, dim3((ne + 63) / 64), dim3(64), 0, ss, s, B); }
      if (pm & 256u) {   // the rows read the chain's field elements: they follow it on the side stream
        if (c->full_W) launch_o0_rows(c, c->o0d, B, ss);
        if (c->abc_m) launch_o0_rows(c, c->abcd, B, ss);
      }
      hipEventRecord(c->rs_done[slot], ss);
      if (tm) { hipEventRecord(evs[++ki], ss); c->prep_valid = true; c->prep_launches++; }
      if (hipGetLastError() != hipSuccess) return ZKWG_RC_HIP_ERROR;
      return ZKWG_RC_OK;
    }
  }
  if (!(s.rslb && !c->rs_sync) && (pm & 256u)) {   // (timed with the last prepare kernel)
    if (c->full_W) launch_o0_rows(c, c->o0d, B, st);
    if (c->abc_m) launch_o0_rows(c, c->abcd, B, st);
  }
  if (tm) { hipEventRecord(evs[++ki], st); c->prep_valid = true; c->prep_launches++; }
  if (hipGetLastError() != hipSuccess) return ZKWG_RC_HIP_ERROR;
  return ZKWG_RC_OK;
}

static int expand_impl(zkwg_circuit_t* c, const void* d_in, uint64_t n, const void* d_scratch, uint64_t first,
                       uint64_t count, void* d_out, uint64_t out_stride, void* hip_stream, bool mont, bool abc = false) {
  if (!c || !d_in || !d_out || !d_scratch) return ZKWG_RC_BAD_ARG;
  if (c->device < 0) return ZKWG_RC_NO_DEVICE;
  if (count == 0) return ZKWG_RC_OK;
  const ZkSched& s = c->s;
  if (abc && !c->abc_m) return ZKWG_RC_BAD_CONFIG;
  // the descriptor table written from: the attached system's A.w | B.w | C.w, a numbered circuit's wires, or none (kept-v1 pieces)
  const ZkO0Dev* OD = abc ? &c->abcd : (c->full_W ? &c->o0d : nullptr);
  if (first + count > n || out_stride < (abc ? 3 * c->abc_m : out_W(c)) * 32 || (out_stride & 15)) return ZKWG_RC_BAD_ARG;
  if (((uintptr_t)d_scratch & 255) || ((uintptr_t)d_out & 15)) return ZKWG_RC_BAD_ARG;
  std::lock_guard<std::mutex> lock(c->dev_mutex);
  ZkDeviceGuard dg(c->device);
  if (!dg.ok) return ZKWG_RC_HIP_ERROR;
  hipStream_t st = (hipStream_t)hip_stream;
  if (mont) {
    if (!c->d_rtab || !c->d_invtab_m) {
      // v * R mod r for v < 65536 (2 MiB) and the inverse table in Montgomery form, built once per handle and
      // published together (a half-built pair must never be seen by a later call)
      std::vector<Fr> tab(65536);
      Fr acc = fr_zero();
      const Fr Rm = fr_R();
      for (u32 v = 0; v < 65536; ++v) { tab[v] = acc; acc = fr_add(acc, Rm); }
      std::vector<Fr> inv;
      build_inv_table(s.inv_half, inv);
      for (Fr& x : inv) x = fr_to_mont(x);
      Fr *d_r = nullptr, *d_i = nullptr;
      int rc = ZKWG_RC_OK;
      if (hipMalloc((void**)&d_r, tab.size() * sizeof(Fr)) != hipSuccess || hipMalloc((void**)&d_i, inv.size() * sizeof(Fr)) != hipSuccess) rc = ZKWG_RC_OOM;
      else if (hipMemcpy(d_r, tab.data(), tab.size() * sizeof(Fr), hipMemcpyHostToDevice) != hipSuccess ||
               hipMemcpy(d_i, inv.data(), inv.size() * sizeof(Fr), hipMemcpyHostToDevice) != hipSuccess) rc = ZKWG_RC_HIP_ERROR;
      if (rc != ZKWG_RC_OK) { hipFree(d_r); hipFree(d_i); return rc; }
      hipFree(c->d_rtab); hipFree(c->d_invtab_m);
      c->d_rtab = d_r; c->d_invtab_m = d_i;
    }
  }
  // sub-launches: the O0 row kernels index emails with blockIdx.y; zk_expand3's grid is pieces x emails
  u64 sub = OD ? std::min<u64>(count, 32768) : count;
  { const u64 per = OD ? OD->nportions : c->n_ent; if (per) sub = std::min<u64>(sub, std::max<u64>(1, 0x7fffffffull / per)); }
  ZkBufs B;
  fill_bufs(c, B, d_in, n, (void*)d_scratch);
  if (s.rslb && !c->rs_sync)   // the merge chain of this scratch buffer may still be running on the side stream
    for (int i = 0; i < ZK_RS_SLOTS; ++i)
      if (c->rs_scr[i] == d_scratch) hipStreamWaitEvent(st, c->rs_done[i], 0);
  const bool tm = c->timing != 0;
  hipEvent_t* evs = c->ev[c->launches % ZK_EV_RING];
  if (tm) hipEventRecord(evs[0], st);
  for (u64 off = 0; off < count; off += sub) {
    const u64 cnt = std::min(sub, count - off);
    u8* out_sub = (u8*)d_out + off * out_stride;
    B.wit = (uint4*)out_sub;
    B.wit_stride16 = out_stride / 16;
    B.e_first = (u32)(first + off);
    B.n_emails = (u32)(first + off + cnt);
    ZkX3 A;
    fill_x3(c, B, A);
    const u64 conv = cnt * (u64)(s.img_fr + ZK_MONT_LIMBS);
    if (OD) {
      // numbered circuit (`--O0` / `--O1`), one pass: the rows that are real sums go into the image extensions of these
      // emails, then every wire is written from its descriptor (zkwg_o0.h) -- no staging buffer, no gather
      const ZkO0Dev& O = *OD;
      const u64 units = ((cnt + O.emails_per_wg - 1) / O.emails_per_wg) * (u64)O.nportions;
      if (units > 0x7fffffffull) return ZKWG_RC_BAD_ARG;
      if (mont) hipLaunchKernelGGL(zk_image_to_mont, dim3((u32)((conv + 255) / 256)), dim3(256), 0, st, A);
      const dim3 g3((u32)units), b3(256);
#define ZK_LAUNCH_O0(K) do { if (c->o0_pipe) { if (mont) hipLaunchKernelGGL(zk_expand3_o0p_mont_k##K, g3, b3, 0, st, A, O); else hipLaunchKernelGGL(zk_expand3_o0p_k##K, g3, b3, 0, st, A, O); } \
                             else { if (mont) hipLaunchKernelGGL(zk_expand3_o0_mont_k##K, g3, b3, 0, st, A, O); else hipLaunchKernelGGL(zk_expand3_o0_k##K, g3, b3, 0, st, A, O); } } while (0)
      if (c->x3_k_o0 == 1 && c->o0_pipe == 2) { if (mont) hipLaunchKernelGGL(zk_expand3_o0b_mont_k1, g3, b3, 0, st, A, O); else hipLaunchKernelGGL(zk_expand3_o0b_k1, g3, b3, 0, st, A, O); }
      else if (c->x3_k_o0 == 1 && c->o0_pipe == 3) { if (mont) hipLaunchKernelGGL(zk_expand3_o0c_mont_k1, g3, b3, 0, st, A, O); else hipLaunchKernelGGL(zk_expand3_o0c_k1, g3, b3, 0, st, A, O); }
      else if (c->x3_k_o0 == 1) ZK_LAUNCH_O0(1); else if (c->x3_k_o0 == 2) ZK_LAUNCH_O0(2); else ZK_LAUNCH_O0(4);
#undef ZK_LAUNCH_O0
      continue;
    }
    {
      // one piece of 256 K slots per workgroup (zkwg_kernels_expand3.hip)
      const u64 units3 = cnt * (u64)c->n_ent;
      if (units3 > 0x7fffffffull) return ZKWG_RC_BAD_ARG;
      if (mont) hipLaunchKernelGGL(zk_image_to_mont, dim3((u32)((conv + 255) / 256)), dim3(256), 0, st, A);
      const dim3 g3((u32)units3), b3(256);
#define ZK_LAUNCH_X3(K) do { if (mont) hipLaunchKernelGGL(zk_expand3_mont_k##K, g3, b3, 0, st, A); else hipLaunchKernelGGL(zk_expand3_k##K, g3, b3, 0, st, A); } while (0)
      if (c->x3_k == 1) ZK_LAUNCH_X3(1); else if (c->x3_k == 2) ZK_LAUNCH_X3(2); else if (c->x3_k == 8) ZK_LAUNCH_X3(8); else ZK_LAUNCH_X3(4);
#undef ZK_LAUNCH_X3
    }
  }
  if (tm) { hipEventRecord(evs[1], st); c->ev_valid = true; c->launches++; }
  if (hipGetLastError() != hipSuccess) return ZKWG_RC_HIP_ERROR;
  return ZKWG_RC_OK;
}

int zkwg_expand_device(zkwg_circuit_t* c, const void* d_in, uint64_t n, const void* d_scratch, uint64_t first,
                       uint64_t count, void* d_out, uint64_t out_stride, void* hip_stream) {
  return expand_impl(c, d_in, n, d_scratch, first, count, d_out, out_stride, hip_stream, false);
}
int zkwg_expand_montgomery_device(zkwg_circuit_t* c, const void* d_in, uint64_t n, const void* d_scratch, uint64_t first,
                                  uint64_t count, void* d_out, uint64_t out_stride, void* hip_stream) {
  return expand_impl(c, d_in, n, d_scratch, first, count, d_out, out_stride, hip_stream, true);
}

// ---- the first Groth16 prover stage from the compact image -------------------------------------------
// A.w, B.w, C.w of every constraint are linear combinations over the witness, exactly like the wires of a numbered
// circuit that are sums of others: the 3 m combinations become one more descriptor table (zkwg_o0.h) over the image --
// a combination that is one wire copies that wire's descriptor, one of bits / small integers with small coefficients is
// a 64-bit integer row, the rest are rows mod r -- and zk_expand3_o0 streams them out.  No 32-byte witness is read.
int zkwg_circuit_attach_r1cs(zkwg_circuit_t* c, const uint8_t* r1cs, uint64_t len) {
  if (!c || !r1cs) return ZKWG_RC_BAD_ARG;
  // lock order of the host-buffer path: hb_mutex, then dev_mutex.  Both are held: the attachment grows the image layout (c->s)
  // that the host-path calls read under hb_mutex and the device entry points under dev_mutex.
  std::lock_guard<std::mutex> hb_lock(c->hb_mutex);
  std::lock_guard<std::mutex> lock(c->dev_mutex);
  if (c->abc_m) { g_last_error = "attach_r1cs: the handle already has a constraint system"; return ZKWG_RC_BAD_CONFIG; }
  try {
    ZkR1csHost R;
    if (!zk_r1cs_parse(r1cs, len, R)) { g_last_error = "the .r1cs file could not be parsed: " + R.err; return ZKWG_RC_BAD_CONFIG; }
    // A numbered handle (zkwg_circuit_create_full) is keyed to the compiler's `.r1cs` -- the system a zkey carries
    // (packages/helpers/src/chunked-zkey.ts:80-84): its wires are the file's, not kept-v1 slots.
    const bool numbered = c->full_W != 0;
    const u64 n_wires = numbered ? c->full_W : c->s.W;
    if (R.n_wires != n_wires) { g_last_error = "the .r1cs has " + std::to_string(R.n_wires) + " wires, the witness layout " + std::to_string(n_wires); return ZKWG_RC_BAD_CONFIG; }
    const u64 m = R.n_constraints;
    if (3 * m >= 0x7fffffffull) { g_last_error = "too many constraints"; return ZKWG_RC_BAD_CONFIG; }
    // numbered: wire -> kept-v1 slot it copies, or its definition over kept-v1 slots (the linear plan of zkwg_full.h, derived
    // again from this file: the handle dropped its host copy after building the device tables)
    std::vector<u32> inv;
    std::vector<u64> row_of;
    ZkLinPlan Pn;
    if (numbered) {
      if (c->kept_dst.empty()) { g_last_error = "attach_r1cs: the numbered handle has no layout map"; return ZKWG_RC_BAD_CONFIG; }
      inv.assign(n_wires, 0xffffffffu);
      for (u64 slot = 0; slot < c->kept_dst.size(); ++slot) if (c->kept_dst[slot] != 0xffffffffu && c->kept_dst[slot] < n_wires) inv[c->kept_dst[slot]] = (u32)slot;
      std::vector<u8> produced(n_wires, 0);
      for (u64 w = 0; w < n_wires; ++w) produced[w] = inv[w] != 0xffffffffu;
      std::string perr;
      if (!zk_linear_plan(R, produced, Pn, perr)) { g_last_error = perr; return ZKWG_RC_BAD_CONFIG; }
      row_of.assign(n_wires, ~0ull);
      for (u64 r = 0; r < Pn.n_rows(); ++r) row_of[Pn.dst[r]] = r;
    }
    // output order: the A values, then B, then C (zkwg_r1cs_evaluate_device's)
    ZkLinPlan P;
    std::vector<u32> desc_slot(3 * m);
    P.row_ptr.assign(1, 0);
    P.src.reserve(R.wire.size()); P.coef.reserve(R.wire.size()); P.kind.reserve(R.wire.size());
    const Fr one_m = fr_R(), minus_one_m = fr_neg(fr_R());
    std::vector<std::pair<u32, Fr>> acc;
    for (u32 which = 0; which < 3; ++which)
      for (u64 i = 0; i < m; ++i) {
        const u64 a = R.row_ptr[3 * i + which], b = R.row_ptr[3 * i + which + 1];
        const u64 dst = (u64)which * m + i;
        const u64 t0 = P.src.size();
        if (!numbered) {
          for (u64 t = a; t < b; ++t) {
            if (R.wire[t] >= n_wires) { g_last_error = "the .r1cs names a wire outside the witness"; return ZKWG_RC_BAD_CONFIG; }
            P.src.push_back(R.wire[t]); P.coef.push_back(fr_from_mont(R.coef[t])); P.kind.push_back(R.kind[t]);
          }
        } else {
          // substitute every wire by its kept-v1 source(s), merge equal slots (Montgomery coefficients throughout)
          acc.clear();
          for (u64 t = a; t < b; ++t) {
            const u32 w = R.wire[t];
            if (w >= n_wires) { g_last_error = "the .r1cs names a wire outside the witness"; return ZKWG_RC_BAD_CONFIG; }
            if (inv[w] != 0xffffffffu) { acc.emplace_back(inv[w], R.coef[t]); continue; }
            const u64 r = row_of[w];
            if (r == ~0ull) { g_last_error = "internal: wire " + std::to_string(w) + " has neither a slot nor a definition"; return ZKWG_RC_BAD_CONFIG; }
            for (u64 q = Pn.row_ptr[r]; q < Pn.row_ptr[r + 1]; ++q) {
              const u32 sl = inv[Pn.src[q]];
              if (sl == 0xffffffffu) { g_last_error = "internal: a linear row reads a wire the schedule does not produce"; return ZKWG_RC_BAD_CONFIG; }
              const u8 k = Pn.kind[q];
              acc.emplace_back(sl, k == ZK_COEF_ONE ? R.coef[t] : (k == ZK_COEF_MINUS_ONE ? fr_neg(R.coef[t]) : fr_mont_mul(R.coef[t], fr_to_mont(Pn.coef[q]))));
            }
          }
          if (acc.size() > 1) std::stable_sort(acc.begin(), acc.end(), [](const std::pair<u32, Fr>& x, const std::pair<u32, Fr>& y) { return x.first < y.first; });
          for (size_t t = 0; t < acc.size();) {
            Fr sum = acc[t].second;
            size_t q = t + 1;
            while (q < acc.size() && acc[q].first == acc[t].first) { sum = fr_add(sum, acc[q].second); ++q; }
            if (!fr_is_zero(sum)) {
              P.src.push_back(acc[t].first); P.coef.push_back(fr_from_mont(sum));
              P.kind.push_back(fr_eq(sum, one_m) ? ZK_COEF_ONE : (fr_eq(sum, minus_one_m) ? ZK_COEF_MINUS_ONE : ZK_COEF_GENERIC));
            }
            t = q;
          }
        }
        P.row_ptr.push_back(P.src.size());
        P.dst.push_back((u32)dst);
        desc_slot[dst] = (P.src.size() - t0 == 1 && P.kind[t0] == ZK_COEF_ONE) ? P.src[t0] : 0xfffffffeu;
      }
    { ZkLinPlan e; std::swap(Pn, e); std::vector<u64>().swap(row_of); std::vector<u32>().swap(inv); }
    std::string err;
    ZkSched s2 = c->s;   // (the image grows by the row results: committed only when everything succeeded)
    ZkO0Tables T;         // (built aside: a failed attachment leaves the handle as it was)
    if (!zk_o0_build(s2, c->segs, P, desc_slot, P.src, T, err)) { g_last_error = err; return ZKWG_RC_BAD_CONFIG; }
    if (getenv("ZKWG_DEBUG_PLAN"))
      fprintf(stderr, "[zkwg] A.w|B.w|C.w tables: %llu combinations, %llu single wires, %llu empty, %llu small rows (%llu terms), %llu field rows (%llu terms), %llu pre-decoded slots; image %u -> %u small words, %u -> %u field elements\n",
              (unsigned long long)(3 * m), (unsigned long long)T.n_alias, (unsigned long long)T.n_const, (unsigned long long)T.n_small(),
              (unsigned long long)T.s_coef.size(), (unsigned long long)T.n_fr(), (unsigned long long)T.f_kind.size(),
              (unsigned long long)T.gen_seg.size(), c->s.img_small, s2.img_small, c->s.img_fr, s2.img_fr);
    if (c->device >= 0) {
      ZkDeviceGuard dg(c->device);
      if (!dg.ok) return ZKWG_RC_HIP_ERROR;
      if (!upload_o0(c, T, c->abcd, 3 * m, true)) { free_o0(c->abcd); return ZKWG_RC_OOM; }   // (the host copy serves zkwg_expand_abc_host)
    }
    c->abct = std::move(T);
    c->s = s2;
    c->abc_m = m;
    // the cached staging buffers of the host-buffer path were sized for the old image layout (zkwg_scratch_bytes grew):
    // drop them, the next zkwg_calculate_batch allocates them again
    if (c->device >= 0 && (c->hb_tile || c->hx_bytes)) {
      ZkDeviceGuard dg(c->device);
      hipDeviceSynchronize();
      hipFree(c->hb_in); hipFree(c->hb_out[0]); hipFree(c->hb_out[1]); hipFree(c->hb_scr); hipFree(c->hb_status[0]); hipFree(c->hb_status[1]);
      c->hb_in = c->hb_out[0] = c->hb_out[1] = c->hb_scr = nullptr;
      c->hb_status[0] = c->hb_status[1] = nullptr;
      c->hb_tile = 0;
      for (int i = 0; i < 2; ++i) { if (c->hx_img[i]) hipHostFree(c->hx_img[i]); c->hx_img[i] = nullptr; }
      c->hx_bytes = 0;
    }
    if (c->device >= 0 && c->rp_scr_bytes) {
      ZkDeviceGuard dg(c->device);
      hipDeviceSynchronize();
      hipFree(c->rp_scr[0]); hipFree(c->rp_scr[1]);
      c->rp_scr[0] = c->rp_scr[1] = nullptr; c->rp_scr_bytes = 0;
    }
  } catch (const std::bad_alloc&) {
    return ZKWG_RC_OOM;
  } catch (const std::exception& e) {
    g_last_error = e.what();
    return ZKWG_RC_BAD_CONFIG;
  }
  return ZKWG_RC_OK;
}
uint64_t zkwg_abc_bytes(const zkwg_circuit_t* c) { return c ? 96 * c->abc_m : 0; }
int zkwg_expand_abc_device(zkwg_circuit_t* c, const void* d_in, uint64_t n, const void* d_scratch, uint64_t first, uint64_t count,
                           int montgomery, void* d_abc, uint64_t abc_stride, void* hip_stream) {
  return expand_impl(c, d_in, n, d_scratch, first, count, d_abc, abc_stride, hip_stream, montgomery != 0, true);
}
// The same values written by the host from a host copy of the scratch buffer (standard form): with `rows_on_host` = 0 the image
// comes from zkwg_prepare_device of a handle that already had the system attached (its row results are in the image; only
// the 4.4 MB image crosses PCIe instead of 72 MB of evaluations); with 1 the row tables are evaluated here as well -- the
// whole path of a layout-only handle, which is how the CPU tests check zk_o0_build's tables without a GPU.
static int tables_host(const zkwg_circuit* c, const ZkO0Tables& T, u64 W3, const uint8_t* records, uint64_t n, uint8_t* scratch_host,
                       uint64_t first, uint64_t count, int rows_on_host, uint8_t* out, uint64_t out_stride) {
  if (!c || !records || !scratch_host || !out) return ZKWG_RC_BAD_ARG;
  if (!W3 || T.desc.size() != 2 * W3) return ZKWG_RC_BAD_CONFIG;   // (a device handle of a numbered circuit keeps its tables on the device only)
  const ZkSched& s = c->s;
  if (first + count > n || out_stride < W3 * 32 || (out_stride & 15) || ((uintptr_t)out & 15)) return ZKWG_RC_BAD_ARG;
  const ZkScratchLayout L = scratch_layout(s, n);
  for (u64 el = 0; el < count; ++el) {
    const u64 e = first + el;
    u32* small_w = (u32*)(scratch_host + L.off_small) + e * s.img_small;
    Fr* frv_w = (Fr*)(scratch_host + L.off_fr) + e * s.img_fr;
    ZkCtx cx;
    cx.rec = records + e * s.in_stride; cx.bits = (const u64*)(scratch_host + L.off_bits) + e * s.img_bits; cx.small = small_w;
    cx.half = (int)s.inv_half; cx.m_dfa_cm = s.m_dfa_cm; cx.m_dfa_pm = s.m_dfa_pm; cx.m_dfa_st = s.m_dfa_st;
    cx.nd = c->has_net ? &c->h_netd : nullptr;
    ZkRefSrc R;
    R.frv = (const uint4*)frv_w; R.invtab = (const uint4*)c->invtab_host.data(); R.rec = cx.rec; R.small = cx.small;
    if (rows_on_host) {
      // zk_o0_generic, zk_o0_rows_small / _long / zk_o0_chains_small, zk_o0_rows_fr of zkwg_kernels_expand3.hip, one row at a time
      for (size_t g = 0; g < T.gen_seg.size(); ++g) small_w[T.gen_base + g] = zk_decode_any(c->segs[T.gen_seg[g]], T.gen_r[g], cx);
      long long prev = 0;
      for (u64 j = 0; j < T.n_small(); ++j) {
        long long acc = T.s_chain[j] ? prev : 0;
        for (u64 t = T.s_ptr[j]; t < T.s_ptr[j + 1]; ++t)
          acc += (long long)T.s_coef[t] * zk_code_int(zk_desc_decode(T.s_term[2 * t], T.s_term[2 * t + 1], cx), cx);
        prev = acc;
        const u32 where = T.s_out[j];
        small_w[where & 0x7fffffffu] = (u32)(u64)acc;
        if (where >> 31) small_w[(where & 0x7fffffffu) + 1] = (u32)((u64)acc >> 32);
      }
      for (u64 j = 0; j < T.n_fr(); ++j) {
        Fr acc = fr_zero();
        for (u64 t = T.f_ptr[j]; t < T.f_ptr[j + 1]; ++t) {
          const Fr x = zk_code_value(zk_desc_decode(T.f_term[2 * t], T.f_term[2 * t + 1], cx), R);
          if (T.f_kind[t] == ZK_COEF_ONE) acc = fr_add(acc, x);
          else if (T.f_kind[t] == ZK_COEF_MINUS_ONE) acc = fr_sub(acc, x);
          else acc = fr_add(acc, fr_mont_mul(x, T.f_coefm[t]));
        }
        frv_w[T.fr_base + j] = acc;
      }
    }
    u8* w = out + el * out_stride;
    for (u64 i = 0; i < W3; ++i) zk_host_put(w + 32 * i, zk_wire_code(T.desc[2 * i], T.desc[2 * i + 1], T.aff.data(), cx), R);
  }
  _mm_sfence();
  return ZKWG_RC_OK;
}
int zkwg_expand_abc_host(const zkwg_circuit_t* c, const uint8_t* records, uint64_t n, uint8_t* scratch_host, uint64_t first,
                         uint64_t count, int rows_on_host, uint8_t* out, uint64_t out_stride) {
  if (!c) return ZKWG_RC_BAD_ARG;
  return tables_host(c, c->abct, 3 * c->abc_m, records, n, scratch_host, first, count, rows_on_host, out, out_stride);
}
// layout-only handles of a numbered circuit (zkwg_circuit_create_full): the complete witness from one image, through the very
// descriptor / row tables the device kernels read (tests)
int zkwg_expand_full_host(const zkwg_circuit_t* c, const uint8_t* records, uint64_t n, uint8_t* scratch_host, uint64_t first,
                          uint64_t count, uint8_t* out, uint64_t out_stride) {
  if (!c) return ZKWG_RC_BAD_ARG;
  return tables_host(c, c->o0t, c->full_W, records, n, scratch_host, first, count, 1, out, out_stride);
}

int zkwg_calculate_batch_device(zkwg_circuit_t* c, const void* d_in, uint64_t n, void* d_out,
                                uint64_t out_stride, void* d_status, void* d_scratch, void* hip_stream) {
  if (!c || !d_in || !d_out || !d_status || !d_scratch) return ZKWG_RC_BAD_ARG;
  if (n == 0) return ZKWG_RC_OK;
  int rc = zkwg_prepare_device(c, d_in, n, d_status, d_scratch, hip_stream);
  if (rc != ZKWG_RC_OK) return rc;
  return zkwg_expand_device(c, d_in, n, d_scratch, 0, n, d_out, out_stride, hip_stream);
}

// Device staging buffers of the host-buffer path, cached in the handle (grow-only).
static int ensure_host_path_buffers(zkwg_circuit* c, u64 tile) {
  if (c->hb_tile >= tile) return ZKWG_RC_OK;
  hipFree(c->hb_in); hipFree(c->hb_out[0]); hipFree(c->hb_out[1]); hipFree(c->hb_scr);
  hipFree(c->hb_status[0]); hipFree(c->hb_status[1]);
  c->hb_in = c->hb_out[0] = c->hb_out[1] = c->hb_scr = nullptr;
  c->hb_status[0] = c->hb_status[1] = nullptr;
  c->hb_tile = 0;
  const u64 wbytes = out_W(c) * 32;
  if (hipMalloc((void**)&c->hb_in, tile * c->s.in_stride) != hipSuccess ||
      hipMalloc((void**)&c->hb_out[0], tile * wbytes) != hipSuccess ||
      hipMalloc((void**)&c->hb_out[1], tile * wbytes) != hipSuccess ||
      hipMalloc((void**)&c->hb_scr, zkwg_scratch_bytes(c, tile)) != hipSuccess ||
      hipMalloc((void**)&c->hb_status[0], tile * sizeof(int)) != hipSuccess ||
      hipMalloc((void**)&c->hb_status[1], tile * sizeof(int)) != hipSuccess)
    return ZKWG_RC_OOM;
  c->hb_tile = tile;
  return ZKWG_RC_OK;
}

// d_rows (optional, device memory of the handle's GPU): n x 96 bytes, w[1..3] of every email
// (pubkeyHash, shaHi, shaLo for the EmailVerifier main) captured tile by tile for the result table.
static int calculate_batch_impl(zkwg_circuit_t* c, const uint8_t* packed, uint64_t n, uint8_t* out_wtns,
                                uint64_t out_stride, int32_t* status, uint64_t max_tile, uint8_t* d_rows) {
  if (!c || !packed || !status) return ZKWG_RC_BAD_ARG;
  if (c->device < 0) return ZKWG_RC_NO_DEVICE;
  if (n == 0) return ZKWG_RC_OK;
  const u64 wbytes = out_W(c) * 32;
  if (out_wtns && out_stride < wbytes) return ZKWG_RC_BAD_ARG;
  std::lock_guard<std::mutex> lock(c->hb_mutex);   // one host-path call at a time per handle
  if (hipSetDevice(c->device) != hipSuccess) return ZKWG_RC_HIP_ERROR;
  // tile: two witness buffers must fit; default 256 emails (the D2H of one tile overlaps the kernels of the next)
  size_t free_b = 0, total_b = 0;
  hipMemGetInfo(&free_b, &total_b);
  const u64 have = (u64)free_b + c->hb_tile * (2 * wbytes);
  const u64 per_email = 2 * wbytes + c->s.in_stride + zkwg_scratch_bytes(c, 64) / 64 + 64;
  u64 tile = max_tile ? max_tile : 256;
  tile = std::min<u64>(tile, std::max<u64>(1, (u64)(have * 0.8) / per_email));
  tile = std::min<u64>(tile, n);
  int rc = ensure_host_path_buffers(c, tile);
  if (rc != ZKWG_RC_OK) return rc;
  hipStream_t st = c->own_stream, cs = c->copy_stream;
  u64 t = 0;
  for (u64 base = 0; rc == ZKWG_RC_OK && base < n; base += tile, ++t) {
    const u64 cnt = std::min<u64>(tile, n - base);
    const int b = (int)(t & 1);
    if (t >= 2 && hipStreamWaitEvent(st, c->hb_copied[b], 0) != hipSuccess) { rc = ZKWG_RC_HIP_ERROR; break; }
    if (hipMemcpyAsync(c->hb_in, packed + base * c->s.in_stride, cnt * c->s.in_stride, hipMemcpyHostToDevice, st) != hipSuccess) { rc = ZKWG_RC_HIP_ERROR; break; }
    rc = zkwg_calculate_batch_device(c, c->hb_in, cnt, c->hb_out[b], wbytes, c->hb_status[b], c->hb_scr, st);
    if (rc != ZKWG_RC_OK) break;
    if (d_rows && hipMemcpy2DAsync(d_rows + base * 96, 96, c->hb_out[b] + 32, wbytes, 96, cnt, hipMemcpyDeviceToDevice, st) != hipSuccess) { rc = ZKWG_RC_HIP_ERROR; break; }
    if (hipEventRecord(c->hb_done[b], st) != hipSuccess || hipStreamWaitEvent(cs, c->hb_done[b], 0) != hipSuccess) { rc = ZKWG_RC_HIP_ERROR; break; }
    if (out_wtns) {
      hipError_t e = (out_stride == wbytes)
          ? hipMemcpyAsync(out_wtns + base * out_stride, c->hb_out[b], cnt * wbytes, hipMemcpyDeviceToHost, cs)
          : hipMemcpy2DAsync(out_wtns + base * out_stride, out_stride, c->hb_out[b], wbytes, wbytes, cnt, hipMemcpyDeviceToHost, cs);
      if (e != hipSuccess) { rc = ZKWG_RC_HIP_ERROR; break; }
    }
    if (hipMemcpyAsync(status + base, c->hb_status[b], cnt * sizeof(int), hipMemcpyDeviceToHost, cs) != hipSuccess) { rc = ZKWG_RC_HIP_ERROR; break; }
    if (hipEventRecord(c->hb_copied[b], cs) != hipSuccess) { rc = ZKWG_RC_HIP_ERROR; break; }
  }
  if (hipStreamSynchronize(st) != hipSuccess) rc = rc == ZKWG_RC_OK ? ZKWG_RC_HIP_ERROR : rc;
  if (hipStreamSynchronize(cs) != hipSuccess) rc = rc == ZKWG_RC_OK ? ZKWG_RC_HIP_ERROR : rc;
  return rc;
}


// ------------------------------------------------------------------ host expansion (the delivered rate)
// A witness delivered to HOST memory crosses PCIe at 32 bytes per signal (56.9 MB per email, ~55 GB/s: 963 witnesses/s).
// Its information is the 0.45 MB image.  zkwg_expand_host runs the same segment decoders (zkwg_expand_dec.h, compiled
// for the host) over a downloaded image and writes the witness straight into the caller's buffer with non-temporal
// stores, on `threads` host threads -- the expansion then costs host memory bandwidth instead of PCIe.  Bit-identical
// to zk_expand by construction (same decoders), checked in tests/test_host_expand.py.
int zkwg_expand_host(const zkwg_circuit_t* c, const uint8_t* records, uint64_t n, const uint8_t* scratch_host, uint64_t first,
                     uint64_t count, uint8_t* out, uint64_t out_stride, int threads) {
  if (!c || !records || !scratch_host || !out) return ZKWG_RC_BAD_ARG;
  if (c->full_W) return ZKWG_RC_BAD_CONFIG;      // numbered circuits: their row results are computed on the device
  const ZkSched& s = c->s;
  if (first + count > n || out_stride < s.W * 32 || (out_stride & 15) || ((uintptr_t)out & 15)) return ZKWG_RC_BAD_ARG;
  const ZkScratchLayout L = scratch_layout(s, n);
  const u64* bits = (const u64*)(scratch_host + L.off_bits);
  const u32* small = (const u32*)(scratch_host + L.off_small);
  const Fr* frv = (const Fr*)(scratch_host + L.off_fr);
  const u64 CH = 1u << 16;                                   // slots per work item
  const u64 per = (s.W + CH - 1) / CH, items = count * per;
  std::atomic<u64> next(0);
  auto work = [&]() {
    for (;;) {
      const u64 it = next.fetch_add(1);
      if (it >= items) break;
      const u64 el = it / per, slot0 = (it % per) * CH, slot1 = std::min<u64>(s.W, slot0 + CH), e = first + el;
      ZkCtx cx;
      cx.rec = records + e * s.in_stride; cx.bits = bits + e * s.img_bits; cx.small = small + e * s.img_small;
      cx.half = (int)s.inv_half; cx.m_dfa_cm = s.m_dfa_cm; cx.m_dfa_pm = s.m_dfa_pm; cx.m_dfa_st = s.m_dfa_st;
      cx.nd = c->has_net ? &c->h_netd : nullptr;
        ZkRefSrc R;
      R.frv = (const uint4*)(frv + e * s.img_fr); R.invtab = (const uint4*)c->invtab_host.data(); R.rec = cx.rec; R.small = cx.small;
      u8* w = out + el * out_stride;
      size_t lo = 0, hi = c->segs.size();
      while (hi - lo > 1) { const size_t mid = (lo + hi) / 2; if (c->segs[mid].slot <= slot0) lo = mid; else hi = mid; }
      for (size_t si = lo; si < c->segs.size() && c->segs[si].slot < slot1; ++si) {
        const ZkSeg& sg = c->segs[si];
        const u64 a = std::max<u64>(sg.slot, slot0), b = std::min<u64>(sg.slot + sg.nslots, slot1);
        const u32 r0 = (u32)(a - sg.slot) + sg.r0, cnt = (u32)(b - a);
        u8* dst = w + 32 * a;
        switch (sg.type) {
#define ZK_X(T, D) case T: zk_host_segment<D>(sg, cx, R, r0, cnt, dst); break;
          ZK_FOR_SEG_TYPES(ZK_X)
#undef ZK_X
          default: memset(dst, 0, 32ull * cnt); break;
        }
      }
    }
    _mm_sfence();
  };
  const int T = std::max(1, threads);
  std::vector<std::thread> th;
  for (int i = 1; i < T; ++i) th.emplace_back(work);
  work();
  for (auto& t : th) t.join();
  return ZKWG_RC_OK;
}
int zkwg_set_host_expand(zkwg_circuit_t* c, int threads) {
  if (!c || threads < 0) return ZKWG_RC_BAD_ARG;
  std::lock_guard<std::mutex> lock(c->hb_mutex);
  c->host_expand_threads = threads;
  return ZKWG_RC_OK;
}

// zkwg_calculate_batch with the expansion on the host: per tile H2D records -> prepare kernels -> D2H of the image
// (0.45 MB per email instead of 56.9 MB of witness) into pinned staging; the host threads expand tile t into the
// caller's buffer while the device prepares tile t + 1.
static int calculate_batch_hostexpand(zkwg_circuit_t* c, const uint8_t* packed, uint64_t n, uint8_t* out_wtns,
                                      uint64_t out_stride, int32_t* status, uint64_t max_tile) {
  const ZkSched& s = c->s;
  if (hipSetDevice(c->device) != hipSuccess) return ZKWG_RC_HIP_ERROR;
  const u64 tile = std::min<u64>(max_tile ? max_tile : 256, n);
  const u64 scr_bytes = zkwg_scratch_bytes(c, tile);
  // device: records + scratch + status for two tiles in flight
  u8 *d_in[2] = {nullptr, nullptr}, *d_scr[2] = {nullptr, nullptr};
  int* d_st[2] = {nullptr, nullptr};
  int rc = ZKWG_RC_OK;
  if (c->hx_bytes < scr_bytes) {
    for (int i = 0; i < 2; ++i) { if (c->hx_img[i]) hipHostFree(c->hx_img[i]); c->hx_img[i] = nullptr; }
    c->hx_bytes = 0;
    for (int i = 0; i < 2; ++i) if (hipHostMalloc((void**)&c->hx_img[i], scr_bytes, hipHostMallocDefault) != hipSuccess) rc = ZKWG_RC_OOM;
    if (rc == ZKWG_RC_OK) c->hx_bytes = scr_bytes;
  }
  for (int i = 0; i < 2 && rc == ZKWG_RC_OK; ++i)
    if (hipMalloc((void**)&d_in[i], tile * s.in_stride) != hipSuccess || hipMalloc((void**)&d_scr[i], scr_bytes) != hipSuccess ||
        hipMalloc((void**)&d_st[i], tile * sizeof(int)) != hipSuccess) rc = ZKWG_RC_OOM;
  hipStream_t st = c->own_stream;
  const ZkScratchLayout L = scratch_layout(s, tile);
  const u64 img_lo = L.off_bits, img_hi = L.off_img_end;      // the image arrays (not the SHA chaining states / staging / Montgomery copies)
  auto submit = [&](u64 t) -> int {
    const u64 base = t * tile, cnt = std::min<u64>(tile, n - base);
    const int b = (int)(t & 1);
    if (hipMemcpyAsync(d_in[b], packed + base * s.in_stride, cnt * s.in_stride, hipMemcpyHostToDevice, st) != hipSuccess) return ZKWG_RC_HIP_ERROR;
    // NB the scratch layout depends on the batch size: every tile is prepared as a batch of `tile` emails; the records beyond
    // cnt of a partial last tile are zeroed (their images are never expanded, but the kernels must not chew on
    // uninitialised device memory: ADVICE r3)
    if (cnt < tile && hipMemsetAsync(d_in[b] + cnt * s.in_stride, 0, (tile - cnt) * s.in_stride, st) != hipSuccess) return ZKWG_RC_HIP_ERROR;
    int r = zkwg_prepare_device(c, d_in[b], tile, d_st[b], d_scr[b], st);
    if (r != ZKWG_RC_OK) return r;
    if (s.rslb && !c->rs_sync)
      for (int i = 0; i < ZK_RS_SLOTS; ++i) if (c->rs_scr[i] == d_scr[b]) hipStreamWaitEvent(st, c->rs_done[i], 0);
    if (hipMemcpyAsync(c->hx_img[b] + img_lo, d_scr[b] + img_lo, img_hi - img_lo, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipMemcpyAsync(status + base, d_st[b], cnt * sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipEventRecord(c->hb_done[b], st) != hipSuccess) return ZKWG_RC_HIP_ERROR;
    return ZKWG_RC_OK;
  };
  const u64 ntiles = (n + tile - 1) / tile;
  if (rc == ZKWG_RC_OK) rc = submit(0);
  for (u64 t = 0; rc == ZKWG_RC_OK && t < ntiles; ++t) {
    if (t + 1 < ntiles) rc = submit(t + 1);                     // the device works on tile t + 1 ...
    if (rc != ZKWG_RC_OK) break;
    const int b = (int)(t & 1);
    if (hipEventSynchronize(c->hb_done[b]) != hipSuccess) { rc = ZKWG_RC_HIP_ERROR; break; }
    const u64 base = t * tile, cnt = std::min<u64>(tile, n - base);
    // ... while the host expands tile t (records: the caller's own buffer, indexed from the tile's first email)
    // (the image was laid out for a batch of `tile` emails: zkwg_expand_host derives the same layout from its n argument;
    //  only the records of the cnt emails it expands are read)
    rc = zkwg_expand_host(c, packed + base * s.in_stride, tile, c->hx_img[b], 0, cnt, out_wtns + base * out_stride, out_stride,
                          c->host_expand_threads);
  }
  if (hipStreamSynchronize(st) != hipSuccess && rc == ZKWG_RC_OK) rc = ZKWG_RC_HIP_ERROR;
  for (int i = 0; i < 2; ++i) { hipFree(d_in[i]); hipFree(d_scr[i]); hipFree(d_st[i]); }
  return rc;
}

int zkwg_calculate_batch(zkwg_circuit_t* c, const uint8_t* packed, uint64_t n, uint8_t* out_wtns,
                         uint64_t out_stride, int32_t* status, uint64_t max_tile) {
  if (c && c->device >= 0 && c->host_expand_threads > 0 && out_wtns && packed && status && n && !c->full_W && out_stride >= out_W(c) * 32 && !(out_stride & 15) && !((uintptr_t)out_wtns & 15)) {
    std::lock_guard<std::mutex> lock(c->hb_mutex);
    return calculate_batch_hostexpand(c, packed, n, out_wtns, out_stride, status, max_tile);
  }
  return calculate_batch_impl(c, packed, n, out_wtns, out_stride, status, max_tile, nullptr);
}

// ------------------------------------------------------------------ device-resident pipeline below the boundary
// What bench.py's Pipeline does, for hosts without torch (the Node host of BASELINE.json's north_star): records go to the
// device once, the prepare kernels of sub-batch i + 1 run on one stream while sub-batch i is expanded tile by tile on another,
// witnesses are written into a two-tile ring in HBM -- placed where HBM takes the stores fastest (the
// real expansion is timed into spare candidate tiles once per handle) -- and handed to `consumer` (a device-side prover)
// stream-ordered; only the statuses and the three public outputs per email come back.  d_rows: optional device buffer,
// n x 96 bytes (the multi-device path gathers them over RCCL).
static int calculate_batch_resident_impl(zkwg_circuit_t* c, const uint8_t* packed, uint64_t n, int32_t* status, uint64_t tile_req,
                                         uint64_t prep_req, uint8_t* d_rows, zkwg_tile_fn consumer, void* user) {
  if (!c || !packed || !status) return ZKWG_RC_BAD_ARG;
  if (c->device < 0) return ZKWG_RC_NO_DEVICE;
  if (n == 0) return ZKWG_RC_OK;
  std::lock_guard<std::mutex> lock(c->hb_mutex);
  if (hipSetDevice(c->device) != hipSuccess) return ZKWG_RC_HIP_ERROR;
  const u64 wbytes = out_W(c) * 32, in_stride = c->s.in_stride;
  u64 tile = std::min<u64>(tile_req ? tile_req : 512, n);
  u64 prep = std::max<u64>(tile, std::min<u64>(prep_req ? prep_req : 1024, n));
  prep = prep / tile * tile;
  // fit: two output tiles, two scratch buffers, the records
  size_t free_b = 0, total_b = 0;
  hipMemGetInfo(&free_b, &total_b);
  const u64 held = 2 * c->rp_tile_bytes + 2 * c->rp_scr_bytes + c->rp_in_cap * in_stride;
  while (tile > 1 && 2 * tile * wbytes + 2 * zkwg_scratch_bytes(c, prep) + n * in_stride > (u64)((free_b + held) * 0.9)) { tile = (tile + 1) / 2; prep = std::max(tile, prep / 2 / tile * tile); }
  auto ensure = [&](void** p, u64& have, u64 want) -> bool {
    if (have >= want) return true;
    hipFree(*p); *p = nullptr; have = 0;
    if (hipMalloc(p, want) != hipSuccess) return false;
    have = want;
    return true;
  };
  if (!c->rp_exp) {
    int lo = 0, hi = 0;
    hipDeviceGetStreamPriorityRange(&lo, &hi);
    if (hipStreamCreateWithPriority(&c->rp_exp, hipStreamNonBlocking, hi) != hipSuccess) return ZKWG_RC_HIP_ERROR;
    for (int i = 0; i < 2; ++i) { hipEventCreateWithFlags(&c->rp_prep_done[i], hipEventDisableTiming); hipEventCreateWithFlags(&c->rp_exp_done[i], hipEventDisableTiming); }
  }
  {
    u64 cap = c->rp_in_cap * in_stride;
    if (!ensure((void**)&c->rp_in, cap, n * in_stride)) return ZKWG_RC_OOM;
    c->rp_in_cap = cap / in_stride;
    u64 sc = c->rp_n_cap * sizeof(int);
    if (!ensure((void**)&c->rp_status, sc, n * sizeof(int))) return ZKWG_RC_OOM;
    c->rp_n_cap = sc / sizeof(int);
    const u64 scr_b = zkwg_scratch_bytes(c, prep);
    if (c->rp_scr_bytes < scr_b) {
      hipFree(c->rp_scr[0]); hipFree(c->rp_scr[1]); c->rp_scr[0] = c->rp_scr[1] = nullptr; c->rp_scr_bytes = 0;
      if (hipMalloc((void**)&c->rp_scr[0], scr_b) != hipSuccess || hipMalloc((void**)&c->rp_scr[1], scr_b) != hipSuccess) return ZKWG_RC_OOM;
      c->rp_scr_bytes = scr_b;
    }
  }
  hipStream_t P = c->own_stream, E = c->rp_exp;
  if (hipMemcpyAsync(c->rp_in, packed, n * in_stride, hipMemcpyHostToDevice, P) != hipSuccess) return ZKWG_RC_HIP_ERROR;
  const u64 tile_bytes = tile * wbytes;
  bool place = c->rp_tile_bytes < tile_bytes;      // (a larger tile: the ring is allocated again, and placed again)
  if (place) { rp_free_ring(c); c->rp_tile_bytes = 0; }
  int rc = ZKWG_RC_OK;
  u64 ring = 0;
  const u64 nsub = (n + prep - 1) / prep;
  for (u64 sb = 0; sb < nsub && rc == ZKWG_RC_OK; ++sb) {
    const u64 lo = sb * prep, cnt = std::min<u64>(prep, n - lo);
    const int b = (int)(sb & 1);
    if (sb >= 2) hipStreamWaitEvent(P, c->rp_exp_done[b], 0);      // scratch buffer b is free again
    rc = zkwg_prepare_device(c, c->rp_in + lo * in_stride, cnt, c->rp_status + lo, c->rp_scr[b], P);
    if (rc != ZKWG_RC_OK) break;
    hipEventRecord(c->rp_prep_done[b], P);
    hipStreamWaitEvent(E, c->rp_prep_done[b], 0);
    if (place) {
      // once per handle (and tile size): the two tiles of the output ring, mapped from 1 GiB physical chunks (zkwg_vmm.hip) -- every such
      // buffer takes zk_expand's stores at the rate only the best of round 4's hipMalloc candidates reached (profiles/r05/r05_f_*), with no
      // spare candidates and no transient memory.  ZKWG_PLACE_RING=0: two plain hipMalloc buffers.
      place = false;
      const bool chunked = !(getenv("ZKWG_PLACE_RING") && atoi(getenv("ZKWG_PLACE_RING")) == 0);
      c->rp_chunked = chunked;
      for (int k = 0; k < 2 && rc == ZKWG_RC_OK; ++k) {
        void* p = nullptr;
        if (chunked) {
          // spare candidate chunks (half as many again, at most 16 GiB per tile): each takes a probe fill, the fastest are kept
          const u32 nch = (u32)((tile_bytes + (1ull << 30) - 1) >> 30);
          rc = zkwg_device_alloc_chunked_ex(c->device, tile_bytes, 0, std::min<u32>(nch / 2, 16u), &p, nullptr, 0, nullptr);
          if (rc != ZKWG_RC_OK) rc = zkwg_device_alloc_chunked(c->device, tile_bytes, 0, &p);
        }
        else if (hipMalloc(&p, tile_bytes) != hipSuccess) rc = ZKWG_RC_OOM;
        c->rp_out[k] = (u8*)p;
      }
      c->rp_place_n = 0;
      if (rc != ZKWG_RC_OK) break;
      c->rp_tile_bytes = tile_bytes;
    }
    for (u64 first = 0; first < cnt && rc == ZKWG_RC_OK; first += tile, ++ring) {
      const u64 count = std::min<u64>(tile, cnt - first);
      u8* o = c->rp_out[ring & 1];
      rc = zkwg_expand_device(c, c->rp_in + lo * in_stride, cnt, c->rp_scr[b], first, count, o, wbytes, E);
      if (rc != ZKWG_RC_OK) break;
      if (d_rows && zk_rows_copy_launch(o, wbytes, d_rows + (lo + first) * 96, (u32)count, E) != 0) { rc = ZKWG_RC_HIP_ERROR; break; }
      if (consumer) consumer(user, c->device, o, wbytes, lo + first, count, (void*)E);
    }
    hipEventRecord(c->rp_exp_done[b], E);
  }
  if (rc == ZKWG_RC_OK && hipMemcpyAsync(status, c->rp_status, n * sizeof(int), hipMemcpyDeviceToHost, E) != hipSuccess) rc = ZKWG_RC_HIP_ERROR;
  if (hipStreamSynchronize(P) != hipSuccess && rc == ZKWG_RC_OK) rc = ZKWG_RC_HIP_ERROR;
  if (hipStreamSynchronize(E) != hipSuccess && rc == ZKWG_RC_OK) rc = ZKWG_RC_HIP_ERROR;
  return rc;
}

int zkwg_calculate_batch_resident(zkwg_circuit_t* c, const uint8_t* packed, uint64_t n, int32_t* status, uint8_t* table,
                                  uint64_t tile, uint64_t prep, zkwg_tile_fn consumer, void* user) {
  if (!c || !packed || !status) return ZKWG_RC_BAD_ARG;
  if (c->device < 0) return ZKWG_RC_NO_DEVICE;
  if (n == 0) return ZKWG_RC_OK;
  ZkDeviceGuard dg(c->device);
  if (!dg.ok) return ZKWG_RC_HIP_ERROR;
  u8* d_rows = nullptr;
  if (table && hipMalloc((void**)&d_rows, n * 96) != hipSuccess) return ZKWG_RC_OOM;
  int rc = calculate_batch_resident_impl(c, packed, n, status, tile, prep, d_rows, consumer, user);
  if (rc == ZKWG_RC_OK && table) {
    std::vector<u8> rows(n * 96);
    if (hipMemcpy(rows.data(), d_rows, n * 96, hipMemcpyDeviceToHost) != hipSuccess) rc = ZKWG_RC_HIP_ERROR;
    else for (u64 i = 0; i < n; ++i) { memcpy(table + 100 * i, status + i, 4); memcpy(table + 100 * i + 4, rows.data() + 96 * i, 96); }
  }
  hipFree(d_rows);
  return rc;
}
// The buffers zkwg_calculate_batch_resident keeps in the handle between calls -- records, two scratch buffers, the two-tile witness ring
// (2 x 29 GB at the headline circuit's tile of 512), statuses -- go back to the device; the next call allocates and places them again.
int zkwg_resident_release(zkwg_circuit_t* c) {
  if (!c) return ZKWG_RC_BAD_ARG;
  if (c->device < 0) return ZKWG_RC_OK;
  std::lock_guard<std::mutex> lock(c->hb_mutex);
  if (hipSetDevice(c->device) != hipSuccess) return ZKWG_RC_HIP_ERROR;
  if (c->rp_exp) hipStreamSynchronize(c->rp_exp);
  if (c->own_stream) hipStreamSynchronize(c->own_stream);
  hipFree(c->rp_in); c->rp_in = nullptr; c->rp_in_cap = 0;
  hipFree(c->rp_scr[0]); hipFree(c->rp_scr[1]); c->rp_scr[0] = c->rp_scr[1] = nullptr; c->rp_scr_bytes = 0;
  rp_free_ring(c); c->rp_tile_bytes = 0;
  hipFree(c->rp_status); c->rp_status = nullptr; c->rp_n_cap = 0;
  return ZKWG_RC_OK;
}
// candidate tiles the ring was chosen from (average milliseconds of one tile's expansion into each; kept[2] = the chosen)
int zkwg_resident_placement(const zkwg_circuit_t* c, float* ms, int cap, int kept[2]) {
  if (!c) return 0;
  std::lock_guard<std::mutex> lock(const_cast<zkwg_circuit_t*>(c)->hb_mutex);   // (written by zkwg_calculate_batch_resident under the same lock)
  for (int i = 0; i < c->rp_place_n && i < cap; ++i) if (ms) ms[i] = c->rp_place_ms[i];
  if (kept) { kept[0] = c->rp_place_kept[0]; kept[1] = c->rp_place_kept[1]; }
  return c->rp_place_n;
}

int zkwg_generate_inputs_device(zkwg_circuit_t* c, const zkwg_dkim_batch* b, uint64_t n, void* d_records,
                                void* d_gen_status, void* hip_stream) {
  if (!c || !b || !d_records || !d_gen_status) return ZKWG_RC_BAD_ARG;
  if (c->device < 0) return ZKWG_RC_NO_DEVICE;
  if (c->s.main_kind != ZKWG_MAIN_EMAIL_VERIFIER) return ZKWG_RC_BAD_CONFIG;
  if (n == 0) return ZKWG_RC_OK;
  if (!b->headers || !b->header_len || !b->pubkey_be || !b->signature_be) return ZKWG_RC_BAD_ARG;
  if (c->s.body && (!b->bodies || !b->body_len || !b->body_hash_b64)) return ZKWG_RC_BAD_ARG;
  if (b->selector_len && !b->selector) return ZKWG_RC_BAD_ARG;
  ZkDkimBatch D;
  D.headers = b->headers; D.header_len = b->header_len; D.bodies = b->bodies; D.body_len = b->body_len;
  D.body_hash_b64 = b->body_hash_b64; D.pubkey_be = b->pubkey_be; D.signature_be = b->signature_be;
  D.selector = b->selector; D.header_stride = b->header_stride; D.body_stride = b->body_stride;
  D.selector_len = b->selector_len;
  ZkDeviceGuard dg(c->device);
  if (!dg.ok) return ZKWG_RC_HIP_ERROR;
  hipLaunchKernelGGL(zk_gen_inputs, dim3((u32)n), dim3(64), 0, (hipStream_t)hip_stream, c->s, D, (u8*)d_records,
                     (int*)d_gen_status, (u32)n);
  return hipGetLastError() == hipSuccess ? ZKWG_RC_OK : ZKWG_RC_HIP_ERROR;
}

// Page-locked host memory for witness output buffers: lets the D2H copy of one tile overlap the
// kernels of the next (pageable destinations work too, the copy is then staged by the runtime).
void* zkwg_alloc_pinned(uint64_t bytes) {
  void* p = nullptr;
  return hipHostMalloc(&p, bytes, hipHostMallocDefault) == hipSuccess ? p : nullptr;
}
void zkwg_free_pinned(void* p) { if (p) hipHostFree(p); }


// A stream restricted to a subset of the compute units (hipExtStreamCreateWithCUMask): bit i of `mask` enables CU i.
// bench.py uses it to keep the latency-bound prepare kernels off the CUs zk_expand streams from.
void* zkwg_stream_create_masked(int device, const uint32_t* mask, int words) {
  if (!mask || words <= 0) return nullptr;
  ZkDeviceGuard dg(device);
  if (!dg.ok) return nullptr;
  hipStream_t st = nullptr;
  if (hipExtStreamCreateWithCUMask(&st, (uint32_t)words, mask) != hipSuccess) return nullptr;
  return (void*)st;
}
void zkwg_stream_destroy(void* stream) { if (stream) hipStreamDestroy((hipStream_t)stream); }

uint64_t zkwg_wtns_size(const zkwg_circuit_t* c) { return 12 + 12 + 40 + 12 + out_W(c) * 32; }

int zkwg_write_wtns(const zkwg_circuit_t* c, const uint8_t* witness, uint8_t* out, uint64_t cap) {
  if (!c || !witness || !out) return ZKWG_RC_BAD_ARG;
  if (cap < zkwg_wtns_size(c)) return ZKWG_RC_BAD_ARG;
  u8* p = out;
  auto w32 = [&](u32 v) { memcpy(p, &v, 4); p += 4; };
  auto w64 = [&](u64 v) { memcpy(p, &v, 8); p += 8; };
  memcpy(p, "wtns", 4); p += 4;
  w32(2); w32(2);
  w32(1); w64(40);
  w32(32);
  const u64 prime[4] = {ZK_P0, ZK_P1, ZK_P2, ZK_P3};
  memcpy(p, prime, 32); p += 32;
  w32((u32)out_W(c));
  w32(2); w64(out_W(c) * 32);
  memcpy(p, witness, out_W(c) * 32);
  return ZKWG_RC_OK;
}

uint64_t zkwg_write_sym(const zkwg_circuit_t* c, char* out, uint64_t cap) {
  ZkSched tmp = c->s;
  ZkWalker w;
  w.net = c->has_net ? &c->net : nullptr;
  w.names = true;
  u64 pos = 0;
  auto put = [&](u64 slot, const std::string& name) {
    char buf[64];
    int n = snprintf(buf, sizeof(buf), "%llu,%llu,0,", (unsigned long long)slot, (unsigned long long)slot);
    u64 need = (u64)n + name.size() + 1;
    if (out && pos + need <= cap) {
      memcpy(out + pos, buf, n);
      memcpy(out + pos + n, name.data(), name.size());
      out[pos + n + name.size()] = '\n';
    }
    pos += need;
  };
  if (!c->sym_names.empty()) {
    for (u64 i = 0; i < c->sym_names.size(); ++i) put(i, c->sym_names[i]);
    return pos;
  }
  w.sink = put;
  switch (tmp.main_kind) {
    case ZKWG_MAIN_SHA256_BYTES: zk_walk_main_sha(w, tmp); break;
    case ZKWG_MAIN_RSA_VERIFIER: zk_walk_main_rsa(w, tmp); break;
    case ZKWG_MAIN_FP_MUL: zk_walk_main_fpmul(w, tmp, tmp.fpg.n, tmp.fpg.k); break;
    case ZKWG_MAIN_EMAIL_VERIFIER: zk_walk_main_ev(w, tmp); break;
  }
  return pos;
}

// ------------------------------------------------------------------ multi-device
// One handle + one host thread per GPU, contiguous shards, no data-path collective.  The only exchange is
// the gather of the 100-byte/email result table {status, pubkeyHash, shaHi, shaLo} on devices[0] over
// RCCL (ncclGroupStart + ncclSend / ncclRecv over xGMI).  RCCL is opened with dlopen when n_dev > 1 (and its header
// is not included), so a single-GPU deployment carries no dependency on it.
// The few RCCL declarations the gather needs (the library is opened with dlopen, so a single-GPU deployment
// neither links nor includes RCCL): nccl.h's ncclComm_t, ncclResult_t (0 = ncclSuccess), ncclDataType_t (1 = ncclUint8).
typedef struct ncclComm* zk_nccl_comm_t;
enum { ZK_NCCL_SUCCESS = 0, ZK_NCCL_UINT8 = 1 };
struct zkwg_multi {
  std::vector<zkwg_circuit_t*> h;
  std::vector<int> dev;
  std::vector<zk_nccl_comm_t> comm;
  void* rccl = nullptr;
  int (*CommInitAll)(zk_nccl_comm_t*, int, const int*) = nullptr;
  int (*CommDestroy)(zk_nccl_comm_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*Send)(const void*, size_t, int, int, zk_nccl_comm_t, hipStream_t) = nullptr;
  int (*Recv)(void*, size_t, int, int, zk_nccl_comm_t, hipStream_t) = nullptr;
};

void zkwg_shard_range(uint64_t n, int n_shards, int i, uint64_t* first, uint64_t* count) {
  // contiguous ranges whose sizes differ by at most one (the first n % n_shards shards take the extra email)
  if (n_shards <= 0 || i < 0 || i >= n_shards) { if (first) *first = 0; if (count) *count = 0; return; }
  const uint64_t q = n / (uint64_t)n_shards, r = n % (uint64_t)n_shards;
  const uint64_t f = (uint64_t)i * q + std::min<uint64_t>((uint64_t)i, r);
  if (first) *first = f;
  if (count) *count = q + ((uint64_t)i < r ? 1 : 0);
}

void zkwg_multi_destroy(zkwg_multi_t* m) {
  if (!m) return;
  for (auto cm : m->comm) if (cm && m->CommDestroy) m->CommDestroy(cm);
  for (auto c : m->h) zkwg_circuit_destroy(c);
  if (m->rccl) dlclose(m->rccl);
  delete m;
}

int zkwg_multi_create(const zkwg_config* cfg, const int* devices, int n_dev, zkwg_multi_t** out) {
  if (!cfg || !devices || n_dev <= 0 || !out) return ZKWG_RC_BAD_ARG;
  zkwg_multi* m = new zkwg_multi();
  int rc = ZKWG_RC_OK;
  for (int i = 0; i < n_dev && rc == ZKWG_RC_OK; ++i) {
    zkwg_circuit_t* c = nullptr;
    rc = zkwg_circuit_create(cfg, devices[i], &c);
    if (rc == ZKWG_RC_OK) { m->h.push_back(c); m->dev.push_back(devices[i]); }
  }
  if (rc == ZKWG_RC_OK && n_dev > 1) {
    // ZKWG_RCCL_LIB names the library (a deployment's own build; the single-GPU test of this branch loads a
    // stand-in that implements the six calls with hipMemcpyAsync, tests/native/rccl_stub.cpp)
    const char* lib = getenv("ZKWG_RCCL_LIB");
    if (lib && *lib) m->rccl = dlopen(lib, RTLD_NOW | RTLD_LOCAL);
    else {
      m->rccl = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
      if (!m->rccl) m->rccl = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    }
    if (!m->rccl) { g_last_error = std::string(lib && *lib ? lib : "librccl.so") + " could not be loaded (needed for the result-table gather with n_dev > 1)"; rc = ZKWG_RC_BAD_CONFIG; }
    else {
      m->CommInitAll = (decltype(m->CommInitAll))dlsym(m->rccl, "ncclCommInitAll");
      m->CommDestroy = (decltype(m->CommDestroy))dlsym(m->rccl, "ncclCommDestroy");
      m->GroupStart = (decltype(m->GroupStart))dlsym(m->rccl, "ncclGroupStart");
      m->GroupEnd = (decltype(m->GroupEnd))dlsym(m->rccl, "ncclGroupEnd");
      m->Send = (decltype(m->Send))dlsym(m->rccl, "ncclSend");
      m->Recv = (decltype(m->Recv))dlsym(m->rccl, "ncclRecv");
      if (!m->CommInitAll || !m->CommDestroy || !m->GroupStart || !m->GroupEnd || !m->Send || !m->Recv) {
        g_last_error = "librccl.so lacks a required symbol"; rc = ZKWG_RC_BAD_CONFIG;
      } else {
        m->comm.assign(n_dev, nullptr);
        if (m->CommInitAll(m->comm.data(), n_dev, devices) != ZK_NCCL_SUCCESS) { m->comm.clear(); g_last_error = "ncclCommInitAll failed"; rc = ZKWG_RC_HIP_ERROR; }
      }
    }
  }
  if (rc != ZKWG_RC_OK) { zkwg_multi_destroy(m); return rc; }
  *out = m;
  return ZKWG_RC_OK;
}
int zkwg_multi_devices(const zkwg_multi_t* m) { return m ? (int)m->h.size() : 0; }
zkwg_circuit_t* zkwg_multi_circuit(zkwg_multi_t* m, int i) { return (m && i >= 0 && i < (int)m->h.size()) ? m->h[i] : nullptr; }

int zkwg_calculate_batch_multi(zkwg_multi_t* m, const uint8_t* packed, uint64_t n, uint8_t* out_wtns,
                               uint64_t out_stride, int32_t* status, uint8_t* table, uint64_t max_tile) {
  if (!m || !packed || !status || m->h.empty()) return ZKWG_RC_BAD_ARG;
  if (n == 0) return ZKWG_RC_OK;
  const int nd = (int)m->h.size();
  const uint64_t in_stride = zkwg_input_stride(m->h[0]);
  std::vector<uint8_t*> d_rows(nd, nullptr), d_tab(nd, nullptr);
  std::vector<int32_t*> d_st(nd, nullptr);
  std::vector<int> rcs(nd, ZKWG_RC_OK);
  std::vector<uint64_t> first(nd), cnt(nd);
  for (int i = 0; i < nd; ++i) zkwg_shard_range(n, nd, i, &first[i], &cnt[i]);
  auto worker = [&](int i) {
    if (cnt[i] == 0) return;
    if (hipSetDevice(m->dev[i]) != hipSuccess) { rcs[i] = ZKWG_RC_HIP_ERROR; return; }
    if (table && hipMalloc((void**)&d_rows[i], cnt[i] * 96) != hipSuccess) { rcs[i] = ZKWG_RC_OOM; return; }
    // out_wtns = NULL: nothing but the result table leaves the GPUs -- the device-resident two-stream pipeline (witnesses into
    // each GPU's own placed ring); otherwise each shard's witnesses leave through that GPU's PCIe link
    if (!out_wtns) rcs[i] = calculate_batch_resident_impl(m->h[i], packed + first[i] * in_stride, cnt[i], status + first[i], max_tile, 0, d_rows[i], nullptr, nullptr);
    else rcs[i] = calculate_batch_impl(m->h[i], packed + first[i] * in_stride, cnt[i], out_wtns + first[i] * out_stride, out_stride, status + first[i],
                                       max_tile, d_rows[i]);
    if (rcs[i] != ZKWG_RC_OK || !table) return;
    // device-side table rows of this shard: {status i32, 96 bytes}
    if (hipMalloc((void**)&d_tab[i], cnt[i] * 100) != hipSuccess || hipMalloc((void**)&d_st[i], cnt[i] * 4) != hipSuccess) { rcs[i] = ZKWG_RC_OOM; return; }
    hipStream_t st = m->h[i]->own_stream;
    bool ok = hipMemcpyAsync(d_st[i], status + first[i], cnt[i] * 4, hipMemcpyHostToDevice, st) == hipSuccess &&
              hipMemcpy2DAsync(d_tab[i], 100, d_st[i], 4, 4, cnt[i], hipMemcpyDeviceToDevice, st) == hipSuccess &&
              hipMemcpy2DAsync(d_tab[i] + 4, 100, d_rows[i], 96, 96, cnt[i], hipMemcpyDeviceToDevice, st) == hipSuccess &&
              hipStreamSynchronize(st) == hipSuccess;
    if (!ok) rcs[i] = ZKWG_RC_HIP_ERROR;
  };
  int caller_dev = -1;
  if (hipGetDevice(&caller_dev) != hipSuccess) caller_dev = -1;   // worker(0) and the gather change the calling thread's device
  {
    std::vector<std::thread> th;
    for (int i = 1; i < nd; ++i) th.emplace_back(worker, i);
    worker(0);
    for (auto& t : th) t.join();
  }
  int rc = ZKWG_RC_OK;
  for (int i = 0; i < nd; ++i) if (rcs[i] != ZKWG_RC_OK) rc = rcs[i];
  if (rc == ZKWG_RC_OK && table) {
    // gather on devices[0]: rank i sends its rows, rank 0 receives them at the shard's offset
    uint8_t* d_all = nullptr;
    if (hipSetDevice(m->dev[0]) != hipSuccess || hipMalloc((void**)&d_all, n * 100) != hipSuccess) rc = ZKWG_RC_OOM;
    if (rc == ZKWG_RC_OK) {
      hipStream_t s0 = m->h[0]->own_stream;
      bool ok = cnt[0] == 0 || hipMemcpyAsync(d_all, d_tab[0], cnt[0] * 100, hipMemcpyDeviceToDevice, s0) == hipSuccess;
      if (nd > 1 && ok) {
        ok = m->GroupStart() == ZK_NCCL_SUCCESS;
        if (ok) {
          // once a group is open it is always closed, whatever the calls inside returned
          for (int i = 1; i < nd && ok; ++i) {
            if (cnt[i] == 0) continue;
            ok = m->Recv(d_all + first[i] * 100, cnt[i] * 100, ZK_NCCL_UINT8, i, m->comm[0], s0) == ZK_NCCL_SUCCESS &&
                 m->Send(d_tab[i], cnt[i] * 100, ZK_NCCL_UINT8, 0, m->comm[i], m->h[i]->own_stream) == ZK_NCCL_SUCCESS;
          }
          ok = (m->GroupEnd() == ZK_NCCL_SUCCESS) && ok;
        }
        for (int i = 1; i < nd; ++i) { if (hipSetDevice(m->dev[i]) != hipSuccess || hipStreamSynchronize(m->h[i]->own_stream) != hipSuccess) ok = false; }
        if (hipSetDevice(m->dev[0]) != hipSuccess) ok = false;
      }
      ok = ok && hipMemcpyAsync(table, d_all, n * 100, hipMemcpyDeviceToHost, s0) == hipSuccess;
      if (hipStreamSynchronize(s0) != hipSuccess) ok = false;
      if (!ok) rc = ZKWG_RC_HIP_ERROR;
    }
    if (d_all) hipFree(d_all);
  }
  for (int i = 0; i < nd; ++i) {
    if (d_rows[i] || d_tab[i] || d_st[i]) { hipSetDevice(m->dev[i]); hipFree(d_rows[i]); hipFree(d_tab[i]); hipFree(d_st[i]); }
  }
  if (caller_dev >= 0) (void)hipSetDevice(caller_dev);
  return rc;
}

}  // extern "C"
