// Host build of the RSA path the DEVICE runs (csrc/zkwg_rsa_wave.h: ballot carry look-ahead, readlane / shuffle
// exchanges, lane-parallel Knuth D, safegcd) on a 64-fiber wavefront (tests/native/wavesim.h).  A library of its own:
// zkwg_rsa_core.h is compiled here in its lane-parallel mode, in libzkwg_hosttest.so in its phase-sequential one.
// Test infrastructure, not a product fallback.
#define ZKWG_WAVESIM 1
#include "wavesim.h"
#include <string.h>
#include "../../include/zkwg.h"
#include "zkwg_rsa_wave.h"
#include "zkwg_layout.h"
#include "zkwg_build.h"

extern "C" {
// the RSA block of one email exactly as the kernel zk_rsa calls it; digest = 8 state words or NULL.
// returns S.ok (1 = every assertion holds); *exchanges = cross-lane operations the wavefront executed
int wt_run_rsa(const zkwg_config* cfg, const uint8_t* rec, const uint32_t* digest, uint64_t* bits, uint32_t* small, void* frv,
               uint64_t* exchanges) {
  ZkSched s;
  std::vector<ZkSeg> segs;
  if (!build_sched(*cfg, s, segs)) return -1;
  ZkRsaLds* S = new ZkRsaLds();
  const uint64_t n = wavesim::run([&] { zkw_rsa_email(*S, s.rsa, rec, digest, bits, small, (Fr*)frv); });
  if (exchanges) *exchanges = n;
  const int ok = (int)S->ok;
  delete S;
  return ok;
}
}

// ---- removeSoftLineBreaks: the Poseidon(2) merge chain exactly as the kernel zk_rslb_merge runs it (csrc/zkwg_rslb_wave.h:
// 4 lanes per email, Montgomery-form state, one product per step for all lanes), on the simulated wavefront.
#include "zkwg_rslb_wave.h"
extern "C" {
// frv: n_emails x img_fr field elements (standard form, 32 bytes each) with the chunk digests at f_rs_chunk .. + rs_nch;
// on return the merge permutations' S-box signals sit at f_rs_hash + zk_rs_chunk_off(c) + 612 (c >= 1) and r at f_rs_chunk.
int wt_run_rslb_merge(uint32_t n_emails, uint32_t rs_nch, void* frv, uint32_t img_fr, uint32_t f_rs_chunk, uint32_t f_rs_hash, uint64_t* exchanges) {
  std::vector<Fr> C, M, t2;
  build_poseidon_constants(3, 8, 57, C, M);
  if (!zk_build_poseidon_sparse(3, 57, C, M, t2)) return -1;
  const uint32_t per = 64u / ZK_RS_MERGE_LANES;
  uint64_t n = 0;
  ZkRsMergeLds* S = new ZkRsMergeLds();
  for (uint32_t block = 0; block * per < n_emails; ++block)
    n += wavesim::run([&] { zk_rslb_merge_wave(*S, t2.data(), (Fr*)frv, block, n_emails, img_fr, rs_nch, f_rs_chunk, f_rs_hash); });
  delete S;
  if (exchanges) *exchanges = n;
  return 0;
}
}
