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
