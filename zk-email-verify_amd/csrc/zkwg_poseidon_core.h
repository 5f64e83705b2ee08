// PoseidonLarge(121,17) -> Poseidon(9) (packages/circuits/utils/hash.circom:15-39,
// circomlib poseidon.circom [EXT]): x^5 S-box, t = 10, 8 full + 60 partial rounds.
// Wave-collective: the 10 x 10 MDS product is spread over the lanes (lane = matrix entry).
// Emits the 420 quadratic S-box signals (Sigma.out, .in2, .in4) and the hash.
#pragma once
#include "zkwg_rsa_core.h"

#define ZK_POS_T 10
#define ZK_POS_RF 8
#define ZK_POS_RP 60

struct ZkPosLds {
  Fr st[ZK_POS_T];             // state (Montgomery form)
  Fr prod[ZK_POS_T * ZK_POS_T];
};

// pubkey limbs (17 x 121 bits) -> frv[f_pos .. f_pos+420), returns hash (standard form) in *hash
ZK_DEV inline void zk_poseidon_large(ZkPosLds& S, const u64 (*limb)[2], const Fr* C, const Fr* M,
                                     Fr* frv_pos, Fr* hash) {
  // poseidonInput[i] = in[2i] + 2^121 * in[2i+1] (i < 8), in[16] (i = 8); state = [0, inputs]
  ZK_PAR_FOR(j, ZK_POS_T) {
    Fr v = fr_zero();
    if (j >= 1) {
      u32 i = j - 1;
      v.l[0] = limb[2 * i][0];
      v.l[1] = limb[2 * i][1];
      if (i < 8) {  // + hi << 121
        u64 h0 = limb[2 * i + 1][0], h1 = limb[2 * i + 1][1];
        v.l[1] |= h0 << 57;
        v.l[2] = (h0 >> 7) | (h1 << 57);
        v.l[3] = h1 >> 7;
      }
    }
    S.st[j] = fr_to_mont(v);
  }
  ZK_SYNC();
  u32 fr_round = 0;
  for (u32 r = 0; r < ZK_POS_RF + ZK_POS_RP; ++r) {
    const bool full = r < ZK_POS_RF / 2 || r >= ZK_POS_RF / 2 + ZK_POS_RP;
    ZK_PAR_FOR(j, ZK_POS_T) {
      Fr x = fr_add(S.st[j], C[r * ZK_POS_T + j]);
      if (full || j == 0) {
        Fr x2 = fr_mont_mul(x, x);
        Fr x4 = fr_mont_mul(x2, x2);
        Fr x5 = fr_mont_mul(x4, x);
        u32 base = full ? (fr_round * ZK_POS_T + j) * 3 : (ZK_POS_RF * ZK_POS_T + (r - ZK_POS_RF / 2)) * 3;
        frv_pos[base] = x5; frv_pos[base + 1] = x2; frv_pos[base + 2] = x4;  // Montgomery form for now
        x = x5;
      }
      S.st[j] = x;
    }
    if (full) ++fr_round;
    ZK_SYNC();
    ZK_PAR_FOR(e, ZK_POS_T * ZK_POS_T) {
      u32 j = e % ZK_POS_T;
      S.prod[e] = fr_mont_mul(M[e], S.st[j]);  // M[i][j] * s[j], e = i*10 + j
    }
    ZK_SYNC();
    ZK_PAR_FOR(i, ZK_POS_T) {
      Fr acc = S.prod[i * ZK_POS_T];
      for (u32 j = 1; j < ZK_POS_T; ++j) acc = fr_add(acc, S.prod[i * ZK_POS_T + j]);
      S.st[i] = acc;
    }
    ZK_SYNC();
  }
  ZK_PAR_FOR(e, 420) { frv_pos[e] = fr_from_mont(frv_pos[e]); }
  ZK_SEQ { *hash = fr_from_mont(S.st[0]); }
  ZK_SYNC();
}
