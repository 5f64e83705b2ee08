// BodyHashRegex DFA scan (zkwg v1 circuit, see zkwg_layout.h zk_walk_bh_regex).
//
// [EXT] @zk-email/zk-regex-circom 2.3.2 `body_hash_regex.circom` is instantiated at
// packages/circuits/email-verifier.circom:126 with the regex
//     (\r\n|^)dkim-signature:([a-z]+=[^;]+; )+bh=[a-zA-Z0-9+/=]+;
// (zk-regex feeds byte 255 in front of the message to stand for `^`).  The tables come from
// tools/gen_bh_dfa.py (minimal DFA of the anchored regex).  Circuit semantics: state 0 is permanently
// active; a transition out of state 0 fires only when no other state continues (from_zero_enabled).
//
// One sequential pass over in[] = [255, header...] yields st[j] (the active non-zero state before in[j],
// 0 = none), a backward pass yields live[j] (the thread in st[j] reaches the accept state without
// restarting), from which reveal0 and the quadratic helper signals follow.
#pragma once
#include "zkwg_fr.h"
#include "zkwg_bh_dfa.h"

// hdr[0..N): header bytes.  delta: ZK_DFA_DELTA (host) or its device copy.
// st[0..N+3), live[0..N+3): scratch/outputs.  own: live_c1[nb], live_t[nb], prev_states0[NP][N], is_reveal0[N]
// (nb = N + 1).  rev[0..N): reveal0.  Returns the number of positions in the accept state.
ZK_HD u32 zk_bh_dfa_scan(const u8* hdr, u32 N, const unsigned char (*delta)[256], u8* st, u8* live, u32* own, u32* rev) {
  const u32 nb = N + 1;
  u32 cur = 0, acc_count = 0;
  st[0] = 0;
  for (u32 i = 0; i < nb; ++i) {
    const u32 b = i == 0 ? 255u : hdr[i - 1];
    u32 nx = cur ? delta[cur][b] : 255u;
    if (nx == 255u) { nx = delta[0][b]; if (nx == 255u) nx = 0; }
    cur = nx;
    st[i + 1] = (u8)cur;
    acc_count += (cur == ZK_DFA_ACCEPT);
  }
  st[nb + 1] = 0;
  live[nb + 1] = 0;
  for (u32 j = nb; j >= 1; --j) {
    u32 c1 = 0;
    if (j < nb) {  // from_zero_enabled[j] = no transition out of a non-zero state at position j
      const u32 sj = st[j];
      const u32 fze = sj ? (delta[sj][hdr[j - 1]] == 255u) : 1u;
      c1 = live[j + 1] & (1u - fze);
    }
    const u32 acc = st[j] == ZK_DFA_ACCEPT;
    const u32 tt = (1u - acc) & c1;
    own[j - 1] = c1;
    own[nb + j - 1] = tt;
    live[j] = (u8)(acc | tt);
  }
  for (u32 i = 0; i < N; ++i) {
    u32 sub = 0;
    for (u32 k = 0; k < ZK_DFA_NPUBLIC; ++k) {
      const u32 pv = (st[i + 1] == ZK_DFA_PUBLIC[k][0] && st[i + 2] == ZK_DFA_PUBLIC[k][1]);
      own[2 * nb + k * N + i] = pv;
      sub |= pv;
    }
    const u32 ir = sub & live[i + 2];
    own[2 * nb + ZK_DFA_NPUBLIC * N + i] = ir;
    rev[i] = ir ? hdr[i] : 0;
  }
  return acc_count;
}
