#!/usr/bin/env python3
"""Generates the DFA tables of the body-hash regex used by EmailVerifier
(packages/circuits/email-verifier.circom:126, [EXT] @zk-email/zk-regex-circom body_hash_regex):

    (\\r\\n|^)dkim-signature:([a-z]+=[^;]+; )+bh=[a-zA-Z0-9+/=]+;        public part: the bh value

zk-regex feeds byte 255 for `^` and compiles the *anchored* regex to a minimal DFA; the generated
circuit keeps state 0 permanently active and lets a transition out of state 0 fire only when no other
state is active.  The real generated circuit is not available offline, so zkwg defines its own circuit
of the same style over the tables produced here ("zkwg BodyHashRegex v1"):

  outputs: zk-email-verify_amd/data/bh_dfa.json        (read by the oracle and the tests)
           zk-email-verify_amd/csrc/zkwg_bh_dfa.h      (compiled into libzkwg.so)

Run:  python tools/gen_bh_dfa.py
"""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PAT = b"dkim-signature:"
AZ = set(range(ord("a"), ord("z") + 1))
B64 = AZ | set(range(ord("A"), ord("Z") + 1)) | set(range(ord("0"), ord("9") + 1)) | {ord("+"), ord("/"), ord("=")}

# ---- anchored epsilon-free NFA: state -> [(byteset, next)]
START, CR, LS = "S", "CR", "LS"
nfa = {START: [({13}, CR), ({255}, LS)], CR: [({10}, LS)]}
prev = LS
for k, ch in enumerate(PAT):
    nxt = f"P{k + 1}"
    nfa.setdefault(prev, []).append(({ch}, nxt))
    prev = nxt
AFTER = prev                      # after ':'
NAME, EQ, VAL, SC, SP, B1, B2, B3, B4, ACC = "NAME", "EQ", "VAL", "SC", "SP", "B1", "B2", "B3", "B4", "ACC"
NOT_SC = set(range(256)) - {ord(";")}
nfa.setdefault(AFTER, []).append((AZ, NAME))
nfa[NAME] = [(AZ, NAME), ({ord("=")}, EQ)]
nfa[EQ] = [(NOT_SC, VAL)]
nfa[VAL] = [(NOT_SC, VAL), ({ord(";")}, SC)]
nfa[SC] = [({ord(" ")}, SP)]
nfa[SP] = [(AZ, NAME), ({ord("b")}, B1)]
nfa[B1] = [({ord("h")}, B2)]
nfa[B2] = [({ord("=")}, B3)]
nfa[B3] = [(B64, B4)]
nfa[B4] = [(B64, B4), ({ord(";")}, ACC)]
nfa[ACC] = []
PUBLIC_NFA = {(B3, B4), (B4, B4)}  # transitions that consume the public (revealed) part

# ---- subset construction
def step(S, b):
    out = set()
    for s in S:
        for bs, n in nfa.get(s, []):
            if b in bs:
                out.add(n)
    return frozenset(out)

start = frozenset([START])
states = [start]
index = {start: 0}
delta = {}
work = [start]
while work:
    S = work.pop()
    for b in range(256):
        T = step(S, b)
        if not T:
            continue
        if T not in index:
            index[T] = len(states)
            states.append(T)
            work.append(T)
        delta[(index[S], b)] = index[T]
n = len(states)
accept = [i for i, S in enumerate(states) if ACC in S]
# public DFA transitions: those realising a public NFA transition
public = set()
for (i, b), j in delta.items():
    for s in states[i]:
        for bs, t in nfa.get(s, []):
            if b in bs and (s, t) in PUBLIC_NFA and t in states[j]:
                public.add((i, j))

# ---- minimisation (partition refinement); keep the public-transition structure distinguishable
def signature(i, part):
    return tuple(part.get(delta.get((i, b), -1), -1) for b in range(256))

part = {i: (1 if i in accept else 0) for i in range(n)}
part[-1] = -1
while True:
    sigs = {}
    newp = {}
    for i in range(n):
        key = (part[i], signature(i, part))
        newp[i] = sigs.setdefault(key, len(sigs))
    newp[-1] = -1
    if len(set(newp[i] for i in range(n))) == len(set(part[i] for i in range(n))):
        part = newp
        break
    part = newp
# renumber: block of the start state first, then BFS order for determinism
blocks = {}
order = []
seen = set()
queue = [0]
while queue:
    i = queue.pop(0)
    blk = part[i]
    if blk in seen:
        continue
    seen.add(blk)
    order.append(blk)
    for b in range(256):
        j = delta.get((i, b))
        if j is not None and part[j] not in seen:
            queue.append(j)
ren = {blk: k for k, blk in enumerate(order)}
S = len(order)
D = [[255] * 256 for _ in range(S)]          # 255 = dead
for (i, b), j in delta.items():
    D[ren[part[i]]][b] = ren[part[j]]
ACCEPT = sorted({ren[part[i]] for i in accept})
PUBLIC = sorted({(ren[part[i]], ren[part[j]]) for (i, j) in public})
assert len(ACCEPT) == 1 and S < 255

# ---- group each state's outgoing bytes into (target -> byte set), express byte sets as classes
def ranges(bs):
    out, xs = [], sorted(bs)
    lo = prev = xs[0]
    for x in xs[1:]:
        if x != prev + 1:
            out.append((lo, prev)); lo = x
        prev = x
    out.append((lo, prev))
    return out

classes = []        # each: sorted list of (lo, hi)
def class_id(bs):
    r = ranges(bs)
    if r not in classes:
        classes.append(r)
    return classes.index(r)

transitions = []    # (from, to, class)
for s in range(S):
    by_target = {}
    for b in range(256):
        if D[s][b] != 255:
            by_target.setdefault(D[s][b], set()).add(b)
    for t in sorted(by_target):
        transitions.append((s, t, class_id(by_target[t])))

# primitive tests: eq (single byte) and ranges (lo<hi); big complements are expressed as NOT(small class)
prims = []          # ("eq", ch) | ("range", lo, hi)
cls_def = []        # per class: {"neg": bool, "members": [prim indices]}
for r in classes:
    size = sum(hi - lo + 1 for lo, hi in r)
    neg = size > 128
    rr = ranges(set(range(256)) - {x for lo, hi in r for x in range(lo, hi + 1)}) if neg else r
    members = []
    for lo, hi in rr:
        p = ("eq", lo) if lo == hi else ("range", lo, hi)
        if p not in prims:
            prims.append(p)
        members.append(prims.index(p))
    cls_def.append({"neg": neg, "members": members})

table = {
    "regex": "(\\r\\n|^)dkim-signature:([a-z]+=[^;]+; )+bh=[a-zA-Z0-9+/=]+;",
    "n_states": S, "accept": ACCEPT[0], "delta": D, "public": PUBLIC,
    "prims": [list(p) for p in prims], "classes": cls_def, "transitions": [list(t) for t in transitions],
}
os.makedirs(os.path.join(ROOT, "zk-email-verify_amd", "data"), exist_ok=True)
json.dump(table, open(os.path.join(ROOT, "zk-email-verify_amd", "data", "bh_dfa.json"), "w"))

# ---- C header
eqs = [p for p in prims if p[0] == "eq"]
rngs = [p for p in prims if p[0] == "range"]
h = []
h.append("// GENERATED by tools/gen_bh_dfa.py -- DFA tables of the body-hash regex (do not edit).")
h.append("#pragma once")
h.append(f"#define ZK_DFA_STATES {S}")
h.append(f"#define ZK_DFA_ACCEPT {ACCEPT[0]}")
h.append(f"#define ZK_DFA_NPRIM {len(prims)}")
h.append(f"#define ZK_DFA_NCLASS {len(cls_def)}")
h.append(f"#define ZK_DFA_NTRANS {len(transitions)}")
h.append(f"#define ZK_DFA_NPUBLIC {len(PUBLIC)}")
h.append("// prim: {kind (0 eq, 1 range), lo, hi}")
h.append("static const unsigned char ZK_DFA_PRIM[ZK_DFA_NPRIM][3] = {" + ", ".join(
    "{%d,%d,%d}" % ((0, p[1], p[1]) if p[0] == "eq" else (1, p[1], p[2])) for p in prims) + "};")
h.append("// class: {neg, nmembers, members...(max 8)}")
mx = max(len(c["members"]) for c in cls_def)
h.append(f"#define ZK_DFA_MAXMEM {mx}")
h.append("static const unsigned char ZK_DFA_CLASS[ZK_DFA_NCLASS][2 + ZK_DFA_MAXMEM] = {" + ", ".join(
    "{" + ",".join(str(x) for x in [int(c["neg"]), len(c["members"])] + c["members"] + [0] * (mx - len(c["members"]))) + "}"
    for c in cls_def) + "};")
h.append("// transition: {from, to, class}")
h.append("static const unsigned char ZK_DFA_TRANS[ZK_DFA_NTRANS][3] = {" + ", ".join("{%d,%d,%d}" % t for t in transitions) + "};")
h.append("static const unsigned char ZK_DFA_PUBLIC[ZK_DFA_NPUBLIC][2] = {" + ", ".join("{%d,%d}" % t for t in PUBLIC) + "};")
h.append("// delta[state][byte] -> next state, 255 = dead")
h.append("static const unsigned char ZK_DFA_DELTA[ZK_DFA_STATES][256] = {")
for s in range(S):
    h.append("  {" + ",".join(str(x) for x in D[s]) + "},")
h.append("};")
# per-byte truth masks: bit p = primitive test p holds, bit c = class c holds (negation applied)
def prim_true(p, b):
    return b == p[1] if p[0] == "eq" else p[1] <= b <= p[2]
primmask = [sum(1 << k for k, p in enumerate(prims) if prim_true(p, b)) for b in range(256)]
clsmask = []
for b in range(256):
    m = 0
    for k, cdef in enumerate(cls_def):
        v = any(prim_true(prims[q], b) for q in cdef["members"])
        if v != cdef["neg"]:
            m |= 1 << k
    clsmask.append(m)
assert len(prims) <= 32 and len(cls_def) <= 32
h.append("static const unsigned int ZK_DFA_PRIMMASK[256] = {" + ",".join("0x%xu" % x for x in primmask) + "};")
h.append("static const unsigned int ZK_DFA_CLSMASK[256] = {" + ",".join("0x%xu" % x for x in clsmask) + "};")
h.append("// members of each class as a bit mask over the primitive tests")
h.append("static const unsigned int ZK_DFA_CLASS_MEMBERS[ZK_DFA_NCLASS] = {" + ",".join(
    "0x%xu" % sum(1 << q for q in c["members"]) for c in cls_def) + "};")
text = "\n".join(h) + "\n"
open(os.path.join(ROOT, "zk-email-verify_amd", "csrc", "zkwg_bh_dfa.h"), "w").write(text)
# the oracle keeps its own copy of the generated tables (test infrastructure stays self-contained)
open(os.path.join(ROOT, "oracle", "c", "bh_dfa_tables.h"), "w").write(text)
print(f"states={S} accept={ACCEPT[0]} prims={len(prims)} (eq {len(eqs)}, range {len(rngs)}) classes={len(cls_def)} "
      f"transitions={len(transitions)} public={PUBLIC}")
