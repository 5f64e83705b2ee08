"""Test-only helpers: host build of the product's schedule builder + wave-collective RSA
core (tests/native/hosttest.cpp), and a Python restatement of zk_expand's segment semantics
so that CPU tests can turn a compact image into a witness without a GPU."""
import ctypes as C
import os
import subprocess

from conftest import ROOT

P = 21888242871839275222246405745257275088548364400416034343698204186575808495617
_SO = os.path.join(ROOT, "tests", "native", "libzkwg_hosttest.so")
_SRC = os.path.join(ROOT, "tests", "native", "hosttest.cpp")
_CSRC = os.path.join(ROOT, "zk-email-verify_amd", "csrc")


class Seg(C.Structure):
    _fields_ = [("slot", C.c_uint64), ("nslots", C.c_uint32), ("type", C.c_uint32),
                ("src", C.c_uint32), ("a", C.c_uint32), ("b", C.c_uint32), ("c", C.c_uint32),
                ("r0", C.c_uint32), ("pad", C.c_uint32)]


_WSO = os.path.join(ROOT, "tests", "native", "libzkwg_wavetest.so")


def load_wave():
    """tests/native/wavetest.cpp: csrc/zkwg_rsa_wave.h -- the RSA path the device runs -- on a 64-fiber wavefront (wavesim.h)"""
    src = [os.path.join(ROOT, "tests", "native", f) for f in ("wavetest.cpp", "wavesim.h")]
    deps = src + [os.path.join(_CSRC, f) for f in os.listdir(_CSRC) if f.endswith(".h")]
    if not os.path.exists(_WSO) or any(os.path.getmtime(d) > os.path.getmtime(_WSO) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-I", _CSRC, "-I", os.path.join(ROOT, "tests", "native"), src[0], "-o", _WSO])
    lib = C.CDLL(_WSO)
    lib.wt_run_rsa.restype = C.c_int
    lib.wt_run_rsa.argtypes = [C.c_void_p] * 6 + [C.POINTER(C.c_uint64)]
    lib.wt_run_rslb_merge.restype = C.c_int
    lib.wt_run_rslb_merge.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64)]
    return lib


def load():
    deps = [_SRC] + [os.path.join(_CSRC, f) for f in os.listdir(_CSRC) if f.endswith(".h")]
    if not os.path.exists(_SO) or any(os.path.getmtime(d) > os.path.getmtime(_SO) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-I", _CSRC, _SRC, "-o", _SO])
    lib = C.CDLL(_SO)
    lib.ht_create.restype = C.c_void_p
    lib.ht_create.argtypes = [C.c_void_p]
    lib.ht_create_sym.restype = C.c_void_p
    lib.ht_create_sym.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64]
    lib.ht_destroy.argtypes = [C.c_void_p]
    lib.ht_W.restype = C.c_uint64
    lib.ht_W.argtypes = [C.c_void_p]
    lib.ht_segs.restype = C.POINTER(Seg)
    lib.ht_segs.argtypes = [C.c_void_p]
    for f in ("ht_nsegs", "ht_img_bits", "ht_img_small", "ht_img_fr", "ht_in_stride", "ht_inv_half", "ht_m_one"):
        getattr(lib, f).restype = C.c_uint32
        getattr(lib, f).argtypes = [C.c_void_p]
    lib.ht_in_off.restype = C.c_uint32
    lib.ht_in_off.argtypes = [C.c_void_p, C.c_int]
    lib.ht_run_rsa.restype = C.c_int
    lib.ht_run_rsa.argtypes = [C.c_void_p] * 6
    lib.ht_run_fpmul.restype = C.c_int
    lib.ht_run_fpmul.argtypes = [C.c_void_p] * 5
    lib.ht_poseidon.argtypes = [C.c_void_p] * 3
    lib.ht_poseidon_sparse.restype = C.c_int
    lib.ht_poseidon_sparse.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.ht_p29_violations.restype = C.c_ulonglong
    lib.ht_poseidon29.restype = C.c_int
    lib.ht_poseidon29.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.ht_r1cs_first_bad.restype = C.c_longlong
    lib.ht_r1cs_first_bad.argtypes = [C.c_char_p, C.c_uint64, C.c_char_p]
    lib.ht_net_load.restype = C.c_void_p
    lib.ht_net_load.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_uint32]
    lib.ht_net_destroy.argtypes = [C.c_void_p]
    lib.ht_net_self_check.restype = C.c_int
    lib.ht_net_self_check.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_uint32]
    lib.ht_net_check_error.restype = C.c_char_p
    lib.ht_net_check_error.argtypes = [C.c_void_p]
    lib.ht_net_damage_chain.argtypes = [C.c_void_p]
    lib.ht_net_error.restype = C.c_char_p
    lib.ht_net_error.argtypes = [C.c_void_p]
    lib.ht_net_names.restype = C.c_char_p
    lib.ht_net_names.argtypes = [C.c_void_p]
    for f in ("ht_net_kept", "ht_net_inv_need"):
        getattr(lib, f).restype = C.c_uint32
        getattr(lib, f).argtypes = [C.c_void_p]
    lib.ht_net_eval.restype = C.c_int
    lib.ht_net_eval.argtypes = [C.c_void_p] * 5
    lib.ht_regex_scan.restype = C.c_uint32
    lib.ht_regex_scan.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    return lib


def synthetic_sym(symbols, n_public, seed, drop_frac=0.01, max_block=400):
    """A `.sym` file a compiler could have written for the same circuit, made from the kept-v1 table
    `symbols` = [(slot, name)]: main I/O stays in front, the other signals are cut into blocks that are
    shuffled, a few signals are marked eliminated (witness index -1), label indices are unrelated.
    Returns (text, dst) with dst[kept-v1 slot] = witness index or None."""
    import random
    rng = random.Random(seed)
    W = len(symbols)
    rest = list(range(n_public + 1, W))
    blocks = []
    i = 0
    while i < len(rest):
        n = rng.randrange(1, max_block)
        blocks.append(rest[i:i + n])
        i += n
    rng.shuffle(blocks)
    dst = [None] * W
    w = 0
    for s in range(n_public + 1):
        dst[s] = w
        w += 1
    for b in blocks:
        for s in b:
            if rng.random() < drop_frac:
                continue
            dst[s] = w
            w += 1
    lines = [f"{7 * s + 3},{-1 if dst[s] is None else dst[s]},{s % 97},{symbols[s][1]}" for s in range(1, W)]
    rng.shuffle(lines)
    return "\n".join(lines) + "\n", dst


def apply_sym(witness, dst):
    out = [None] * (max(d for d in dst if d is not None) + 1)
    for s, d in enumerate(dst):
        if d is not None:
            out[d] = witness[s]
    return out


def expand(lib, h, rec, bits, small, frv):
    """Python restatement of zk_expand (zkwg_kernels_expand.hip): image -> witness ints."""
    W = lib.ht_W(h)
    segs = lib.ht_segs(h)
    half = lib.ht_inv_half(h)
    out = [None] * W

    def i32(x):
        return x - (1 << 32) if x >> 31 else x

    def inv(d):
        return 0 if d == 0 else pow(d % P, P - 2, P)

    def fr(i):
        return int.from_bytes(bytes(frv[32 * i:32 * i + 32]), "little")

    for si in range(lib.ht_nsegs(h)):
        s = segs[si]
        for rr in range(s.nslots):
            r = s.r0 + rr
            t = s.type
            if t == 0:
                v = small[s.src + r]
            elif t == 1:
                v = fr(s.src + r)
            elif t == 2:
                g, bit = divmod(r, s.a)
                v = (bits[s.src + g * s.b + (bit >> 6)] >> (bit & 63)) & 1
            elif t in (3, 4, 5):
                per, nw, last = {3: (162, 5, 4), 4: (131, 4, 3), 5: (161, 5, 4)}[t]
                i, q = divmod(r, per)
                sub = min(q >> 5, last)
                v = (bits[s.src + i * nw + sub] >> (q - 32 * sub)) & 1
            elif t == 6:
                d = i32(small[s.src + (r >> 1)])
                v = (1 if d == 0 else 0) if r % 2 == 0 else inv(max(-half, min(half, d)))
            elif t == 7:
                NB = s.a
                idx = i32(small[s.src])
                k, q = divmod(r, 3 * NB)
                if q < NB:
                    v = ((small[s.b + (k >> 5)] >> (31 - (k & 31))) & 1) if q == idx else 0
                else:
                    tt = q - NB
                    j = tt >> 1
                    v = (1 if j == idx else 0) if tt % 2 == 0 else inv(max(-half, min(half, idx - j)))
            elif t == 8:
                v = rec[s.src + r]
            elif t == 9:
                v = (rec[s.src + (r >> 3)] >> (r & 7)) & 1
            elif t == 10:
                v = int.from_bytes(bytes(rec[s.src + 16 * r:s.src + 16 * r + 16]), "little")
            elif t == 11:
                base = i32(small[s.src]) + (1 << s.a)
                i, bit = divmod(r, s.a + 1)
                v = ((base - i) >> bit) & 1
            elif t == 12:
                bl, N = s.a, s.c
                per = 6 + bl + 1
                start = i32(small[s.src])
                if r < per - 2:
                    i, q = 0, (r if r < 4 else r + 2)
                else:
                    i, q = divmod(r - (per - 2), per)
                    i += 1
                if q < 6:
                    d = (start - i) if q < 2 else (i32(small[s.b + i]) if q < 4 else i32(small[s.b + i - 1]))
                    v = (1 if d == 0 else 0) if q % 2 == 0 else inv(max(-half, min(half, d)))
                else:
                    v = ((start + 43 + (1 << bl) - i) >> (q - 6)) & 1
            elif t == 13:
                N = s.a
                j, i = divmod(r, N)
                sh = small[s.src] & ((2 << j) - 1)
                v = small[s.b + (i + sh) % N]
            elif t in (14, 15):
                per = 68 if t == 15 else 6
                g, q = divmod(r, per)
                ch = small[s.src + g]
                rAZ, raz, r09 = int(65 <= ch <= 90), int(97 <= ch <= 122), int(48 <= ch <= 57)
                sAZ = rAZ * (ch - 65); saz = sAZ + raz * (ch - 71); s09 = saz + r09 * (ch + 4)
                spl = s09 + (ch == 43) * (ch + 19); ssl = spl + (ch == 47) * (ch + 16)
                if t == 14:
                    v = (ssl >> q) & 1
                elif q < 8:
                    v = [rAZ, sAZ, raz, saz, r09, s09, spl, ssl][q]
                elif q < 62:
                    k, bit = divmod(q - 8, 9)
                    v = ([ch + 256 - 91, 64 + 256 - ch, ch + 256 - 123, 96 + 256 - ch, ch + 256 - 58, 47 + 256 - ch][k] >> bit) & 1
                else:
                    k = (q - 62) >> 1
                    d = ch - [43, 47, 61][k]
                    v = (1 if d == 0 else 0) if (q - 62) % 2 == 0 else inv(max(-half, min(half, d)))
            else:
                raise ValueError(t)
            assert out[s.slot + rr] is None
            out[s.slot + rr] = v
    assert all(x is not None for x in out)
    return out


class LoadedRegex:
    """A regex template loaded by the product's front end (zkwg_circom.h) and evaluated on the host with
    the code zk_net_eval runs on the device (zkwg_net_core.h)."""

    def __init__(self, path, n, include_dirs=(), template="BodyHashRegex"):
        self.lib = load()
        self.h = self.lib.ht_net_load(str(path).encode(), ":".join(str(d) for d in include_dirs).encode(), template.encode(), n)
        err = self.lib.ht_net_error(self.h)
        if err is not None:
            raise ValueError(err.decode())
        self.n = n
        self.kept = self.lib.ht_net_kept(self.h)
        self.names = self.lib.ht_net_names(self.h).decode().split("\n")[:-1]
        assert len(self.names) == self.kept

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.ht_net_destroy(self.h)
            self.h = None

    def self_check(self, path, include_dirs=(), template="BodyHashRegex", damage=False):
        """the loader's own check of its scan tables against the plain gate list (zkwg_net_host.h zkc::self_check, what
        zkwg_circuit_create_regex runs once per handle) -> None, or the error text; damage=True zeroes the forward chain's table first"""
        if damage:
            self.lib.ht_net_damage_chain(C.c_void_p(self.h))
        if self.lib.ht_net_self_check(C.c_void_p(self.h), str(path).encode(), ":".join(str(d) for d in include_dirs).encode(), template.encode(), self.n):
            return None
        return self.lib.ht_net_check_error(C.c_void_p(self.h)).decode()

    def chain_info(self):
        """-> (positions served from the forward chain tables, from the backward ones, steps left in the gate list)"""
        out = (C.c_uint32 * 4)()
        self.lib.ht_net_chain_info(C.c_void_p(self.h), out)
        return int(out[0]), int(out[1]), int(out[2])

    def evaluate(self, msg):
        """-> (ok, {name: field element}, match, reveal list)"""
        msg = bytes(msg) + bytes(self.n - len(msg))
        words = (C.c_uint32 * self.kept)()
        match = C.c_uint32(0xffffffff)
        reveal = (C.c_uint32 * self.n)()
        ok = self.lib.ht_net_eval(self.h, msg, words, C.byref(match), reveal)
        assert ok >= 0, "the runs of the region do not tile it"
        vals = {}
        for nm, w in zip(self.names, words):
            d = w & 0x7fffffff
            if d & 0x40000000:
                d -= 1 << 31
            vals[nm] = pow(d, -1, P) if (w >> 31) and d else d % P
        return bool(ok), vals, match.value, list(reveal)
