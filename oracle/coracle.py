"""ctypes front end of the fast C oracle (oracle/c/zkwg_oracle.c).  Test infrastructure only."""
import ctypes as C
import os
import subprocess

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "c")
_SO = os.path.join(_DIR, "libzkwg_oracle.so")
_lib = None
_native = None


def _bind(lib):
    lib.zkwg_oracle_calculate.restype = C.c_uint64
    lib.zkwg_oracle_calculate.argtypes = [C.c_uint32] * 4 + [C.c_uint64] + [C.c_void_p] * 9 + [
        C.c_void_p, C.c_uint64, C.c_void_p, C.c_int]
    lib.zkwg_oracle_set_masks.restype = None
    lib.zkwg_oracle_set_masks.argtypes = [C.c_void_p, C.c_void_p]
    lib.zkwg_oracle_set_decoded.restype = None
    lib.zkwg_oracle_set_decoded.argtypes = [C.c_void_p]
    lib.zkwg_oracle_time.restype = C.c_uint64
    lib.zkwg_oracle_time.argtypes = lib.zkwg_oracle_calculate.argtypes
    lib.zkwg_oracle_set_sums.restype = None
    lib.zkwg_oracle_set_sums.argtypes = [C.c_void_p]
    lib.zkwg_oracle_touch.restype = None
    lib.zkwg_oracle_touch.argtypes = [C.c_void_p, C.c_uint64, C.c_int]
    return lib


def load_native():
    """-march=native build made on THIS machine (bench.py cpu_baseline); (lib, build string).
    Falls back to the portable library when the compiler is missing or the build fails."""
    global _native
    if _native is None:
        so = os.path.join(_DIR, "libzkwg_oracle_native.so")
        try:
            # always rebuild: a file that travelled from another machine was tuned for that CPU
            if os.path.exists(so):
                os.unlink(so)
            subprocess.check_call(["make", "-C", _DIR, "-s", "native"], stdout=subprocess.DEVNULL,
                                  stderr=subprocess.DEVNULL)
            _native = (_bind(C.CDLL(so)), "gcc -O3 -march=native -fopenmp, built on this box")
        except (OSError, subprocess.CalledProcessError):
            _native = (load(), "gcc -O3 -fopenmp, portable x86-64 build (native build failed)")
    return _native


def load(build_if_missing=True):
    global _lib
    if _lib is None:
        src = os.path.join(_DIR, "zkwg_oracle.c")
        stale = not os.path.exists(_SO) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(_SO))
        if stale and build_if_missing:
            subprocess.check_call(["make", "-C", _DIR, "-s"])
        _lib = _bind(C.CDLL(_SO))
    return _lib


def _limbs(vals):
    return b"".join((int(v) % (1 << 128)).to_bytes(16, "little") for v in vals)


def calculate(main_kind, max_header, max_body, ignore_body, inputs, threads=1, want_witness=True):
    """inputs: list of CircuitInput-like dicts (ints / decimal strings).  Returns (list of witness
    bytes, status list, W)."""
    lib = load()
    n = len(inputs)
    hdr = hl = body = bl = pre = pub = sig = msg = bhi = None
    u32a = lambda xs: (C.c_uint32 * n)(*[int(x) & 0xFFFFFFFF for x in xs])
    if main_kind == 1:
        hdr = b"".join(bytes(int(b) for b in i["paddedIn"]) for i in inputs)
        hl = u32a(i["paddedInLength"] for i in inputs)
    elif main_kind == 2:
        pub = b"".join(_limbs(i["modulus"]) for i in inputs)
        sig = b"".join(_limbs(i["signature"]) for i in inputs)
        msg = b"".join(_limbs(i["message"]) for i in inputs)
    else:
        hdr = b"".join(bytes(int(b) for b in i["emailHeader"]) for i in inputs)
        hl = u32a(i["emailHeaderLength"] for i in inputs)
        pub = b"".join(_limbs(i["pubkey"]) for i in inputs)
        sig = b"".join(_limbs(i["signature"]) for i in inputs)
        if not ignore_body:
            body = b"".join(bytes(int(b) for b in i["emailBody"]) for i in inputs)
            bl = u32a(i["emailBodyLength"] for i in inputs)
            pre = b"".join(bytes(int(b) for b in i["precomputedSHA"]) for i in inputs)
            bhi = u32a(i["bodyHashIndex"] for i in inputs)
    hm = bm = None
    if main_kind == 0 and "headerMask" in inputs[0]:
        hm = b"".join(bytes(int(b) for b in i["headerMask"]) for i in inputs)
    if main_kind == 0 and "bodyMask" in inputs[0]:
        bm = b"".join(bytes(int(b) for b in i["bodyMask"]) for i in inputs)
    dec = None
    if main_kind == 0 and "decodedEmailBodyIn" in inputs[0]:
        dec = b"".join(bytes(int(b) for b in i["decodedEmailBodyIn"]) for i in inputs)
    lib.zkwg_oracle_set_masks(hm, bm)
    lib.zkwg_oracle_set_decoded(dec)
    args = (main_kind, max_header, max_body, ignore_body, n, hdr, hl, body, bl, pre, pub, sig, msg, bhi)
    W = lib.zkwg_oracle_calculate(*args, None, 0, None, 1)
    status = (C.c_int * n)()
    out = None
    if want_witness:
        out = (C.c_uint8 * (n * W * 32))()
    lib.zkwg_oracle_calculate(*args, out, W * 32, status, threads)
    wits = [C.string_at(C.addressof(out) + i * W * 32, W * 32) for i in range(n)] if want_witness else None
    lib.zkwg_oracle_set_masks(None, None)
    lib.zkwg_oracle_set_decoded(None)
    return wits, list(status), W


def run_fields(max_header, max_body, ignore_body, fields, n, threads=1, out=None, per_thread_out=False, lib=None,
               pretouch=False):
    """Time/run the C oracle on pre-marshalled field arrays (as produced by zkwg.synth.packed_batch).
    `out`: optional ctypes buffer of n * 32 * W bytes (or threads * 32 * W with per_thread_out=True:
    the cpu_baseline timing mode, every witness fully written into its thread's buffer).
    Returns (W, status list, seconds)."""
    import time
    lib = lib or load()
    u32a = lambda xs: (C.c_uint32 * n)(*xs)
    hdr, pub, sig = bytes(fields["header"]), bytes(fields["pubkey"]), bytes(fields["sig"])
    hl = u32a(fields["hlen"])
    body = bl = pre = bhi = None
    if not ignore_body:
        body, pre = bytes(fields["body"]), bytes(fields["pre"])
        bl, bhi = u32a(fields["blen"]), u32a(fields["bhi"])
    dec = bytes(fields["decoded"]) if fields.get("decoded") else None
    lib.zkwg_oracle_set_decoded(dec)
    args = (0, max_header, max_body, ignore_body, n, hdr, hl, body, bl, pre, pub, sig, None, bhi)
    W = lib.zkwg_oracle_calculate(0, max_header, max_body, ignore_body, 1, hdr, hl, body, bl, pre, pub, sig, None, bhi,
                                  None, 0, None, 1)
    status = (C.c_int * n)()
    if pretouch and per_thread_out and out is not None:
        lib.zkwg_oracle_touch(out, W * 32, threads)
    t0 = time.perf_counter()
    fn = lib.zkwg_oracle_time if per_thread_out else lib.zkwg_oracle_calculate
    fn(*args, out, (W * 32) if out is not None else 0, status, threads)
    dt = time.perf_counter() - t0
    lib.zkwg_oracle_set_decoded(None)
    return W, list(status), dt


def take_fields(fields, idxs, n_total):
    """Sub-batch of run_fields-style field arrays (emails `idxs` of a batch of n_total)."""
    out = {}
    for k, v in fields.items():
        if isinstance(v, list):
            out[k] = [v[i] for i in idxs] if v else v
        else:
            per = len(v) // n_total if n_total else 0
            out[k] = b"".join(bytes(v[i * per:(i + 1) * per]) for i in idxs) if per else bytes(v)
    return out


def checksums(max_header, max_body, ignore_body, fields, n, threads=1):
    """Per-email 64-bit checksum (sum_j word64[j] * (2 j + 1) mod 2^64) of the oracle's witnesses,
    computed without keeping the witnesses (one scratch witness per thread).  -> (W, status, sums)."""
    lib = load()
    W, _, _ = run_fields(max_header, max_body, ignore_body, take_fields(fields, [0], n), 1)
    buf = (C.c_uint8 * (threads * 32 * W))()
    sums = (C.c_uint64 * n)()
    lib.zkwg_oracle_set_sums(sums)
    try:
        W, st, _ = run_fields(max_header, max_body, ignore_body, fields, n, threads=threads, out=buf, per_thread_out=True)
    finally:
        lib.zkwg_oracle_set_sums(None)
    return W, st, list(sums)
