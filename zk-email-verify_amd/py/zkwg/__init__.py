"""zkwg -- Python host side of the MI355X-native batched witness generator.

Mirrors the interface the reference uses for this path:

* circom_runtime's `WitnessCalculator` (`calculateWitness`, `calculateBinWitness`,
  `calculateWTNSBin`) as called through `circom_tester.wasm` in
  packages/circuits/tests/email-verifier.test.ts:21-44 and through
  `snarkjs.groth16.fullProve` in packages/helpers/src/chunked-zkey.ts:80-84;
* the `CircuitInput` object of packages/helpers/src/input-generators.ts:6-18.

Everything below the `WitnessCalculator` goes through the C-ABI of libzkwg.so; there
is no CPU fallback (a missing library or GPU raises).
"""
import ctypes as C

from . import _lib
from ._lib import Config, MAIN_EMAIL_VERIFIER, MAIN_SHA256_BYTES, MAIN_RSA_VERIFIER, MAIN_FP_MUL

FIELD_MODULUS = 21888242871839275222246405745257275088548364400416034343698204186575808495617


class ZkwgError(RuntimeError):
    pass


def _check(rc):
    if rc != 0:
        raise ZkwgError(_lib.load().zkwg_strerror(rc).decode() + f" (rc={rc})")


def _limbs_bytes(vals, n_limbs=17):
    """list of ints (each < 2^128) -> n_limbs x 16-byte little-endian."""
    vals = list(vals)
    if len(vals) > n_limbs:
        raise ZkwgError("Too many values for input signal")
    if len(vals) < n_limbs:
        raise ZkwgError("Not enough values for input signal")
    out = bytearray()
    for v in vals:
        v = int(v) % FIELD_MODULUS
        if v >> 128:
            raise ZkwgError("limb does not fit the packed 128-bit input path")
        out += v.to_bytes(16, "little")
    return bytes(out)


# CircuitInput signal name -> packed record field (include/zkwg.h enum zkwg_input_field)
_FIELD_OF = {"emailHeader": 0, "paddedIn": 0, "emailBody": 1, "precomputedSHA": 2, "pubkey": 3, "modulus": 3, "a": 3,
             "signature": 4, "b": 4, "message": 5, "p": 5, "emailHeaderLength": 6, "paddedInLength": 6, "emailBodyLength": 7,
             "bodyHashIndex": 8, "headerMask": 9, "bodyMask": 10, "decodedEmailBodyIn": 11}


class Circuit:
    """A compiled circuit handle (`component main = ...` with its template parameters)."""

    def __init__(self, main_kind=MAIN_EMAIL_VERIFIER, max_header=1024, max_body=1536, n=121, k=17,
                 ignore_body_hash_check=0, device=0, enable_header_masking=0, enable_body_masking=0,
                 remove_soft_line_breaks=0, sym=None, sym_alias=None, r1cs=None, regex=None, regex_include_dirs=(),
                 regex_template=None):
        """regex: path of a zk-regex style template (`body_hash_regex.circom`): BodyHashRegex is compiled from that
        file instead of zkwg's built-in DFA circuit (zkwg_circuit_create_regex; regex_include_dirs: where its
        includes are searched; regex_template: template name, default BodyHashRegex).
        sym: text of the compiled circuit's `.sym` file -> the witness follows ITS indices
        (zkwg_circuit_create_sym); sym_alias: optional "ours=theirs" rename rules, one per line;
        r1cs: bytes of the compiled circuit's `.r1cs` -> complete witness of a circuit compiled with
        --O0 / --O1: signals the schedule does not produce are derived from its linear constraints
        (zkwg_circuit_create_full)."""
        self.lib = _lib.load()
        self.cfg = Config(main_kind, max_header, max_body, n, k, ignore_body_hash_check, enable_header_masking,
                          enable_body_masking, remove_soft_line_breaks, 0)
        h = C.c_void_p()
        if regex is not None:
            src = _lib.RegexSource(str(regex).encode(), ":".join(str(d) for d in regex_include_dirs).encode(),
                                   regex_template.encode() if regex_template else None)
            sb = None if sym is None else (sym.encode() if isinstance(sym, str) else bytes(sym))
            ab = None if sym_alias is None else (sym_alias.encode() if isinstance(sym_alias, str) else bytes(sym_alias))
            rb = None if r1cs is None else bytes(r1cs)
            rc = self.lib.zkwg_circuit_create_regex(C.byref(self.cfg), device, C.byref(src), sb, len(sb) if sb else 0,
                                                    ab, len(ab) if ab else 0, rb, len(rb) if rb else 0, C.byref(h))
        elif sym is None:
            rc = self.lib.zkwg_circuit_create(C.byref(self.cfg), device, C.byref(h))
        else:
            sb = sym.encode() if isinstance(sym, str) else bytes(sym)
            ab = None if sym_alias is None else (sym_alias.encode() if isinstance(sym_alias, str) else bytes(sym_alias))
            if r1cs is None:
                rc = self.lib.zkwg_circuit_create_sym(C.byref(self.cfg), device, sb, len(sb), ab, len(ab) if ab else 0, C.byref(h))
            else:
                rb = bytes(r1cs)
                rc = self.lib.zkwg_circuit_create_full(C.byref(self.cfg), device, sb, len(sb), ab, len(ab) if ab else 0,
                                                       rb, len(rb), C.byref(h))
        if rc == -1:
            raise ZkwgError(f"{self.lib.zkwg_strerror(rc).decode()}: {self.lib.zkwg_last_error().decode()}")
        _check(rc)
        self._sym_layout = sym is not None
        self.h = h
        self.device = device
        self.W = self.lib.zkwg_witness_len(h)
        self.witness_bytes = self.lib.zkwg_witness_bytes(h)
        self.n_public = self.lib.zkwg_num_public(h)
        self.in_stride = self.lib.zkwg_input_stride(h)

    def regex_info(self):
        """loaded regex template: gate-list statistics (zkwg_regex_info)"""
        out = (C.c_uint64 * 8)()
        _check(self.lib.zkwg_regex_info(self.h, out))
        keys = ("kept", "temporaries", "gates", "asserts", "chunks", "steps", "lds_value_words", "gates_64bit")
        return dict(zip(keys, [int(x) for x in out]))

    def close(self):
        if getattr(self, "h", None):
            self.lib.zkwg_circuit_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- input marshalling (CircuitInput -> packed record) -------------------------------
    def _signal_sizes(self):
        c = self.cfg
        if c.main_kind == MAIN_SHA256_BYTES:
            return {"paddedIn": c.max_header, "paddedInLength": 1}
        if c.main_kind == MAIN_RSA_VERIFIER:
            return {"message": c.k, "signature": c.k, "modulus": c.k}
        if c.main_kind == MAIN_FP_MUL:
            return {"a": c.k, "b": c.k, "p": c.k}
        sizes = {"emailHeader": c.max_header, "emailHeaderLength": 1, "pubkey": c.k, "signature": c.k}
        if c.enable_header_masking:
            sizes["headerMask"] = c.max_header
        if not c.ignore_body_hash_check:
            sizes.update({"bodyHashIndex": 1, "precomputedSHA": 32, "emailBody": c.max_body,
                          "emailBodyLength": 1})
            if c.remove_soft_line_breaks:
                sizes["decodedEmailBodyIn"] = c.max_body
            if c.enable_body_masking:
                sizes["bodyMask"] = c.max_body
        return sizes

    def pack(self, inp):
        """CircuitInput dict (numbers / decimal strings / lists of them) -> packed record bytes.
        Error messages follow circom_runtime."""
        sizes = self._signal_sizes()
        flat = {}
        for key, val in inp.items():
            if key not in sizes:
                raise ZkwgError(f"Signal not found: {key}")
            vals = [int(x) for x in val] if isinstance(val, (list, tuple)) else [int(val)]
            if len(vals) > sizes[key]:
                raise ZkwgError(f"Too many values for input signal {key}")
            if len(vals) < sizes[key]:
                raise ZkwgError(f"Not enough values for input signal {key}")
            flat[key] = [v % FIELD_MODULUS for v in vals]
        if len(flat) != len(sizes):
            raise ZkwgError(f"Not all inputs have been set. Only {len(flat)} out of {len(sizes)}")

        # A value that does not fit its packed slot goes through the generic 32-byte-per-signal path
        # (zkwg_pack_field, after the packed fields are written): the record keeps its low bits plus a
        # range flag and the circuit's own range check of that signal fails the email ("Assert Failed"),
        # as circom_runtime does for the same input.
        generic = []

        def as_bytes(vals, what):
            if any(v > 255 for v in vals):
                generic.append((_FIELD_OF[what], vals))
                return bytes(v & 0xFF for v in vals)
            return bytes(vals)

        def as_u32(v, what):
            if v >> 32:
                generic.append((_FIELD_OF[what], [v]))
                return v & 0xFFFFFFFF
            return v

        def as_limbs(vals, what):
            if any(v >> 128 for v in vals):
                generic.append((_FIELD_OF[what], vals))
            return _limbs_bytes([v & ((1 << 128) - 1) for v in vals] + [0] * (17 - len(vals)))

        c = self.cfg
        rec = (C.c_uint8 * self.in_stride)()
        header = body = pre = pub = sig = msg = None
        hlen = blen = bhi = 0
        if c.main_kind == MAIN_SHA256_BYTES:
            header = as_bytes(flat["paddedIn"], "paddedIn")
            hlen = as_u32(flat["paddedInLength"][0], "paddedInLength")
        elif c.main_kind == MAIN_RSA_VERIFIER:
            msg = as_limbs(flat["message"], "message"); sig = as_limbs(flat["signature"], "signature")
            pub = as_limbs(flat["modulus"], "modulus")
        elif c.main_kind == MAIN_FP_MUL:
            # the chunks of a, b, p travel in the pubkey / signature / message slots of the record
            pub = as_limbs(flat["a"], "a"); sig = as_limbs(flat["b"], "b"); msg = as_limbs(flat["p"], "p")
        else:
            header = as_bytes(flat["emailHeader"], "emailHeader")
            hlen = as_u32(flat["emailHeaderLength"][0], "emailHeaderLength")
            pub = as_limbs(flat["pubkey"], "pubkey"); sig = as_limbs(flat["signature"], "signature")
            if not c.ignore_body_hash_check:
                body = as_bytes(flat["emailBody"], "emailBody")
                blen = as_u32(flat["emailBodyLength"][0], "emailBodyLength")
                pre = as_bytes(flat["precomputedSHA"], "precomputedSHA")
                bhi = as_u32(flat["bodyHashIndex"][0], "bodyHashIndex")
        _check(self.lib.zkwg_pack_input(self.h, rec, header, hlen, body, blen, pre, pub, sig, msg, bhi))
        if c.main_kind == MAIN_EMAIL_VERIFIER and (c.enable_header_masking or c.enable_body_masking):
            hm = as_bytes(flat["headerMask"], "headerMask") if c.enable_header_masking else None
            bm = as_bytes(flat["bodyMask"], "bodyMask") if c.enable_body_masking else None
            _check(self.lib.zkwg_pack_masks(self.h, rec, hm, bm))
        if c.main_kind == MAIN_EMAIL_VERIFIER and c.remove_soft_line_breaks:
            _check(self.lib.zkwg_pack_decoded_body(self.h, rec, as_bytes(flat["decodedEmailBodyIn"], "decodedEmailBodyIn")))
        for field, vals in generic:
            buf = b"".join(int(v).to_bytes(32, "little") for v in vals)
            _check(self.lib.zkwg_pack_field(self.h, rec, field, 0, buf, len(vals)))
        return bytes(rec)

    # -- batch calculation -------------------------------------------------------------------
    def calculate_batch_host(self, records, want_witness=True, max_tile=0):
        """records: bytes of n packed records.  Returns (witness bytes or None, status list)."""
        n = len(records) // self.in_stride
        assert n * self.in_stride == len(records)
        status = (C.c_int32 * n)()
        out = (C.c_uint8 * (n * self.witness_bytes))() if want_witness else None
        _check(self.lib.zkwg_calculate_batch(self.h, records, n, out, self.witness_bytes, status, max_tile))
        return (bytes(out) if want_witness else None), list(status)

    def set_host_expand(self, threads):
        """threads > 0: calculate_batch_host expands on the host from the downloaded image (zkwg_set_host_expand); 0: on the device."""
        _check(self.lib.zkwg_set_host_expand(self.h, int(threads)))

    def expand_host(self, records, n, scratch_host, first, count, threads=1):
        """zkwg_expand_host: witnesses of emails [first, first+count) from a host copy of the scratch buffer -> bytes"""
        out = (C.c_uint8 * (count * self.witness_bytes + 16))()
        addr = C.addressof(out)
        off = (-addr) % 16
        _check(self.lib.zkwg_expand_host(self.h, records, n, scratch_host, first, count, C.c_void_p(addr + off), self.witness_bytes, threads))
        return bytes(memoryview(out)[off:off + count * self.witness_bytes])

    def time_host_path(self, records, n, max_tile=64, pinned=True, repeats=2):
        """Seconds of one zkwg_calculate_batch call that delivers n witnesses to host memory
        (pinned via zkwg_alloc_pinned, or pageable); the first call (staging-buffer allocation) is
        not the one reported."""
        import time
        nbytes = n * self.witness_bytes
        status = (C.c_int32 * n)()
        ptr = None
        if pinned:
            self.lib.zkwg_alloc_pinned.restype = C.c_void_p
            ptr = self.lib.zkwg_alloc_pinned(nbytes)
            if not ptr:
                raise ZkwgError("zkwg_alloc_pinned failed")
            out = C.cast(ptr, C.POINTER(C.c_uint8))
        else:
            out = (C.c_uint8 * nbytes)()
        try:
            dt = None
            for _ in range(max(1, repeats)):
                t0 = time.perf_counter()
                _check(self.lib.zkwg_calculate_batch(self.h, records, n, out, self.witness_bytes, status, max_tile))
                dt = time.perf_counter() - t0
            if any(status):
                raise ZkwgError("host path: a synthetic email did not verify")
            return dt
        finally:
            if ptr:
                self.lib.zkwg_free_pinned.argtypes = [C.c_void_p]
                self.lib.zkwg_free_pinned(ptr)

    def calculate_batch_device(self, d_in, n, d_out, d_status, d_scratch, stream=None):
        """Device-resident launch; arguments are torch CUDA tensors (plumbing only)."""
        sp = stream.cuda_stream if stream is not None else 0
        _check(self.lib.zkwg_calculate_batch_device(self.h, d_in.data_ptr(), n, d_out.data_ptr(),
                                                    self.witness_bytes, d_status.data_ptr(),
                                                    d_scratch.data_ptr(), sp))

    def prepare_device(self, d_in, n, d_status, d_scratch, stream=None):
        """Phase 1: all compute kernels for n emails -> compact images in d_scratch."""
        sp = stream.cuda_stream if stream is not None else 0
        _check(self.lib.zkwg_prepare_device(self.h, d_in.data_ptr(), n, d_status.data_ptr(), d_scratch.data_ptr(), sp))

    def expand_device(self, d_in, n, d_scratch, first, count, d_out, stream=None, out_stride=None):
        """Phase 2: stream the witnesses of emails [first, first+count) into d_out (`out_stride` bytes apart, default
        back to back; a caller may pad the distance, e.g. to a multiple of 4 KiB)."""
        sp = stream.cuda_stream if stream is not None else 0
        _check(self.lib.zkwg_expand_device(self.h, d_in.data_ptr(), n, d_scratch.data_ptr(), first, count,
                                           d_out.data_ptr(), out_stride or self.witness_bytes, sp))

    def expand_montgomery_device(self, d_in, n, d_scratch, first, count, d_out, stream=None):
        """Phase 2 with Montgomery-form output (x * 2^256 mod r): the fused prover hand-off."""
        sp = stream.cuda_stream if stream is not None else 0
        _check(self.lib.zkwg_expand_montgomery_device(self.h, d_in.data_ptr(), n, d_scratch.data_ptr(), first, count,
                                                      d_out.data_ptr(), self.witness_bytes, sp))

    def attach_r1cs(self, r1cs):
        """Attach a constraint system over this handle's witness layout (bytes of an `.r1cs`, or a zkwg.R1cs) so that
        expand_abc_device can write A.w | B.w | C.w from the compact image (zkwg_circuit_attach_r1cs).  Call it before
        sizing scratch buffers: the image grows."""
        data = r1cs.data if isinstance(r1cs, R1cs) else bytes(r1cs)
        _check(self.lib.zkwg_circuit_attach_r1cs(self.h, data, len(data)))

    @property
    def abc_bytes(self):
        return self.lib.zkwg_abc_bytes(self.h)

    def expand_abc_device(self, d_in, n, d_scratch, first, count, d_out, stream=None, montgomery=False, out_stride=None):
        """A.w | B.w | C.w of emails [first, first + count) of a prepared batch, from the image (zkwg_expand_abc_device)."""
        sp = stream.cuda_stream if stream is not None else 0
        _check(self.lib.zkwg_expand_abc_device(self.h, d_in.data_ptr(), n, d_scratch.data_ptr(), first, count, 1 if montgomery else 0,
                                               d_out.data_ptr(), out_stride or self.abc_bytes, sp))

    def expand_abc_host(self, records, n, scratch_host, first, count, rows_on_host=False):
        """zkwg_expand_abc_host: A.w | B.w | C.w of emails [first, first+count) from a host copy of the scratch buffer (a
        ctypes array; written when rows_on_host) -> bytes"""
        out = (C.c_uint8 * (count * self.abc_bytes + 16))()
        addr = C.addressof(out)
        off = (-addr) % 16
        _check(self.lib.zkwg_expand_abc_host(self.h, records, n, scratch_host, first, count, 1 if rows_on_host else 0,
                                             C.c_void_p(addr + off), self.abc_bytes))
        return bytes(memoryview(out)[off:off + count * self.abc_bytes])

    def expand_full_host(self, records, n, scratch_host, first, count):
        """zkwg_expand_full_host (layout-only handle of a numbered circuit): complete witnesses from host images -> bytes"""
        out = (C.c_uint8 * (count * self.witness_bytes + 16))()
        addr = C.addressof(out)
        off = (-addr) % 16
        _check(self.lib.zkwg_expand_full_host(self.h, records, n, scratch_host, first, count, C.c_void_p(addr + off), self.witness_bytes))
        return bytes(memoryview(out)[off:off + count * self.witness_bytes])

    def calculate_batch_resident(self, records, tile=0, prep=0, consumer=None, want_table=True):
        """zkwg_calculate_batch_resident: the device-resident two-stream pipeline below the boundary.  `consumer(device, d_tile,
        witness_stride, first_email, count, hip_stream)` (optional) is called once per expanded tile.  -> (status list, table
        bytes (n x 100) or None)"""
        n = len(records) // self.in_stride
        status = (C.c_int32 * n)()
        table = (C.c_uint8 * (100 * n))() if want_table else None
        cb_t = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p)
        cb = cb_t((lambda user, dev, tile_p, stride, first, count, stream: consumer(dev, tile_p, stride, first, count, stream))) if consumer else None
        _check(self.lib.zkwg_calculate_batch_resident(self.h, records, n, status, table, tile, prep,
                                                       C.cast(cb, C.c_void_p) if cb else None, None))
        return list(status), (bytes(table) if want_table else None)

    def resident_placement(self):
        ms = (C.c_float * 8)()
        kept = (C.c_int * 2)()
        k = self.lib.zkwg_resident_placement(self.h, ms, 8, kept)
        if k == 0:      # since round 5 the ring is mapped from 1 GiB physical chunks (zkwg_device_alloc_chunked): nothing to choose from
            return {"mode": "chunked (ZKWG_PLACE_RING=0: plain allocations)", "chunk_bytes": 1 << 30}
        return {"ms_per_tile": [round(ms[i], 3) for i in range(k)], "kept": list(kept)}

    def release_resident(self):
        """give the buffers calculate_batch_resident keeps in the handle back to the device (include/zkwg.h zkwg_resident_release)"""
        _check(self.lib.zkwg_resident_release(self.h))

    def scratch_bytes(self, n, montgomery=True):
        """device scratch of an n-email batch; montgomery=False: without the Montgomery-copy area at its end (include/zkwg.h
        zkwg_scratch_bytes_standard: every entry point except the Montgomery-form outputs)"""
        return self.lib.zkwg_scratch_bytes(self.h, n) if montgomery else self.lib.zkwg_scratch_bytes_standard(self.h, n)

    def set_prepare_throttle(self, rsa_wavefronts_per_cu):
        _check(self.lib.zkwg_set_prepare_throttle(self.h, rsa_wavefronts_per_cu))

    def set_prepare_mask(self, mask):
        """measurement aid: launch only the prepare kernels whose bit is set (include/zkwg.h zkwg_set_prepare_mask)"""
        _check(self.lib.zkwg_set_prepare_mask(self.h, mask & 0xFFFFFFFF))

    def set_timing(self, on):
        _check(self.lib.zkwg_set_timing(self.h, 1 if on else 0))

    def kernel_times_ms(self):
        out = {}
        for i in range(self.lib.zkwg_num_kernels(self.h)):
            ms = C.c_float()
            _check(self.lib.zkwg_last_kernel_ms(self.h, i, C.byref(ms)))
            out[self.lib.zkwg_kernel_name(self.h, i).decode()] = (ms.value, self.lib.zkwg_kernel_slots(self.h, i))
        return out

    def timing_summary(self):
        """{kernel: (total_ms, launches, slots_per_email)} over the launches since set_timing(True)."""
        out = {}
        for i in range(self.lib.zkwg_num_kernels(self.h)):
            ms, cnt = C.c_float(), C.c_uint32()
            _check(self.lib.zkwg_timing_summary(self.h, i, C.byref(ms), C.byref(cnt)))
            out[self.lib.zkwg_kernel_name(self.h, i).decode()] = (ms.value, cnt.value, self.lib.zkwg_kernel_slots(self.h, i))
        return out

    def wtns(self, witness_bytes):
        size = self.lib.zkwg_wtns_size(self.h)
        out = (C.c_uint8 * size)()
        _check(self.lib.zkwg_write_wtns(self.h, witness_bytes, out, size))
        return bytes(out)

    def image_layout(self, n):
        """Scratch layout of a prepared batch of n emails (include/zkwg.h zkwg_image_layout): dict of the
        per-email sizes and the byte offsets of the hstates / bits / small / fr arrays."""
        out = (C.c_uint64 * 9)()
        _check(self.lib.zkwg_image_layout(self.h, n, out))
        keys = ("hstate_words", "bits_words", "small_words", "fr_elems", "off_hstates", "off_bits", "off_small", "off_fr", "total_bytes")
        return dict(zip(keys, [int(x) for x in out]))

    def segment_table(self):
        """[(slot, nslots, type, src, a, b, c, r0)] -- the static table zk_expand evaluates (zkwg_segment_table)."""
        n = self.lib.zkwg_segment_table(self.h, None, 0)
        raw = (C.c_uint8 * (40 * n))()
        self.lib.zkwg_segment_table(self.h, raw, n)
        import struct
        return [struct.unpack_from("<QIIIIIIII", raw, 40 * i)[:8] for i in range(n)]

    def layout_map(self):
        """`.sym` layouts: list, per slot of the default (kept-v1) layout, of its witness index (None = eliminated)."""
        n = self.lib.zkwg_layout_map(self.h, None, 0)
        out = (C.c_uint32 * n)()
        self.lib.zkwg_layout_map(self.h, out, n)
        return [None if v == 0xFFFFFFFF else v for v in out]

    def o0_gather_host(self, kept_witness):
        """layout-only handle of a fully numbered circuit: the complete witness from one kept-v1 witness through the
        wire table the device kernels use (zkwg_o0_gather_host)."""
        out = (C.c_uint8 * self.witness_bytes)()
        _check(self.lib.zkwg_o0_gather_host(self.h, bytes(kept_witness), out))
        return bytes(out)

    def linear_complete_host(self, witness):
        """layout-only handle of a fully numbered circuit: derive the non-produced signals of one host witness
        (bytearray of 32 W bytes whose produced slots are filled) in place."""
        arr = (C.c_uint8 * len(witness)).from_buffer(witness)
        _check(self.lib.zkwg_linear_complete_host(self.h, arr))

    def symbols(self):
        """[(slot, name)] of the layout (the .sym table)."""
        need = self.lib.zkwg_write_sym(self.h, None, 0)
        buf = C.create_string_buffer(need)
        self.lib.zkwg_write_sym(self.h, buf, need)
        out = []
        for line in buf.raw[:need].decode().splitlines():
            a, _, _, name = line.split(",", 3)
            out.append((int(a), name))
        return out


    def loadSymbols(self):
        """circom_tester `loadSymbols()`: {"main.x[3]": {"labelIdx", "varIdx", "componentIdx"}}
        (packages/circuits/tests/email-verifier.test.ts:204-206 reaches it through assertOut)."""
        if getattr(self, "_symbols", None) is None:
            need = self.lib.zkwg_write_sym(self.h, None, 0)
            buf = C.create_string_buffer(need)
            self.lib.zkwg_write_sym(self.h, buf, need)
            self._symbols = {}
            for line in buf.raw[:need].decode().splitlines():
                a, b, c, name = line.split(",", 3)
                self._symbols[name] = {"labelIdx": int(a), "varIdx": int(b), "componentIdx": int(c)}
        return self._symbols

    def assertOut(self, witness, expected):
        """circom_tester `assertOut(witness, {name: value | list})`: compares `main.<name>` (lists element-wise) with the
        witness (list of ints or the 32-byte-per-signal buffer); raises AssertionError on the first difference."""
        sym = self.loadSymbols()
        get = (lambda i: int.from_bytes(witness[32 * i:32 * i + 32], "little")) if isinstance(witness, (bytes, bytearray, memoryview)) else (lambda i: int(witness[i]))

        def check(prefix, e):
            if isinstance(e, (list, tuple)):
                for i, x in enumerate(e):
                    check(f"{prefix}[{i}]", x)
            elif isinstance(e, dict):
                for k, x in e.items():
                    check(f"{prefix}.{k}", x)
            else:
                if prefix not in sym:
                    raise AssertionError("Output variable not defined: " + prefix)
                got, want = get(sym[prefix]["varIdx"]), int(e) % 21888242871839275222246405745257275088548364400416034343698204186575808495617
                if got != want:
                    raise AssertionError(f"{prefix}: expected {want}, the witness has {got}")
        check("main", expected)


def _stream_ptr(stream):
    return stream.cuda_stream if stream is not None else 0


class Ntt:
    """Plan of the transform stage that follows A.w | B.w | C.w in groth16.prove (include/zkwg.h "prover stage 2"):
    domain of 2^log2_n points, everything in Montgomery form on the device."""

    def __init__(self, log2_n, device=0):
        self.lib = _lib.load()
        h = C.c_void_p()
        _check(self.lib.zkwg_ntt_create(device, log2_n, C.byref(h)))
        self.h = h
        self.log2_n, self.n, self.device = log2_n, 1 << log2_n, device

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.zkwg_ntt_destroy(self.h)
            self.h = None

    def work_bytes(self, n_emails):
        return self.lib.zkwg_ntt_work_bytes(self.h, n_emails)

    def transform_device(self, d_data, n_polys, inverse=False, stream=None):
        """in place: n_polys arrays of 2^log2_n Montgomery-form elements, natural order in and out (Fr.fft / Fr.ifft)"""
        _check(self.lib.zkwg_ntt_transform_device(self.h, d_data.data_ptr(), n_polys, 1 if inverse else 0, _stream_ptr(stream)))

    def h_evaluations_device(self, d_abc, abc_stride, n_constraints, n_emails, d_work, d_out, out_stride=None, stream=None):
        """a(x) b(x) - c(x) on the odd coset for n_emails records of A.w | B.w | C.w (Montgomery form) -> d_out"""
        _check(self.lib.zkwg_h_evaluations_device(self.h, d_abc.data_ptr(), abc_stride, n_constraints, n_emails, d_work.data_ptr(),
                                                  d_out.data_ptr(), out_stride if out_stride else 32 * self.n, _stream_ptr(stream)))


class Msm:
    """One G1 multi-exponentiation of groth16.prove (include/zkwg.h "prover stage 3"): `bases` = n affine points as the zkey
    stores them (x | y little-endian Montgomery limbs, 64 bytes each, zeros = infinity), resident on the device."""
    Q = 21888242871839275222246405745257275088696311157297823662689037894645226208583

    def __init__(self, bases, device=0, window_bits=0):
        self.lib = _lib.load()
        assert len(bases) % 64 == 0
        self.n = len(bases) // 64
        h = C.c_void_p()
        _check(self.lib.zkwg_msm_create(device, bytes(bases), self.n, window_bits, C.byref(h)))
        self.h, self.device = h, device
        self.window_bits = self.lib.zkwg_msm_window_bits(h)

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.zkwg_msm_destroy(self.h)
            self.h = None

    @staticmethod
    def pack_bases(points):
        """[(x, y) | None] (standard-form integers) -> the zkey's byte layout (Montgomery form)"""
        out = bytearray()
        for p in points:
            if p is None:
                out += bytes(64)
            else:
                out += ((p[0] << 256) % Msm.Q).to_bytes(32, "little") + ((p[1] << 256) % Msm.Q).to_bytes(32, "little")
        return bytes(out)

    def work_bytes(self, n_emails=1):
        return self.lib.zkwg_msm_work_bytes_batch(self.h, n_emails)

    def g1_batch_device(self, d_scalars, n_emails, montgomery, d_work, stream=None, ones_apart=False):
        """n_emails sums in ONE launch series: the scalar vectors lie back to back at d_scalars -> [(x, y) | None] (standard form)"""
        import torch
        d_out = torch.empty(128 * n_emails, dtype=torch.uint8, device=d_scalars.device)
        _check(self.lib.zkwg_msm_enqueue_batch_device(self.h, d_scalars.data_ptr(), 32 * self.n, n_emails, 1 if montgomery else 0, 1 if ones_apart else 0,
                                                      d_work.data_ptr(), d_out.data_ptr(), _stream_ptr(stream)))
        torch.cuda.synchronize(d_scalars.device)
        raw = bytes(d_out.cpu().numpy())
        pts = (C.c_uint8 * (64 * n_emails))()
        _check(self.lib.zkwg_msm_finish_host(1, raw, n_emails, pts))
        rinv = pow(1 << 256, -1, Msm.Q)
        out = []
        for e in range(n_emails):
            b = bytes(pts)[64 * e:64 * e + 64]
            out.append(None if b == bytes(64) else (int.from_bytes(b[:32], "little") * rinv % Msm.Q, int.from_bytes(b[32:], "little") * rinv % Msm.Q))
        return out

    def g1_device(self, d_scalars, montgomery, d_work, stream=None, ones_apart=False):
        """sum_i scalar_i * base_i for the n 32-byte scalars at d_scalars (torch tensor) -> (x, y) standard-form integers or None"""
        out = (C.c_uint8 * 64)()
        _check(self.lib.zkwg_msm_g1_device(self.h, d_scalars.data_ptr(), 1 if montgomery else 0, 1 if ones_apart else 0, d_work.data_ptr(), out,
                                           _stream_ptr(stream)))
        raw = bytes(out)
        if raw == bytes(64):
            return None
        rinv = pow(1 << 256, -1, Msm.Q)
        return (int.from_bytes(raw[:32], "little") * rinv % Msm.Q, int.from_bytes(raw[32:], "little") * rinv % Msm.Q)


class MultiCircuit:
    """The batch sharded over several GPUs of one node through the C-ABI (include/zkwg.h zkwg_multi_*):
    contiguous shards, one handle + host thread per GPU, the 100-byte result table gathered on devices[0]
    over RCCL.  `circuit` is a layout-only handle for packing and geometry."""

    def __init__(self, devices, **circuit_kwargs):
        self.circuit = Circuit(device=-1, **circuit_kwargs)
        self.lib = self.circuit.lib
        devs = (C.c_int * len(devices))(*devices)
        h = C.c_void_p()
        rc = self.lib.zkwg_multi_create(C.byref(self.circuit.cfg), devs, len(devices), C.byref(h))
        if rc != 0:
            raise ZkwgError(self.lib.zkwg_strerror(rc).decode() + ": " + self.lib.zkwg_last_error().decode())
        self.h = h
        self.n_devices = self.lib.zkwg_multi_devices(h)

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.zkwg_multi_destroy(self.h)
            self.h = None

    def calculate_batch_host(self, records, want_witness=True, max_tile=0):
        """-> (witness bytes or None, status list, table rows [(status, pubkeyHash, shaHi, shaLo)])"""
        c = self.circuit
        n = len(records) // c.in_stride
        status = (C.c_int32 * n)()
        out = (C.c_uint8 * (n * c.witness_bytes))() if want_witness else None
        table = (C.c_uint8 * (n * 100))()
        _check(self.lib.zkwg_calculate_batch_multi(self.h, records, n, out, c.witness_bytes, status, table, max_tile))
        tb = bytes(table)
        rows = [(int.from_bytes(tb[100 * i:100 * i + 4], "little", signed=True),
                 int.from_bytes(tb[100 * i + 4:100 * i + 36], "little"),
                 int.from_bytes(tb[100 * i + 36:100 * i + 68], "little"),
                 int.from_bytes(tb[100 * i + 68:100 * i + 100], "little")) for i in range(n)]
        return (bytes(out) if want_witness else None), list(status), rows


def shard_range(n, n_shards, i):
    """C-ABI zkwg_shard_range: (first, count) of shard i."""
    lib = _lib.load()
    f, c = C.c_uint64(), C.c_uint64()
    lib.zkwg_shard_range(n, n_shards, i, C.byref(f), C.byref(c))
    return f.value, c.value


def convert_montgomery_device(d_values, n_values, to_montgomery=True, stream=None):
    """In-place standard <-> Montgomery form of n_values field elements in a torch CUDA uint8 tensor
    (prover hand-off, zkwg_convert_montgomery_device)."""
    sp = stream.cuda_stream if stream is not None else 0
    _check(_lib.load().zkwg_convert_montgomery_device(d_values.data_ptr(), n_values, 1 if to_montgomery else 0, sp))


class R1cs:
    """A compiled circuit's `.r1cs` constraint system, loaded for `checkConstraints` on the device
    (circom_tester `circuit.checkConstraints(witness)`, packages/circuits/tests/email-verifier.test.ts:44)."""

    def __init__(self, data, device=0):
        self.lib = _lib.load()
        data = bytes(data)
        self.data = data
        h = C.c_void_p()
        _check(self.lib.zkwg_r1cs_load(data, len(data), device, C.byref(h)))
        self.h = h
        info = (C.c_uint64 * 6)()
        _check(self.lib.zkwg_r1cs_info(h, info))
        (self.n_wires, self.n_pub_out, self.n_pub_in, self.n_prv_in, self.n_constraints, self.n_labels) = [int(x) for x in info]

    def close(self):
        if getattr(self, "h", None):
            self.lib.zkwg_r1cs_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def first_violations(self, witnesses, n=None, stride=None):
        """witnesses: bytes of n witnesses (32-byte LE values).  -> per witness: index of the first
        violated constraint, or None."""
        stride = stride or 32 * self.n_wires
        n = n if n is not None else len(witnesses) // stride
        bad = (C.c_uint64 * n)()
        _check(self.lib.zkwg_check_constraints(self.h, bytes(witnesses), n, stride, bad))
        return [None if b == 0xFFFFFFFFFFFFFFFF else int(b) for b in bad]

    def first_violations_device(self, d_witness, n, stride, stream=None):
        """d_witness: torch uint8 CUDA tensor holding n witnesses `stride` bytes apart."""
        import torch
        bad = torch.empty(n, dtype=torch.int64, device=d_witness.device)
        st = stream.cuda_stream if stream is not None else torch.cuda.current_stream().cuda_stream
        _check(self.lib.zkwg_check_constraints_device(self.h, d_witness.data_ptr(), n, stride, bad.data_ptr(), st))
        return [None if b == -1 else int(b) for b in bad.cpu().tolist()]

    def evaluate_device(self, d_witness, n, stride, stream=None, montgomery=False):
        """First Groth16 prover stage on device-resident witnesses (zkwg_r1cs_evaluate_device): a torch uint8 CUDA
        tensor [n, 3 * n_constraints * 32] with the A, B, C evaluations of every constraint (Montgomery form in ->
        Montgomery form out)."""
        import torch
        out = torch.empty((n, 96 * self.n_constraints), dtype=torch.uint8, device=d_witness.device)
        st = stream.cuda_stream if stream is not None else torch.cuda.current_stream().cuda_stream
        _check(self.lib.zkwg_r1cs_evaluate_device(self.h, d_witness.data_ptr(), n, stride, 1 if montgomery else 0, out.data_ptr(),
                                                  96 * self.n_constraints, st))
        return out

    def checkConstraints(self, witness):
        """circom_tester semantics on one witness given as a list of ints: raises on the first mismatch."""
        blob = b"".join(int(v).to_bytes(32, "little") for v in witness)
        if len(witness) != self.n_wires:
            raise ZkwgError(f"Invalid witness length. Circuit: {self.n_wires}, witness: {len(witness)}")
        bad = self.first_violations(blob, 1)[0]
        if bad is not None:
            raise ZkwgError(f"Constraint doesn't match (constraint {bad})")


class WitnessCalculator:
    """circom_runtime `WitnessCalculator`-shaped front end over a `Circuit`."""

    def __init__(self, circuit):
        self.circuit = circuit

    def _run_one(self, inp):
        rec = self.circuit.pack(inp)
        wit, status = self.circuit.calculate_batch_host(rec)
        if status[0] != 0:
            raise ZkwgError(self.circuit.lib.zkwg_strerror(status[0]).decode())
        return wit

    def calculateBinWitness(self, inp, sanityCheck=False):
        return self._run_one(inp)

    def calculateWitness(self, inp, sanityCheck=False):
        b = self._run_one(inp)
        return [int.from_bytes(b[i:i + 32], "little") for i in range(0, len(b), 32)]

    def calculateWTNSBin(self, inp, sanityCheck=False):
        return self.circuit.wtns(self._run_one(inp))

    def calculateBatch(self, inputs):
        recs = b"".join(self.circuit.pack(i) for i in inputs)
        wit, status = self.circuit.calculate_batch_host(recs)
        wb = self.circuit.witness_bytes
        return [wit[i * wb:(i + 1) * wb] for i in range(len(inputs))], status

    def constraint_system(self):
        """The circuit's R1CS in the layout this calculator emits (zkwg.r1cs, derived from the reference templates),
        loaded on the circuit's device; built once per calculator."""
        if getattr(self, "_r1cs", None) is None:
            from . import r1cs as zr
            c = self.circuit
            cfg = c.cfg
            if cfg.layout != 0 or getattr(c, "_sym_layout", False):
                raise ZkwgError("constraint system export covers the built-in layout only; load the compiler's .r1cs with zkwg.R1cs")
            sym = c.symbols()
            if cfg.main_kind == MAIN_SHA256_BYTES:
                data = zr.write_r1cs(len(sym), zr.sha256_main_constraints(sym, cfg.max_header), 256, cfg.max_header + 1)
            elif cfg.main_kind == MAIN_RSA_VERIFIER:
                data = zr.write_r1cs(len(sym), zr.rsa_main_constraints(sym), 0, 17, 34)
            elif cfg.main_kind == MAIN_FP_MUL:
                data = zr.write_r1cs(len(sym), zr.fp_mul_main_constraints(sym, cfg.n, cfg.k), cfg.k, 0, 3 * cfg.k)
            else:
                data = zr.email_verifier_r1cs(sym, cfg.max_header, cfg.max_body, cfg.enable_header_masking,
                                              cfg.enable_body_masking, cfg.remove_soft_line_breaks, cfg.ignore_body_hash_check)
            self._r1cs = R1cs(data, device=c.device)
        return self._r1cs

    def checkConstraints(self, witness):
        """circom_tester `await circuit.checkConstraints(witness)` (packages/circuits/tests/email-verifier.test.ts:44)
        against the constraint system of this layout, on the device."""
        self.constraint_system().checkConstraints(witness)


def witness_ints(b):
    return [int.from_bytes(b[i:i + 32], "little") for i in range(0, len(b), 32)]


def generate_inputs_device(circuit, dkim_results, selector=None, stream=None):
    """Batched input generation on the GPU (zkwg_generate_inputs_device).  dkim_results: list of dicts with
    headers, body, bodyHash, publicKey, signature (as `verifyDKIMSignature` returns them).  Returns
    (records: torch.uint8[n, in_stride] on the device, gen_status: list[int])."""
    import torch
    from ._lib import DkimBatch
    n = len(dkim_results)
    dev = torch.device("cuda", circuit.device)
    hs = max(len(d["headers"]) for d in dkim_results)
    bs = max(len(d.get("body") or b"") for d in dkim_results) or 1

    def pack(rows, width):
        buf = bytearray(n * width)
        for i, r in enumerate(rows):
            buf[i * width:i * width + len(r)] = r
        return torch.frombuffer(buf, dtype=torch.uint8).to(dev)

    t_hdr = pack([d["headers"] for d in dkim_results], hs)
    t_hl = torch.tensor([len(d["headers"]) for d in dkim_results], dtype=torch.int32, device=dev)
    t_body = pack([d.get("body") or b"" for d in dkim_results], bs)
    t_bl = torch.tensor([len(d.get("body") or b"") for d in dkim_results], dtype=torch.int32, device=dev)
    t_bh = pack([(d.get("bodyHash") or "").encode().ljust(44, b"\0")[:44] for d in dkim_results], 44)
    t_pk = pack([int(d["publicKey"]).to_bytes(256, "big") for d in dkim_results], 256)
    t_sg = pack([int(d["signature"]).to_bytes(256, "big") for d in dkim_results], 256)
    sel = selector.encode() if isinstance(selector, str) else (selector or b"")
    t_sel = torch.frombuffer(bytearray(sel or b"\0"), dtype=torch.uint8).to(dev)
    b = DkimBatch(t_hdr.data_ptr(), t_hl.data_ptr(), t_body.data_ptr(), t_bl.data_ptr(), t_bh.data_ptr(),
                  t_pk.data_ptr(), t_sg.data_ptr(), t_sel.data_ptr() if sel else None, hs, bs, len(sel))
    recs = torch.empty((n, circuit.in_stride), dtype=torch.uint8, device=dev)
    st = torch.zeros(n, dtype=torch.int32, device=dev)
    sp = stream.cuda_stream if stream is not None else 0
    _check(circuit.lib.zkwg_generate_inputs_device(circuit.h, C.byref(b), n, recs.data_ptr(), st.data_ptr(), sp))
    torch.cuda.synchronize()
    return recs, st.cpu().tolist()
