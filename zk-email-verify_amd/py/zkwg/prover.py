"""The second half of `snarkjs.groth16.fullProve` on the device (reference call site: packages/helpers/src/chunked-zkey.ts:80-84):
inputs -> witness -> A.w | B.w | C.w -> H evaluations -> the five multi-exponentiations -> pi_a, pi_b, pi_c.

Host side of include/zkwg.h's prover stages 1-3 (zkwg_expand_abc_device, zkwg_h_evaluations_device, zkwg_msm_*,
zkwg_groth16_assemble).  The proving key's bases stay resident on the device in the zkey's own point layout; per email only scalars
move.  torch is plumbing (device buffers, streams).

A key comes from a `.zkey` (zkwg.zkey.read_zkey) or -- tests and tools, where no ceremony output exists -- from the discrete
logarithms of its points (ProvingKey.from_scalars: zkwg_fixed_base_device turns them into bases on the device).
"""
import ctypes as C

from . import _lib
from . import Ntt, ZkwgError, _check, _stream_ptr

R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
Q = 21888242871839275222246405745257275088696311157297823662689037894645226208583


def _scalars_bytes(xs):
    return b"".join(int(x % R).to_bytes(32, "little") for x in xs)


def fixed_base(device, group, scalars):
    """[k_i] -> device tensor of k_i * G (group 1: 64-byte points, group 2: 128-byte points; affine, Montgomery form)"""
    import torch
    lib = _lib.load()
    dev = torch.device("cuda", device)
    n = len(scalars)
    d_k = torch.frombuffer(bytearray(_scalars_bytes(scalars)), dtype=torch.uint8).to(dev)
    d_out = torch.empty(n * (64 if group == 1 else 128), dtype=torch.uint8, device=dev)
    _check(lib.zkwg_fixed_base_device(device, group, d_k.data_ptr(), n, d_out.data_ptr(), 0))
    torch.cuda.synchronize(dev)
    return d_out


class _DeviceMsm:
    """multi-exponentiation over bases that already live on the device (zkwg_msm_create_device)"""

    def __init__(self, d_bases, group, device, window_bits=0, slice0=0):
        self.lib = _lib.load()
        self.group, self.device = group, device
        self.n = d_bases.numel() // (64 if group == 1 else 128)
        h = C.c_void_p()
        # (the plan keeps its own table of the bases in the kernels' form: the tensor is only read here)
        _check(self.lib.zkwg_msm_create_ex(device, group, d_bases.data_ptr(), 1, self.n, window_bits, slice0, 0, C.byref(h)))
        self.h = h

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.zkwg_msm_destroy(self.h)
            self.h = None

    def work_bytes(self, n_emails=1):
        return self.lib.zkwg_msm_work_bytes_batch(self.h, n_emails)

    def run_batch(self, scalars_ptr, scalar_stride, n_emails, montgomery, ones_apart, d_work, stream=None):
        """n_emails sums in one launch series -> list of the sums as the zkey would store them (bytes: 64 / 128 each)"""
        import torch
        size = 128 if self.group == 1 else 256
        d_out = torch.empty(size * n_emails, dtype=torch.uint8, device=d_work.device)
        _check(self.lib.zkwg_msm_enqueue_batch_device(self.h, scalars_ptr, scalar_stride, n_emails, 1 if montgomery else 0, 1 if ones_apart else 0,
                                                      d_work.data_ptr(), d_out.data_ptr(), _stream_ptr(stream)))
        torch.cuda.synchronize(d_work.device)
        ps = 64 if self.group == 1 else 128
        pts = (C.c_uint8 * (ps * n_emails))()
        _check(self.lib.zkwg_msm_finish_host(self.group, bytes(d_out.cpu().numpy()), n_emails, pts))
        return [bytes(pts)[ps * e:ps * e + ps] for e in range(n_emails)]

    def enqueue(self, scalars_ptr, montgomery, ones_apart, d_work, out_ptr, stream=None):
        """asynchronous: the sum's accumulator (XYZZ, 128 / 256 bytes) is left at out_ptr on the device"""
        _check(self.lib.zkwg_msm_enqueue_device(self.h, scalars_ptr, 1 if montgomery else 0, 1 if ones_apart else 0, d_work.data_ptr(), out_ptr,
                                                _stream_ptr(stream)))

    def run(self, scalars_ptr, montgomery, ones_apart, d_work, stream=None):
        """-> the sum as the zkey would store it (bytes: 64 / 128, affine Montgomery form, zeros = infinity)"""
        out = (C.c_uint8 * (64 if self.group == 1 else 128))()
        fn = self.lib.zkwg_msm_g1_device if self.group == 1 else self.lib.zkwg_msm_g2_device
        _check(fn(self.h, scalars_ptr, 1 if montgomery else 0, 1 if ones_apart else 0, d_work.data_ptr(), out, _stream_ptr(stream)))
        return bytes(out)


class ProvingKey:
    """bases of one circuit's key on the device + the five verification-key points the prover adds"""

    def __init__(self, device, n_wires, n_public, power, d_a, d_b1, d_b2, d_c, d_h, alpha1, beta1, beta2, delta1, delta2):
        self.device, self.n_wires, self.n_public, self.power = device, n_wires, n_public, power
        self.d_a, self.d_b1, self.d_b2, self.d_c, self.d_h = d_a, d_b1, d_b2, d_c, d_h
        self.alpha1, self.beta1, self.beta2, self.delta1, self.delta2 = alpha1, beta1, beta2, delta1, delta2
        assert d_a.numel() == 64 * n_wires and d_b1.numel() == 64 * n_wires and d_b2.numel() == 128 * n_wires
        assert d_c.numel() == 64 * (n_wires - n_public - 1) and d_h.numel() == 64 << power

    @staticmethod
    def from_scalars(device, n_public, power, a, b, c_private, h, alpha, beta, delta):
        """a, b: the discrete logarithms of the A / B bases of every wire; c_private: those of the C bases of the private wires
        (wire nPublic + 1 onwards); h: of the 2^power H bases; alpha, beta, delta: of the key's points (a toy key: tests, tools)"""
        n_wires = len(a)
        assert len(b) == n_wires and len(c_private) == n_wires - n_public - 1 and len(h) == 1 << power
        pts = fixed_base(device, 1, [alpha, beta, delta])
        p2 = fixed_base(device, 2, [beta, delta])
        raw1, raw2 = bytes(pts.cpu().numpy()), bytes(p2.cpu().numpy())
        return ProvingKey(device, n_wires, n_public, power, fixed_base(device, 1, a), fixed_base(device, 1, b), fixed_base(device, 2, b),
                          fixed_base(device, 1, c_private), fixed_base(device, 1, h),
                          raw1[0:64], raw1[64:128], raw2[0:128], raw1[128:192], raw2[128:256])


class _KeyStruct(C.Structure):
    _fields_ = [("n_wires", C.c_uint64), ("n_public", C.c_uint64), ("log2_domain", C.c_uint64),
                ("a", C.c_void_p), ("b1", C.c_void_p), ("b2", C.c_void_p), ("c", C.c_void_p), ("h", C.c_void_p), ("bases_on_device", C.c_int),
                ("alpha1", C.c_uint8 * 64), ("beta1", C.c_uint8 * 64), ("beta2", C.c_uint8 * 128), ("delta1", C.c_uint8 * 64), ("delta2", C.c_uint8 * 128)]


class Prover:
    """groth16.prove for the emails of a prepared batch.  `circuit`: a device handle whose constraint system `r1cs` (bytes of an
    `.r1cs` over its witness layout, WITH the nPublic + 1 rows snarkjs appends to A -- zkwg.r1cs.append_public_rows) is attached here.
    prove_batch is include/zkwg.h's zkwg_prover_prove_prepared (E emails per launch series, rolling contexts; what a Node host binds too); prove_prepared
    runs the same stages one call at a time from Python and keeps the five sums (tests compare each with its discrete logarithm)."""

    def __init__(self, circuit, r1cs, n_constraints, key, stream=None):
        import torch
        self.c, self.key = circuit, key
        self.lib = _lib.load()
        self.m = n_constraints                      # rows of the attached system (public rows included)
        if (1 << key.power) < self.m:
            raise ZkwgError("the key's domain is smaller than the constraint system")
        if key.n_wires != circuit.W:
            raise ZkwgError(f"the key has {key.n_wires} wires, the circuit's witness {circuit.W}")
        circuit.attach_r1cs(r1cs)
        if circuit.abc_bytes != 96 * self.m:
            raise ZkwgError("n_constraints does not match the attached system")
        self.dev = torch.device("cuda", circuit.device)
        self.last_sums = None
        self._single = False
        self._h, self._slots_n = None, 0

    @classmethod
    def from_zkey(cls, circuit, zkey, slots=8):
        """the prover from the zkey alone, as `groth16.prove(zkey, wtns)` takes it (include/zkwg.h zkwg_prover_create_zkey): rows of A and B
        from section 4, C.w = A.w o B.w, bases from sections 5-9.  prove_batch / prove_records only (the stage-by-stage Python path needs
        the key's bases as tensors: ProvingKey)."""
        self = object.__new__(cls)
        self.c, self.key, self.lib = circuit, None, _lib.load()
        self._zkey = bytes(zkey)
        self._single, self.last_sums = False, None
        h = C.c_void_p()
        _check(self.lib.zkwg_prover_create_zkey(circuit.h, circuit.device, self._zkey, len(self._zkey), slots, C.byref(h)))
        self._h, self._slots_n = h, slots
        self.m = circuit.abc_bytes // 96
        return self

    def __del__(self):
        if getattr(self, "_h", None):
            self.lib.zkwg_prover_destroy(self._h)
            self._h = None

    def _plans(self):
        """the Python-side plans of prove_prepared (built on first use)"""
        if self._single:
            return
        import torch
        circuit, key = self.c, self.key
        self.ntt = Ntt(key.power, device=circuit.device)
        d = circuit.device
        ww = 13 if key.n_wires >= (1 << 16) else 0           # as zkwg_prover_create plans them
        self.msm_a = _DeviceMsm(key.d_a, 1, d, ww, 16)
        self.msm_b1 = _DeviceMsm(key.d_b1, 1, d, ww, 16)
        self.msm_b2 = _DeviceMsm(key.d_b2, 2, d, ww, 16)
        self.msm_c = _DeviceMsm(key.d_c, 1, d, ww, 16)
        self.msm_h = _DeviceMsm(key.d_h, 1, d, 0, 64 if key.power >= 18 else 16)
        wb = max(m.work_bytes() for m in (self.msm_a, self.msm_b1, self.msm_b2, self.msm_c, self.msm_h))
        self.d_msm_work = torch.empty(wb + 256, dtype=torch.uint8, device=self.dev)
        self.d_msm_work = self.d_msm_work[(-self.d_msm_work.data_ptr()) % 256:]
        self.d_wit = torch.empty(circuit.witness_bytes, dtype=torch.uint8, device=self.dev)
        self.d_abc = torch.empty(circuit.abc_bytes, dtype=torch.uint8, device=self.dev)
        self.d_h = torch.empty(32 << key.power, dtype=torch.uint8, device=self.dev)
        self.d_ntt_work = torch.empty(self.ntt.work_bytes(1), dtype=torch.uint8, device=self.dev)
        self._single = True

    def prove_prepared(self, d_in, n, d_scratch, index, r, s, stream=None):
        """the proof of email `index` of a batch of n prepared by circuit.prepare_device(d_in, n, ...): {'pi_a': (x, y), 'pi_b':
        ((x0, x1), (y0, y1)), 'pi_c': (x, y)} as standard-form integers (snarkjs proof.json without the projective ones)"""
        self._plans()
        c, k = self.c, self.key
        c.expand_device(d_in, n, d_scratch, index, 1, self.d_wit, stream)
        c.expand_abc_device(d_in, n, d_scratch, index, 1, self.d_abc, stream, montgomery=True)
        self.ntt.h_evaluations_device(self.d_abc, c.abc_bytes, self.m, 1, self.d_ntt_work, self.d_h, stream=stream)
        wit = self.d_wit.data_ptr()
        sums = {
            "a": self.msm_a.run(wit, False, True, self.d_msm_work, stream),
            "b1": self.msm_b1.run(wit, False, True, self.d_msm_work, stream),
            "b2": self.msm_b2.run(wit, False, True, self.d_msm_work, stream),
            "c": self.msm_c.run(wit + 32 * (k.n_public + 1), False, True, self.d_msm_work, stream),
            "h": self.msm_h.run(self.d_h.data_ptr(), True, False, self.d_msm_work, stream),
        }
        self.last_sums = sums
        pa, pb, pc = (C.c_uint8 * 64)(), (C.c_uint8 * 128)(), (C.c_uint8 * 64)()
        _check(self.lib.zkwg_groth16_assemble(sums["a"], sums["b1"], sums["b2"], sums["c"], sums["h"], k.alpha1, k.beta1, k.beta2, k.delta1, k.delta2,
                                              int(r % R).to_bytes(32, "little"), int(s % R).to_bytes(32, "little"), pa, pb, pc))
        i = lambda b, j: int.from_bytes(bytes(b)[32 * j:32 * j + 32], "little")
        return {"pi_a": (i(pa, 0), i(pa, 1)), "pi_b": ((i(pb, 0), i(pb, 1)), (i(pb, 2), i(pb, 3))), "pi_c": (i(pc, 0), i(pc, 1))}

    # ---- several proofs in flight: include/zkwg.h zkwg_prover_* ---------------------------------------------------------------------
    def _native(self, slots):
        if self._h is not None and self._slots_n >= slots:
            return self._h
        if self._h is not None:
            self.lib.zkwg_prover_destroy(self._h)
            self._h = None
        if self.key is None:                        # made from a zkey: the same entry point again, with more proofs in flight
            h = C.c_void_p()
            _check(self.lib.zkwg_prover_create_zkey(self.c.h, self.c.device, self._zkey, len(self._zkey), slots, C.byref(h)))
            self._h, self._slots_n = h, slots
            return h
        k = self.key
        ks = _KeyStruct(k.n_wires, k.n_public, k.power, k.d_a.data_ptr(), k.d_b1.data_ptr(), k.d_b2.data_ptr(), k.d_c.data_ptr(), k.d_h.data_ptr(), 1)
        for name, size in (("alpha1", 64), ("beta1", 64), ("beta2", 128), ("delta1", 64), ("delta2", 128)):
            C.memmove(getattr(ks, name), getattr(k, name), size)
        h = C.c_void_p()
        _check(self.lib.zkwg_prover_create(self.c.h, self.c.device, None, 0, self.m, C.byref(ks), slots, C.byref(h)))
        self._h, self._slots_n = h, slots
        return h

    @staticmethod
    def _proofs_from_bytes(raw, count):
        i = lambda o: int.from_bytes(raw[o:o + 32], "little")
        out = []
        for q in range(count):
            o = 256 * q
            out.append({"pi_a": (i(o), i(o + 32)), "pi_b": ((i(o + 64), i(o + 96)), (i(o + 128), i(o + 160))), "pi_c": (i(o + 192), i(o + 224))})
        return out

    def prove_batch_bytes(self, d_in, n, d_scratch, indices, blinding, slots=8):
        """prove_batch without the conversion to integers: 256 bytes per proof (pi_a | pi_b | pi_c, standard form, little-endian)"""
        h = self._native(slots)
        idx = (C.c_uint64 * len(indices))(*indices)
        bl = b"".join(int(r % R).to_bytes(32, "little") + int(s % R).to_bytes(32, "little") for r, s in blinding)
        out = (C.c_uint8 * (256 * len(indices)))()
        if indices:
            _check(self.lib.zkwg_prover_prove_prepared(h, d_in.data_ptr(), n, d_scratch.data_ptr(), idx, len(indices), bl, out))
        return bytes(out)

    def prove_batch(self, d_in, n, d_scratch, indices, blinding, slots=8):
        """proofs of the emails `indices` of a prepared batch (complete: synchronise the preparing stream first); blinding = [(r, s)] per
        email -> list of proof dicts (prove_prepared's form).  `slots` proofs are in flight: 1-3 contexts of ceil(slots / contexts) emails,
        every stage one launch series per context."""
        h = self._native(slots)
        idx = (C.c_uint64 * len(indices))(*indices)
        bl = b"".join(int(r % R).to_bytes(32, "little") + int(s % R).to_bytes(32, "little") for r, s in blinding)
        out = (C.c_uint8 * (256 * len(indices)))()
        _check(self.lib.zkwg_prover_prove_prepared(h, d_in.data_ptr(), n, d_scratch.data_ptr(), idx, len(indices), bl, out))
        return self._proofs_from_bytes(bytes(out), len(indices))

    def prove_records(self, records, blinding, slots=8):
        """inputs -> proofs (zkwg_prover_prove_batch): packed input records on the host -> (status list, proofs; None for a failed email)"""
        n = len(records) // self.c.in_stride
        h = self._native(slots)
        bl = b"".join(int(r % R).to_bytes(32, "little") + int(s % R).to_bytes(32, "little") for r, s in blinding)
        st = (C.c_int32 * n)()
        out = (C.c_uint8 * (256 * n))()
        _check(self.lib.zkwg_prover_prove_batch(h, bytes(records), n, bl, st, out))
        proofs = self._proofs_from_bytes(bytes(out), n)
        return list(st), [p if st[i] == 0 else None for i, p in enumerate(proofs)]

    @staticmethod
    def proof_json(p):
        """snarkjs' proof.json layout (what packages/rust-verifier/src/verifier_utils.rs:20-130 parses)"""
        return {"pi_a": [str(p["pi_a"][0]), str(p["pi_a"][1]), "1"],
                "pi_b": [[str(p["pi_b"][0][0]), str(p["pi_b"][0][1])], [str(p["pi_b"][1][0]), str(p["pi_b"][1][1])], ["1", "0"]],
                "pi_c": [str(p["pi_c"][0]), str(p["pi_c"][1]), "1"], "protocol": "groth16", "curve": "bn128"}


def point_from_montgomery(raw):
    """a sum as zkwg_msm_* returns it (affine, Montgomery form) -> standard-form integers: (x, y), ((x0, x1), (y0, y1)) or None"""
    if raw == bytes(len(raw)):
        return None
    rinv = pow(1 << 256, -1, Q)
    v = [int.from_bytes(raw[32 * j:32 * j + 32], "little") * rinv % Q for j in range(len(raw) // 32)]
    return (v[0], v[1]) if len(v) == 2 else ((v[0], v[1]), (v[2], v[3]))
