'use strict';
/*
 * zkwg.js -- Node host side of the MI355X-native batched witness generator.
 *
 * Keeps the API surface the reference calls for this path:
 *   - circom_runtime's WitnessCalculator: calculateWitness / calculateBinWitness /
 *     calculateWTNSBin (packages/circuits/tests/email-verifier.test.ts:43 via circom_tester,
 *     and inside snarkjs.groth16.fullProve, packages/helpers/src/chunked-zkey.ts:80-84);
 *   - snarkjs `wtns.calculate(input, wasm, wtnsFileName | {type: "mem"})`;
 *   - the `CircuitInput` object of packages/helpers/src/input-generators.ts:6-18 is accepted as is.
 * Everything goes through the N-API addon -> C-ABI (include/zkwg.h) -> HIP kernels.
 * Plain JS for node >= 12.22 (no optional chaining / nullish coalescing).
 */
const fs = require('fs');
const path = require('path');
const addon = require(path.join(__dirname, 'zkwg_addon.node'));

const FIELD_MODULUS = 21888242871839275222246405745257275088548364400416034343698204186575808495617n;
const BASE_FIELD_MODULUS = 21888242871839275222246405745257275088696311157297823662689037894645226208583n;   // BN254 Fq (a .zkey states both primes)
const MAIN_EMAIL_VERIFIER = 0, MAIN_SHA256_BYTES = 1, MAIN_RSA_VERIFIER = 2, MAIN_FP_MUL = 3;
const IN = { HEADER: 0, BODY: 1, PRECOMPUTED_SHA: 2, PUBKEY: 3, SIGNATURE: 4, MESSAGE: 5, HEADER_LEN: 6, BODY_LEN: 7, BODY_HASH_INDEX: 8, HEADER_MASK: 9, BODY_MASK: 10, DECODED_BODY: 11, RANGE_FLAGS: 12 };

function norm(v) {
  let x = BigInt(v) % FIELD_MODULUS;
  if (x < 0n) x += FIELD_MODULUS;
  return x;
}

class Circuit {
  /** opts: {mainKind, maxHeader, maxBody, n, k, ignoreBodyHashCheck, enableHeaderMasking, enableBodyMasking, removeSoftLineBreaks, sym, symAlias}
   *  sym: text of the compiled circuit's `.sym` file -> the witness follows its indices (zkwg_circuit_create_sym);
   *  r1cs: Buffer with its `.r1cs` -> complete witness of an --O0 / --O1 build (zkwg_circuit_create_full); device < 0 = layout-only handle;
   *  regex (+ regexIncludeDirs, regexTemplate): path of a zk-regex style `body_hash_regex.circom` -> BodyHashRegex is compiled
   *  from the template text instead of zkwg's built-in circuit (zkwg_circuit_create_regex) */
  constructor(opts, device) {
    this.opts = Object.assign({ mainKind: MAIN_EMAIL_VERIFIER, maxHeader: 1024, maxBody: 1536, n: 121, k: 17, ignoreBodyHashCheck: 0, enableHeaderMasking: 0, enableBodyMasking: 0, removeSoftLineBreaks: 0 }, opts || {});
    this.device = device === undefined ? 0 : device;
    this.handle = addon.createCircuit(this.opts, this.device);
    Object.assign(this, addon.info(this.handle));
  }

  /** threads > 0: batches delivered to host memory are expanded BY the host from the 0.45 MB image each email's compute
   *  kernels leave (zkwg_set_host_expand) -- what a host-side snarkjs prover (chunked-zkey.ts:80-84) consumes, without
   *  pushing 56.9 MB per email through PCIe; 0 (default): expanded on the device, then copied */
  setHostExpand(threads) { addon.setHostExpand(this.handle, threads | 0); }

  /** 32-byte-per-signal witness -> `.wtns` file bytes (snarkjs wtns v2: header section with n8 = 32, the prime, the
   *  witness length; then the values, little-endian, non-Montgomery) */
  wtnsFromBin(bin) {
    const W = this.witnessLen;
    if (bin.length !== 32 * W) throw new Error('zkwg: witness buffer has the wrong length');
    const head = Buffer.alloc(12 + 12 + 40 + 12);
    head.write('wtns', 0, 'latin1'); head.writeUInt32LE(2, 4); head.writeUInt32LE(2, 8);
    head.writeUInt32LE(1, 12); head.writeBigUInt64LE(40n, 16);
    head.writeUInt32LE(32, 24);
    let p = FIELD_MODULUS;
    for (let i = 0; i < 32; ++i) { head[28 + i] = Number(p & 0xffn); p >>= 8n; }
    head.writeUInt32LE(W, 60);
    head.writeUInt32LE(2, 64); head.writeBigUInt64LE(BigInt(32 * W), 68);
    return Buffer.concat([head, Buffer.from(bin)]);
  }

  signalSizes() {
    const o = this.opts;
    if (o.mainKind === MAIN_SHA256_BYTES) return { paddedIn: o.maxHeader, paddedInLength: 1 };
    if (o.mainKind === MAIN_RSA_VERIFIER) return { message: o.k, signature: o.k, modulus: o.k };
    if (o.mainKind === MAIN_FP_MUL) return { a: o.k, b: o.k, p: o.k };   // FpMul(n, k) (tests/test-circuits/fp-mul-test.circom)
    const s = { emailHeader: o.maxHeader, emailHeaderLength: 1, pubkey: o.k, signature: o.k };
    if (o.enableHeaderMasking) s.headerMask = o.maxHeader;
    if (!o.ignoreBodyHashCheck) {
      Object.assign(s, { bodyHashIndex: 1, precomputedSHA: 32, emailBody: o.maxBody, emailBodyLength: 1 });
      if (o.removeSoftLineBreaks) s.decodedEmailBodyIn = o.maxBody;
      if (o.enableBodyMasking) s.bodyMask = o.maxBody;
    }
    return s;
  }

  /** CircuitInput -> packed record (Buffer).  Error texts follow circom_runtime. */
  pack(input) {
    const sizes = this.signalSizes();
    const flat = {};
    let nset = 0;
    for (const key of Object.keys(input)) {
      if (!(key in sizes)) throw new Error('Signal not found: ' + key);
      const vals = (Array.isArray(input[key]) ? input[key] : [input[key]]).map(norm);
      if (vals.length > sizes[key]) throw new Error('Too many values for input signal ' + key);
      if (vals.length < sizes[key]) throw new Error('Not enough values for input signal ' + key);
      flat[key] = vals;
      nset++;
    }
    const want = Object.keys(sizes).length;
    if (nset !== want) throw new Error('Not all inputs have been set. Only ' + nset + ' out of ' + want);
    const rec = Buffer.alloc(this.inputStride);
    const off = this.offsets;
    const handle = this.handle;
    // A value that does not fit its packed slot goes through the generic 32-byte-per-signal path
    // (addon.packField -> zkwg_pack_field): the record keeps the low bits and a range flag, and the
    // circuit's own range check of that signal then fails the email with "Assert Failed" -- what
    // circom_runtime does for the same input (SURVEY.md 8b2/8b3).
    const generic = (field, vals) => {
      const buf = Buffer.alloc(32 * vals.length);
      vals.forEach((v, i) => { let x = v; for (let k = 0; k < 4; ++k) { buf.writeBigUInt64LE(x & 0xffffffffffffffffn, 32 * i + 8 * k); x >>= 64n; } });
      addon.packField(handle, rec, field, 0, buf);
    };
    const bytes = (field, vals, what) => {
      if (vals.some((v) => v > 255n)) return generic(field, vals);
      vals.forEach((v, i) => { rec[off[field] + i] = Number(v); });
    };
    const u32 = (field, v, what) => {
      if (v >> 32n) return generic(field, [v]);
      rec.writeUInt32LE(Number(v), off[field]);
    };
    const limbs = (field, vals, what) => {
      if (vals.some((v) => (v >> 128n) !== 0n)) return generic(field, vals);
      vals.forEach((v, i) => {
        rec.writeBigUInt64LE(v & 0xffffffffffffffffn, off[field] + 16 * i);
        rec.writeBigUInt64LE(v >> 64n, off[field] + 16 * i + 8);
      });
    };
    const o = this.opts;
    if (o.mainKind === MAIN_SHA256_BYTES) {
      bytes(IN.HEADER, flat.paddedIn, 'paddedIn'); u32(IN.HEADER_LEN, flat.paddedInLength[0], 'paddedInLength');
    } else if (o.mainKind === MAIN_RSA_VERIFIER) {
      limbs(IN.MESSAGE, flat.message, 'message'); limbs(IN.SIGNATURE, flat.signature, 'signature'); limbs(IN.PUBKEY, flat.modulus, 'modulus');
    } else if (o.mainKind === MAIN_FP_MUL) {
      // the chunks of a, b, p travel in the pubkey / signature / message slots of the record (include/zkwg.h)
      limbs(IN.PUBKEY, flat.a, 'a'); limbs(IN.SIGNATURE, flat.b, 'b'); limbs(IN.MESSAGE, flat.p, 'p');
    } else {
      bytes(IN.HEADER, flat.emailHeader, 'emailHeader'); u32(IN.HEADER_LEN, flat.emailHeaderLength[0], 'emailHeaderLength');
      limbs(IN.PUBKEY, flat.pubkey, 'pubkey'); limbs(IN.SIGNATURE, flat.signature, 'signature');
      if (o.enableHeaderMasking) bytes(IN.HEADER_MASK, flat.headerMask, 'headerMask');
      if (!o.ignoreBodyHashCheck && o.enableBodyMasking) bytes(IN.BODY_MASK, flat.bodyMask, 'bodyMask');
      if (!o.ignoreBodyHashCheck && o.removeSoftLineBreaks) bytes(IN.DECODED_BODY, flat.decodedEmailBodyIn, 'decodedEmailBodyIn');
      if (!o.ignoreBodyHashCheck) {
        bytes(IN.BODY, flat.emailBody, 'emailBody'); u32(IN.BODY_LEN, flat.emailBodyLength[0], 'emailBodyLength');
        bytes(IN.PRECOMPUTED_SHA, flat.precomputedSHA, 'precomputedSHA'); u32(IN.BODY_HASH_INDEX, flat.bodyHashIndex[0], 'bodyHashIndex');
      }
    }
    return rec;
  }
}

class WitnessCalculator {
  constructor(circuit) { this.circuit = circuit; }

  async _one(input, asWtns) {
    const rec = this.circuit.pack(input);
    const r = await addon.calculateBatch(this.circuit.handle, rec, true, !!asWtns);
    if (r.status[0] !== 0) throw new Error(addon.strerror(r.status[0]));
    return r.witness;
  }
  /** -> Promise<bigint[]>  (circom_runtime calculateWitness).  `sanityCheck` is accepted for signature
   * compatibility: circom_runtime uses it to turn the WASM's `assert` instructions on; here every
   * `===` / assert of the circuit is always evaluated (a failing email always reports "Assert Failed"). */
  async calculateWitness(input, sanityCheck) { return addon.witnessToBigInts(await this._one(input, false)); }
  /** -> Promise<Uint8Array> of 32*W bytes (circom_runtime calculateBinWitness) */
  async calculateBinWitness(input, sanityCheck) { return this._one(input, false); }
  /** -> Promise<Uint8Array> .wtns file bytes (circom_runtime calculateWTNSBin) */
  async calculateWTNSBin(input, sanityCheck) { return this._one(input, true); }
  /** inputs[] -> Promise<{wtns: Buffer[], status: Int32Array}>; never throws for a failed email */
  async calculateBatch(inputs, wantWitness) {
    const recs = Buffer.concat(inputs.map((i) => this.circuit.pack(i)));
    const r = await addon.calculateBatch(this.circuit.handle, recs, wantWitness !== false, false);
    const wb = this.circuit.witnessBytes;
    const wtns = r.witness ? inputs.map((_, i) => r.witness.slice(i * wb, (i + 1) * wb)) : [];
    return { wtns, status: r.status };
  }
}

/** The second half of `snarkjs.groth16.fullProve` (packages/helpers/src/chunked-zkey.ts:80-84) on the device: `groth16.prove(zkey,
 * wtns)` for whole batches -- witness, A.w | B.w | C.w, H evaluations, the five multi-exponentiations over the zkey's bases, proof
 * assembly (include/zkwg.h zkwg_prover_*).
 *     new Prover(circuit, zkey, slots)                 what groth16.prove takes: the zkey ALONE (rows of A and B from its section 4, C.w =
 *                                                      A.w o B.w as buildABC1 forms it, bases from sections 5-9; zkwg_prover_create_zkey)
 *     new Prover(circuit, r1cs, nRows, zkey, slots)    the constraint system from an .r1cs over the circuit's witness layout WITH the
 *                                                      nPublic + 1 rows snarkjs appends to A (nRows = its constraint count); the zkey's
 *                                                      sections 5-9 only
 * `zkey`: the bytes of a snarkjs groth16 .zkey (layout restated from snarkjs, DESIGN.md section 23).  `slots` proofs are in flight (1-3
 * contexts of E emails each, every stage one launch series per context). */
class Prover {
  constructor(circuit, a, b, c, d) {
    this.circuit = circuit;
    if (typeof b !== "number" || c === undefined) {           // (circuit, zkey[, slots])
      const key = Prover.parseZkey(a);
      if (key.nWires !== circuit.witnessLen) throw new Error(`zkwg: the key has ${key.nWires} wires, the circuit's witness ${circuit.witnessLen}`);
      this.nPublic = key.nPublic;
      this.handle = addon.createProverZkey(circuit.handle, circuit.device, a, b || 16);
      return;
    }
    const key = Prover.parseZkey(c);                          // (circuit, r1cs, nRows, zkey[, slots])
    if (key.nWires !== circuit.witnessLen) throw new Error(`zkwg: the key has ${key.nWires} wires, the circuit's witness ${circuit.witnessLen}`);
    this.nPublic = key.nPublic;
    this.handle = addon.createProver(circuit.handle, circuit.device, a, b, key, d || 16);
  }
  /** sections of a groth16 .zkey -> the object addon.createProver takes */
  static parseZkey(buf) {
    if (buf.slice(0, 4).toString("latin1") !== "zkey" || buf.readUInt32LE(4) !== 1) throw new Error("zkwg: not a version-1 .zkey file");
    const nsec = buf.readUInt32LE(8), sec = {};
    let pos = 12;
    for (let i = 0; i < nsec; ++i) {
      const id = buf.readUInt32LE(pos), size = Number(buf.readBigUInt64LE(pos + 4));
      if (pos + 12 + size > buf.length) throw new Error(`zkwg: .zkey section ${id} runs past the end of the file`);
      sec[id] = [pos + 12, size];
      pos += 12 + size;
    }
    for (const need of [1, 2, 5, 6, 7, 8, 9]) if (!sec[need]) throw new Error(`zkwg: .zkey section ${need} is missing`);
    if (buf.readUInt32LE(sec[1][0]) !== 1) throw new Error("zkwg: not a groth16 key");
    let p = sec[2][0];
    const le = (o, n) => { let x = 0n; for (let k = n - 1; k >= 0; --k) x = (x << 8n) | BigInt(buf[o + k]); return x; };
    const n8q = buf.readUInt32LE(p);
    if (n8q !== 32 || le(p + 4, 32) !== BASE_FIELD_MODULUS) throw new Error("zkwg: not a BN254 key (base field)");
    p += 4 + n8q;
    const n8r = buf.readUInt32LE(p);
    if (n8r !== 32 || le(p + 4, 32) !== FIELD_MODULUS) throw new Error("zkwg: not a BN254 key (scalar field)");
    p += 4 + n8r;
    const nWires = buf.readUInt32LE(p), nPublic = buf.readUInt32LE(p + 4), domain = buf.readUInt32LE(p + 8);
    p += 12;
    if (domain === 0 || (domain & (domain - 1)) !== 0) throw new Error("zkwg: the .zkey's domain size is not a power of two");
    if (nPublic + 1 >= nWires) throw new Error("zkwg: the .zkey's header is inconsistent (nPublic, nVars)");
    const want = { 5: 64 * nWires, 6: 64 * nWires, 7: 128 * nWires, 8: 64 * (nWires - nPublic - 1), 9: 64 * domain };
    for (const id of [5, 6, 7, 8, 9]) if (sec[id][1] !== want[id]) throw new Error(`zkwg: .zkey section ${id} holds ${sec[id][1]} bytes, expected ${want[id]}`);
    const take = (n) => { const b = buf.slice(p, p + n); p += n; return b; };
    const alpha1 = take(64), beta1 = take(64), beta2 = take(128); take(128); const delta1 = take(64), delta2 = take(128);
    const s = (id) => buf.slice(sec[id][0], sec[id][0] + sec[id][1]);
    return { nWires, nPublic, log2Domain: 31 - Math.clz32(domain), a: s(5), b1: s(6), b2: s(7), c: s(8), h: s(9), alpha1, beta1, beta2, delta1, delta2 };
  }
  /** uniform in Fr, as snarkjs' Fr.random(): 254 random bits, rejected until below the group order (acceptance 0.76) */
  static randomFr() {
    for (;;) {
      const b = require("crypto").randomBytes(32);
      b[0] &= 0x3f;
      const x = BigInt("0x" + b.toString("hex"));
      if (x < FIELD_MODULUS) return x;
    }
  }
  /** inputs[] (generateEmailVerifierInputs objects) -> Promise<{status: Int32Array, proofs: (snarkjs proof.json object | null)[]}>;
   * blinding: optional [[r, s], ...] bigints (default: random) */
  async proveBatch(inputs, blinding) {
    const recs = Buffer.concat(inputs.map((i) => this.circuit.pack(i)));
    const bl = Buffer.alloc(64 * inputs.length);
    const put = (v, off) => { let x = BigInt(v) % FIELD_MODULUS; for (let k = 0; k < 32; ++k) { bl[off + k] = Number(x & 255n); x >>= 8n; } };
    for (let i = 0; i < inputs.length; ++i) {
      const rs = blinding ? blinding[i] : [Prover.randomFr(), Prover.randomFr()];
      put(rs[0], 64 * i); put(rs[1], 64 * i + 32);
    }
    const r = await addon.proveBatch(this.handle, this.circuit.handle, recs, bl);
    const num = (o) => { let x = 0n; for (let k = 31; k >= 0; --k) x = (x << 8n) | BigInt(r.proofs[o + k]); return x.toString(); };
    const proofs = [];
    for (let i = 0; i < inputs.length; ++i) {
      const o = 256 * i;
      proofs.push(r.status[i] !== 0 ? null : {
        pi_a: [num(o), num(o + 32), "1"], pi_b: [[num(o + 64), num(o + 96)], [num(o + 128), num(o + 160)], ["1", "0"]],
        pi_c: [num(o + 192), num(o + 224), "1"], protocol: "groth16", curve: "bn128" });
    }
    return { status: r.status, proofs };
  }
}

/** Batch calculation sharded over several GPUs of one node (include/zkwg.h zkwg_multi_*): contiguous
 * shards, one handle + host thread per GPU, witnesses leave through each GPU's own PCIe link; the
 * 100-byte/email result table {status, pubkeyHash, shaHi, shaLo} is gathered on devices[0] over RCCL. */
class MultiCalculator {
  /** opts as for Circuit; devices: GPU ordinals, e.g. [0,1,2,3,4,5,6,7] */
  constructor(opts, devices) {
    this.circuit = new Circuit(opts, -1);                 // layout-only handle: packing + geometry
    this.handle = addon.createMulti(this.circuit.opts, devices);
    this.nDevices = addon.multiDevices(this.handle);
  }
  /** inputs[] -> Promise<{wtns: Buffer[], status: Int32Array, table: {status, pubkeyHash, shaHi, shaLo}[]}> */
  async calculateBatch(inputs, wantWitness) {
    const recs = Buffer.concat(inputs.map((i) => this.circuit.pack(i)));
    const r = await addon.calculateBatchMulti(this.handle, recs, wantWitness !== false);
    const wb = this.circuit.witnessBytes;
    const wtns = r.witness ? inputs.map((_, i) => r.witness.slice(i * wb, (i + 1) * wb)) : [];
    const le = (b, o) => { let x = 0n; for (let k = 31; k >= 0; --k) x = (x << 8n) | BigInt(b[o + k]); return x; };
    const table = inputs.map((_, i) => ({ status: r.table.readInt32LE(100 * i), pubkeyHash: le(r.table, 100 * i + 4), shaHi: le(r.table, 100 * i + 36), shaLo: le(r.table, 100 * i + 68) }));
    return { wtns, status: r.status, table };
  }
}

/** circom_tester-shaped handle: what `wasm_tester(circuitPath, {include, ...})` returns in the reference's tests
 * (packages/circuits/tests/email-verifier.test.ts:21-31): `calculateWitness(input, sanityCheck)`,
 * `checkConstraints(witness)`, `assertOut(witness, {name: value | array})`, `loadSymbols()` / `.symbols`
 * (`{ "main.x[3]": {labelIdx, varIdx, componentIdx} }`).  `opts` as for Circuit plus `r1csFile` (Buffer with the
 * circuit's `.r1cs`; without it checkConstraints has nothing to check against and throws). */
class Tester {
  constructor(opts, device) {
    this.circuit = new Circuit(opts, device);
    this.wc = new WitnessCalculator(this.circuit);
    this.symbols = null;
    this.constraints = opts && opts.r1csFile ? new R1cs(opts.r1csFile, device) : null;
  }
  async calculateWitness(input, sanityCheck) { return this.wc.calculateWitness(input, sanityCheck); }
  async loadSymbols() {
    if (this.symbols) return;
    this.symbols = {};
    for (const line of addon.symText(this.circuit.handle).split('\n')) {
      if (!line) continue;
      const p = line.split(',');
      this.symbols[p.slice(3).join(',')] = { labelIdx: Number(p[0]), varIdx: Number(p[1]), componentIdx: Number(p[2]) };
    }
  }
  async checkConstraints(witness) {
    if (!this.constraints) throw new Error('zkwg: no constraint system loaded (pass r1csFile)');
    return this.constraints.checkConstraints(witness);
  }
  /** compares `main.<name>` (arrays element-wise) with the witness; throws like circom_tester on the first difference */
  async assertOut(actualOut, expectedOut) {
    await this.loadSymbols();
    const self = this;
    const check = (prefix, e) => {
      if (Array.isArray(e)) { for (let i = 0; i < e.length; i++) check(prefix + '[' + i + ']', e[i]); return; }
      if (typeof e === 'object' && e !== null && e.constructor.name === 'Object') { for (const k of Object.keys(e)) check(prefix + '.' + k, e[k]); return; }
      const sym = self.symbols[prefix];
      if (sym === undefined) throw new Error('Output variable not defined: ' + prefix);
      const got = actualOut[sym.varIdx].toString(), want = e.toString();
      if (got !== want) throw new Error(prefix + ': expected ' + want + ', the witness has ' + got);
    };
    check('main', expectedOut);
  }
}
/** `wasm_tester` counterpart: the reference passes the path of a `.circom` main file; here the circuit is named by
 * its template parameters (opts) */
async function tester(opts, device) { return new Tester(opts, device); }

/** snarkjs-shaped `wtns.calculate(input, circuitOrWasm, wtnsFileName | {type:"mem"})`.  The second
 * argument is a zkwg Circuit (where the reference passes the path of the circom WASM). */
const wtns = {
  async calculate(input, circuit, wtnsFile) {
    const wc = new WitnessCalculator(circuit);
    const bin = await wc.calculateWTNSBin(input);
    if (wtnsFile && typeof wtnsFile === 'object' && wtnsFile.type === 'mem') { wtnsFile.data = bin; return; }
    fs.writeFileSync(wtnsFile, bin);
  },
};

/** The compiled circuit's `.r1cs`, for circom_tester-style `checkConstraints(witness)` on the device
 * (packages/circuits/tests/email-verifier.test.ts:44). */
class R1cs {
  constructor(fileBytes, device) {
    Object.assign(this, addon.r1csLoad(Buffer.from(fileBytes), device === undefined ? 0 : device));
  }
  /** witness: bigint[] (as calculateWitness returns) -> resolves, or throws "Constraint doesn't match" */
  async checkConstraints(witness) {
    if (witness.length !== this.nWires) throw new Error('Invalid witness length. Circuit: ' + this.nWires + ', witness: ' + witness.length);
    const buf = Buffer.alloc(32 * witness.length);
    witness.forEach((v, i) => { let x = BigInt(v); for (let k = 0; k < 4; ++k) { buf.writeBigUInt64LE(x & 0xffffffffffffffffn, 32 * i + 8 * k); x >>= 64n; } });
    const bad = addon.r1csCheck(this.handle, buf, 1, buf.length)[0];
    if (bad >= 0) throw new Error("Constraint doesn't match (constraint " + bad + ')');
  }
  /** n binary witnesses in one Buffer -> first violated constraint per witness (-1 = all hold) */
  firstViolations(buf, n, stride) { return addon.r1csCheck(this.handle, buf, n, stride || 32 * this.nWires); }
}

/** names[slot] of the circuit's layout (circom_tester `loadSymbols` counterpart, email-verifier.test.ts:204) */
function symbols(circuit) {
  const names = [];
  for (const line of addon.symText(circuit.handle).split('\n')) {
    if (!line) continue;
    const p = line.split(',');
    names[Number(p[1])] = p.slice(3).join(',');
  }
  return names;
}

module.exports = { symbols, R1cs, Circuit, WitnessCalculator, MultiCalculator, Prover, Tester, tester, wtns, FIELD_MODULUS, MAIN_EMAIL_VERIFIER, MAIN_SHA256_BYTES, MAIN_RSA_VERIFIER, MAIN_FP_MUL };
