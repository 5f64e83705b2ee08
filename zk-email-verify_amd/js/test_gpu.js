'use strict';
// GPU test of the Node path: Sha256Bytes(640) like packages/circuits/tests/sha.test.ts, plus wtns.calculate.
const assert = require('assert');
const crypto = require('crypto');
const z = require('./zkwg.js');
function shaPad(msg, max) {  // packages/helpers/src/sha-utils.ts:88-111
  const len = Buffer.alloc(8); len.writeBigUInt64BE(BigInt(msg.length * 8));
  let r = Buffer.concat([msg, Buffer.from([0x80])]);
  while ((r.length * 8 + 64) % 512 !== 0) r = Buffer.concat([r, Buffer.from([0])]);
  r = Buffer.concat([r, len]);
  const n = r.length;
  return [Buffer.concat([r, Buffer.alloc(max - n)]), n];
}
(async () => {
  const c = new z.Circuit({ mainKind: z.MAIN_SHA256_BYTES, maxHeader: 640, maxBody: 0 }, 0);
  const wc = new z.WitnessCalculator(c);
  for (const m of ['0', 'hello world', '']) {
    const [p, n] = shaPad(Buffer.from(m), 640);
    const w = await wc.calculateWitness({ paddedIn: Array.from(p).map(String), paddedInLength: String(n) });
    assert.strictEqual(w[0], 1n);
    let bits = '';
    for (let i = 1; i <= 256; i++) bits += w[i].toString();
    const dig = BigInt('0b' + bits).toString(16).padStart(64, '0');
    assert.strictEqual(dig, crypto.createHash('sha256').update(m).digest('hex'));
  }
  const [p, n] = shaPad(Buffer.from('abc'), 640);
  const mem = { type: 'mem' };
  await z.wtns.calculate({ paddedIn: Array.from(p), paddedInLength: n }, c, mem);
  assert.strictEqual(mem.data.slice(0, 4).toString(), 'wtns');
  assert.strictEqual(mem.data.length, c.wtnsSize);
  await assert.rejects(wc.calculateWitness({ paddedIn: Array.from(p), paddedInLength: n + 1 }), /Assert Failed/);
  const b = await wc.calculateBatch([{ paddedIn: Array.from(p), paddedInLength: n }, { paddedIn: Array.from(p), paddedInLength: 0 }]);
  assert.deepStrictEqual(Array.from(b.status), [0, 4]);
  // checkConstraints on a real device witness: booleanity of the 256 output bits of Sha256Bytes(640)
  // written as an .r1cs file (out_i * (out_i - 1) = 0 over wires 1..256 of the W-wire witness)
  {
    const le = (v, k) => { const bb = Buffer.alloc(k); let x = BigInt(v); for (let i = 0; i < k; ++i) { bb[i] = Number(x & 0xffn); x >>= 8n; } return bb; };
    const lc = (terms) => Buffer.concat([le(terms.length, 4)].concat(terms.map(([w, cf]) => Buffer.concat([le(w, 4), le(cf, 32)]))));
    const W = c.witnessLen, M1 = z.FIELD_MODULUS - 1n;
    const cons = [];
    for (let i = 1; i <= 256; ++i) cons.push(lc([[i, 1]]), lc([[0, M1], [i, 1]]), lc([]));
    const hdr = Buffer.concat([le(32, 4), le(z.FIELD_MODULUS, 32), le(W, 4), le(256, 4), le(641, 4), le(0, 4), le(W, 8), le(256, 4)]);
    const sec = (t, d) => Buffer.concat([le(t, 4), le(d.length, 8), d]);
    const r1 = new z.R1cs(Buffer.concat([Buffer.from('r1cs'), le(1, 4), le(2, 4), sec(2, Buffer.concat(cons)), sec(1, hdr)]), 0);
    const w = await wc.calculateWitness({ paddedIn: Array.from(p), paddedInLength: n });
    await r1.checkConstraints(w);
    const bad = w.slice(); bad[7] = 2n;
    await assert.rejects(r1.checkConstraints(bad), /Constraint doesn't match \(constraint 6\)/);
  }
  // ---- EmailVerifier(576,192,121,17,0,0,0,0): the north_star host path Node -> N-API -> C-ABI -> HIP.
  // Mirrors packages/circuits/tests/email-verifier.test.ts:43 (calculateWitness), :61-79 (tampered input
  // -> "Assert Failed"); the witness is pinned by the SHA-256 of the literal oracle's witness
  // (tests/golden/ev_576_192_case.json, made by tests/golden/make_ev_case.py).
  {
    const fs = require('fs');
    const path = require('path');
    const kase = JSON.parse(fs.readFileSync(path.join(__dirname, '..', '..', 'tests', 'golden', 'ev_576_192_case.json')));
    const ev = new z.Circuit({ mainKind: z.MAIN_EMAIL_VERIFIER, maxHeader: kase.maxHeader, maxBody: kase.maxBody }, 0);
    assert.strictEqual(ev.witnessLen, kase.witnessLen);
    const evc = new z.WitnessCalculator(ev);
    const w = await evc.calculateWitness(kase.input, true);
    assert.strictEqual(w.length, kase.witnessLen);
    assert.strictEqual(w[0], 1n);
    assert.strictEqual(w[1].toString(), kase.pubkeyHash);   // assertOut(witness, {pubkeyHash}) of :188-207
    assert.strictEqual(w[2].toString(), kase.shaHi);
    assert.strictEqual(w[3].toString(), kase.shaLo);
    const bin = await evc.calculateBinWitness(kase.input);
    assert.strictEqual(crypto.createHash('sha256').update(bin).digest('hex'), kase.witnessSha256);
    const wt = await evc.calculateWTNSBin(kase.input);
    assert.strictEqual(Buffer.from(wt.slice(0, 4)).toString(), 'wtns');
    assert.strictEqual(wt.length, ev.wtnsSize);
    // email-verifier.test.ts:61-79: an invalid (tampered) header byte must throw "Assert Failed"
    const tampered = Object.assign({}, kase.input, { emailHeader: kase.input.emailHeader.slice() });
    tampered.emailHeader[10] = String(Number(tampered.emailHeader[10]) ^ 1);
    await assert.rejects(evc.calculateWitness(tampered), /Assert Failed/);
    // :81-102 style: non-zero byte in the padding after emailHeaderLength
    const padded = Object.assign({}, kase.input, { emailHeader: kase.input.emailHeader.slice() });
    padded.emailHeader[kase.maxHeader - 1] = '1';
    await assert.rejects(evc.calculateWitness(padded), /Assert Failed/);
    // generic input path: values that are not bytes / do not fit the packed record reach the circuit's own
    // range checks (Num2Bits(8) lib/sha.circom:27; Num2Bits(10) email-verifier.circom:58) -> "Assert Failed"
    const big = Object.assign({}, kase.input, { emailHeader: kase.input.emailHeader.slice() });
    big.emailHeader[0] = '256';
    await assert.rejects(evc.calculateWitness(big), /Assert Failed/);
    const neg = Object.assign({}, kase.input, { emailHeaderLength: (z.FIELD_MODULUS - 1n).toString() });
    await assert.rejects(evc.calculateWitness(neg), /Assert Failed/);
    // batch: one bad email never aborts the others
    const b3 = await evc.calculateBatch([kase.input, tampered, big, kase.input]);
    assert.deepStrictEqual(Array.from(b3.status), [0, 4, 4, 0]);
    assert.ok(Buffer.from(b3.wtns[0]).equals(Buffer.from(bin)) && Buffer.from(b3.wtns[3]).equals(Buffer.from(bin)));
    // the same batch with the expansion on the host (zkwg_set_host_expand): identical bytes and statuses
    ev.setHostExpand(4);
    const b4 = await evc.calculateBatch([kase.input, tampered, big, kase.input]);
    ev.setHostExpand(0);
    assert.deepStrictEqual(Array.from(b4.status), [0, 4, 4, 0]);
    assert.ok(Buffer.from(b4.wtns[0]).equals(Buffer.from(bin)) && Buffer.from(b4.wtns[3]).equals(Buffer.from(bin)));
    // wtnsFromBin (JS-side container) == the C ABI's zkwg_write_wtns
    assert.ok(ev.wtnsFromBin(Buffer.from(bin)).equals(Buffer.from(wt)));
    // the documented CLI (docs/zk-email-docs/UsageGuide/README.md:132-140): node generate_witness.js <circuit> input.json witness.wtns
    {
      const os = require('os');
      const cp = require('child_process');
      const dir = fs.mkdtempSync(path.join(os.tmpdir(), 'zkwg-'));
      const cli = path.join(__dirname, 'generate_witness.js');
      const spec = `EmailVerifier(${kase.maxHeader},${kase.maxBody},121,17,0,0,0,0)`;
      fs.writeFileSync(path.join(dir, 'input.json'), JSON.stringify(kase.input));
      cp.execFileSync(process.execPath, [cli, spec, path.join(dir, 'input.json'), path.join(dir, 'witness.wtns')]);
      assert.ok(fs.readFileSync(path.join(dir, 'witness.wtns')).equals(Buffer.from(wt)));
      fs.writeFileSync(path.join(dir, 'bad.json'), JSON.stringify(tampered));
      const r1 = cp.spawnSync(process.execPath, [cli, spec, path.join(dir, 'bad.json'), path.join(dir, 'bad.wtns')], { encoding: 'utf8' });
      assert.strictEqual(r1.status, 1);
      assert.ok(/Assert Failed/.test(r1.stderr));
      fs.writeFileSync(path.join(dir, 'batch.json'), JSON.stringify([kase.input, tampered, kase.input]));
      const r2 = cp.spawnSync(process.execPath, [cli, JSON.stringify({ mainKind: z.MAIN_EMAIL_VERIFIER, maxHeader: kase.maxHeader, maxBody: kase.maxBody }), path.join(dir, 'batch.json'), path.join(dir, 'w.wtns')], { encoding: 'utf8' });
      assert.strictEqual(r2.status, 1);
      assert.ok(/email 1: Error: Assert Failed/.test(r2.stderr));
      assert.ok(fs.readFileSync(path.join(dir, 'w_0.wtns')).equals(Buffer.from(wt)) && fs.readFileSync(path.join(dir, 'w_2.wtns')).equals(Buffer.from(wt)));
      assert.ok(!fs.existsSync(path.join(dir, 'w_1.wtns')));
    }
    // circom_tester surface (email-verifier.test.ts:188-207): assertOut(witness, {pubkeyHash}) via loadSymbols
    const t = await z.tester({ mainKind: z.MAIN_EMAIL_VERIFIER, maxHeader: kase.maxHeader, maxBody: kase.maxBody }, 0);
    const tw = await t.calculateWitness(kase.input);
    await t.assertOut(tw, { pubkeyHash: kase.pubkeyHash, shaHi: kase.shaHi, shaLo: BigInt(kase.shaLo) });
    await t.assertOut(tw, { pubkey: kase.input.pubkey });           // arrays element-wise
    await assert.rejects(t.assertOut(tw, { shaHi: '1' }), /main\.shaHi: expected 1/);
    await assert.rejects(t.assertOut(tw, { nothing: 0 }), /Output variable not defined: main\.nothing/);
    await assert.rejects(t.checkConstraints(tw), /no constraint system loaded/);
    assert.strictEqual(t.symbols['main.pubkeyHash'].varIdx, 1);
    // multi-GPU entry (zkwg_calculate_batch_multi) with one device: same witnesses + the gathered result table
    const mc = new z.MultiCalculator({ mainKind: z.MAIN_EMAIL_VERIFIER, maxHeader: kase.maxHeader, maxBody: kase.maxBody }, [0]);
    assert.strictEqual(mc.nDevices, 1);
    const m3 = await mc.calculateBatch([kase.input, tampered, kase.input]);
    assert.deepStrictEqual(Array.from(m3.status), [0, 4, 0]);
    assert.ok(Buffer.from(m3.wtns[0]).equals(Buffer.from(bin)) && Buffer.from(m3.wtns[2]).equals(Buffer.from(bin)));
    assert.deepStrictEqual(m3.table.map((t) => t.status), [0, 4, 0]);
    for (const i of [0, 2]) {
      assert.strictEqual(m3.table[i].pubkeyHash.toString(), kase.pubkeyHash);
      assert.strictEqual(m3.table[i].shaHi.toString(), kase.shaHi);
      assert.strictEqual(m3.table[i].shaLo.toString(), kase.shaLo);
    }
  }
  // ---- BodyHashRegex compiled from a template file (zkwg_circuit_create_regex): the stand-in for zk-regex's
  // body_hash_regex.circom describes the same circuit as the built-in one, so every signal must agree by name
  {
    const fs = require('fs');
    const path = require('path');
    const kase = JSON.parse(fs.readFileSync(path.join(__dirname, '..', '..', 'tests', 'golden', 'ev_576_192_case.json')));
    const tmpl = path.join(__dirname, '..', 'data', 'templates', 'zk-regex-circom', 'circuits', 'common', 'body_hash_regex.circom');
    const ev0 = new z.Circuit({ mainKind: z.MAIN_EMAIL_VERIFIER, maxHeader: kase.maxHeader, maxBody: kase.maxBody }, 0);
    const ev1 = new z.Circuit({ mainKind: z.MAIN_EMAIL_VERIFIER, maxHeader: kase.maxHeader, maxBody: kase.maxBody, regex: tmpl }, 0);
    assert.strictEqual(ev1.witnessLen, ev0.witnessLen);
    const w0 = await new z.WitnessCalculator(ev0).calculateWitness(kase.input);
    const w1 = await new z.WitnessCalculator(ev1).calculateWitness(kase.input);
    const s0 = z.symbols(ev0), s1 = z.symbols(ev1);   // names[slot]
    const byName = new Map();
    for (let i = 1; i < s0.length; ++i) byName.set(s0[i], w0[i]);
    assert.strictEqual(s1.length, kase.witnessLen);
    for (let i = 1; i < s1.length; ++i) assert.strictEqual(w1[i], byName.get(s1[i]), s1[i]);
    assert.throws(() => new z.Circuit({ mainKind: z.MAIN_EMAIL_VERIFIER, maxHeader: kase.maxHeader, maxBody: kase.maxBody, regex: tmpl + '.missing' }, 0), /regex template/);
  }
  {
    // packages/circuits/tests/fp-mul.test.ts:34-46, the reference test as written: FpMul(2,4), 17 * 20 mod 85 = 0
    const t = await z.tester({ mainKind: z.MAIN_FP_MUL, maxHeader: 0, maxBody: 0, n: 2, k: 4 }, 0);
    const w = await t.calculateWitness({ a: [1, 0, 1, 0], b: [0, 1, 1, 0], p: [1, 1, 1, 1] });
    await t.assertOut(w, { out: [0, 0, 0, 0] });
    const w2 = await t.calculateWitness({ a: [3, 1, 0, 0], b: [2, 2, 0, 0], p: [1, 3, 1, 0] });   // 7 * 10 mod 29 = 12
    await t.assertOut(w2, { out: [0, 3, 0, 0] });
    await assert.rejects(t.calculateWitness({ a: [1, 0, 0, 0], b: [1, 0, 0, 0], p: [0, 0, 0, 0] }), /Assert Failed/);
  }
  console.log('js gpu ok');
})().catch((e) => { console.error(e); process.exit(1); });
