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
  console.log('js gpu ok');
})().catch((e) => { console.error(e); process.exit(1); });
