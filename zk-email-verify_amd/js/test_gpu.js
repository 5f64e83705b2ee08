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
  console.log('js gpu ok');
})().catch((e) => { console.error(e); process.exit(1); });
