'use strict';
// CPU smoke of the N-API addon + shim (no GPU): layout-only handle, packing, error texts.
const assert = require('assert');
const z = require('./zkwg.js');
const c = new z.Circuit({ mainKind: z.MAIN_SHA256_BYTES, maxHeader: 128, maxBody: 0 }, -1);
assert.strictEqual(c.witnessLen > 0, true);
assert.strictEqual(c.numPublic, 256 + 128 + 1);
const padded = new Array(128).fill(0); padded[0] = 0x80;
const rec = c.pack({ paddedIn: padded.map(String), paddedInLength: '64' });
assert.strictEqual(rec.length, c.inputStride);
assert.strictEqual(rec[c.offsets[0]], 0x80);
assert.strictEqual(rec.readUInt32LE(c.offsets[6]), 64);
assert.throws(() => c.pack({ nope: 1 }), /Signal not found/);
assert.throws(() => c.pack({ paddedIn: padded.concat([1]), paddedInLength: 64 }), /Too many values for input signal paddedIn/);
assert.throws(() => c.pack({ paddedIn: padded.slice(1), paddedInLength: 64 }), /Not enough values for input signal paddedIn/);
assert.throws(() => c.pack({ paddedIn: padded }), /Not all inputs have been set. Only 1 out of 2/);
const ev = new z.Circuit({ maxHeader: 576, maxBody: 192 }, -1);
assert.strictEqual(ev.numPublic, 20);
// `.sym`-ordered witness: swap the two halves of the non-public part of a small circuit's table
{
  const W = c.witnessLen, np = c.numPublic;
  const names = z.symbols(c);
  assert.strictEqual(names.length, W);
  {
    const half = np + 1 + Math.floor((W - np - 1) / 2);
    const dst = (s) => (s <= np ? s : (s < half ? s + (W - half) : s - (half - np - 1)));
    const text = names.slice(1).map((n, i) => (i + 1) + ',' + dst(i + 1) + ',0,' + n).join('\n') + '\n';
    const cs = new z.Circuit({ mainKind: z.MAIN_SHA256_BYTES, maxHeader: 128, maxBody: 0, sym: text }, -1);
    assert.strictEqual(cs.witnessLen, W);
    assert.strictEqual(z.symbols(cs)[dst(np + 1)], names[np + 1]);
    assert.throws(() => new z.Circuit({ mainKind: z.MAIN_SHA256_BYTES, maxHeader: 128, maxBody: 0, sym: text + '5,' + W + ',0,main.nope\n' }, -1), /not produced by this schedule/);
  }
}
// .r1cs reader (parse-only handle): a*b = c over wires [1, a, b, c]
{
  const le = (v, n) => { const b = Buffer.alloc(n); let x = BigInt(v); for (let i = 0; i < n; ++i) { b[i] = Number(x & 0xffn); x >>= 8n; } return b; };
  const lc = (terms) => Buffer.concat([le(terms.length, 4)].concat(terms.map(([w, c]) => Buffer.concat([le(w, 4), le(c, 32)]))));
  const hdr = Buffer.concat([le(32, 4), le(z.FIELD_MODULUS, 32), le(4, 4), le(1, 4), le(0, 4), le(2, 4), le(4, 8), le(1, 4)]);
  const cons = Buffer.concat([lc([[1, 1]]), lc([[2, 1]]), lc([[3, 1]])]);
  const sec = (t, d) => Buffer.concat([le(t, 4), le(d.length, 8), d]);
  const file = Buffer.concat([Buffer.from('r1cs'), le(1, 4), le(2, 4), sec(1, hdr), sec(2, cons)]);
  const r = new z.R1cs(file, -1);
  assert.strictEqual(r.nWires, 4); assert.strictEqual(r.nConstraints, 1); assert.strictEqual(r.nPrvIn, 2);
  assert.throws(() => new z.R1cs(Buffer.from('nope'), -1), /malformed/);
  assert.throws(() => r.firstViolations(Buffer.alloc(128), 1), /no HIP device/);
}
// flag variant removeSoftLineBreaks: one more input signal, same public signals
const qp = new z.Circuit({ maxHeader: 576, maxBody: 384, removeSoftLineBreaks: 1 }, -1);
assert.strictEqual(qp.numPublic, 20);
assert.strictEqual(qp.witnessLen, 881180);
assert.strictEqual('decodedEmailBodyIn' in qp.signalSizes(), true);
new z.WitnessCalculator(c).calculateWitness({ paddedIn: padded, paddedInLength: 64 }).then(
  () => { console.error('expected a no-device error'); process.exit(1); },
  (e) => { assert.ok(/no HIP device/.test(e.message)); console.log('js cpu ok W(sha128)=' + c.witnessLen + ' W(ev576/192)=' + ev.witnessLen); });
// main = FpMul(2, 4) (packages/circuits/tests/test-circuits/fp-mul-test.circom:5): generic small parameters through the addon
{
  const f = new z.Circuit({ mainKind: z.MAIN_FP_MUL, maxHeader: 0, maxBody: 0, n: 2, k: 4 }, -1);
  assert.strictEqual(f.numPublic, 0);
  const names = z.symbols(f);
  assert.deepStrictEqual(names.slice(0, 6), ['one', 'main.out[0]', 'main.out[1]', 'main.out[2]', 'main.out[3]', 'main.a[0]']);
  const r = f.pack({ a: [1, 0, 1, 0], b: [0, 1, 1, 0], p: [1, 1, 1, 1] });   // tests/fp-mul.test.ts:35-39
  assert.strictEqual(r.readUInt32LE(f.offsets[3]), 1);
  assert.strictEqual(r.readUInt32LE(f.offsets[4] + 16), 1);
  assert.strictEqual(r.readUInt32LE(f.offsets[5] + 48), 1);
  assert.throws(() => f.pack({ a: [1, 0, 1], b: [0, 1, 1, 0], p: [1, 1, 1, 1] }), /Not enough values for input signal a/);
  assert.throws(() => new z.Circuit({ mainKind: z.MAIN_FP_MUL, maxHeader: 0, maxBody: 0, n: 121, k: 17 }, -1));
}
