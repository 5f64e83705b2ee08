#!/usr/bin/env node
'use strict';
// Counterpart of circom's generated `generate_witness.js` as the reference documents it
// (docs/zk-email-docs/UsageGuide/README.md:132-140:  node generate_witness.js circuit.wasm input.json witness.wtns):
//
//     node generate_witness.js <circuit> <input.json> <witness.wtns> [device]
//
// <circuit> names the circuit by its template parameters instead of a WASM file: either a JSON file / inline JSON
// with the options of zkwg.Circuit ({"maxHeader":1024,"maxBody":1536,...,"sym":"path.sym","r1cs":"path.r1cs",
// "regex":"body_hash_regex.circom"}) or the shorthand  EmailVerifier(1024,1536,121,17,0,0,0,0).
// <input.json> holds one CircuitInput object (generateEmailVerifierInputs' output) or an array of them; for an array
// the files are <witness>_<i>.wtns and the exit code is 1 if any email failed ("Assert Failed" is printed per email).
const fs = require('fs');
const path = require('path');
const z = require('./zkwg.js');

function circuitOptions(arg) {
  const m = /^EmailVerifier\(([\d\s,]+)\)$/.exec(arg.trim());
  if (m) {
    const p = m[1].split(',').map((x) => Number(x.trim()));
    if (p.length !== 8) throw new Error('EmailVerifier takes 8 parameters');
    return { mainKind: z.MAIN_EMAIL_VERIFIER, maxHeader: p[0], maxBody: p[1], n: p[2], k: p[3], ignoreBodyHashCheck: p[4], enableHeaderMasking: p[5], enableBodyMasking: p[6], removeSoftLineBreaks: p[7] };
  }
  const f = /^FpMul\(\s*(\d+)\s*,\s*(\d+)\s*\)$/.exec(arg.trim());   // tests/test-circuits/fp-mul-test.circom: FpMul(2, 4)
  if (f) return { mainKind: z.MAIN_FP_MUL, maxHeader: 0, maxBody: 0, n: Number(f[1]), k: Number(f[2]) };
  const text = fs.existsSync(arg) ? fs.readFileSync(arg, 'utf8') : arg;
  const o = JSON.parse(text);
  const base = fs.existsSync(arg) ? path.dirname(arg) : '.';
  if (typeof o.sym === 'string' && fs.existsSync(path.resolve(base, o.sym))) o.sym = fs.readFileSync(path.resolve(base, o.sym), 'utf8');
  if (typeof o.r1cs === 'string') o.r1cs = fs.readFileSync(path.resolve(base, o.r1cs));
  if (typeof o.regex === 'string') o.regex = path.resolve(base, o.regex);
  return o;
}

async function main() {
  const a = process.argv.slice(2);
  if (a.length < 3) {
    console.error('Usage: node generate_witness.js <circuit options | EmailVerifier(..8 params..)> <input.json> <witness.wtns> [device]');
    process.exit(2);
  }
  const circuit = new z.Circuit(circuitOptions(a[0]), a[3] === undefined ? 0 : Number(a[3]));
  const input = JSON.parse(fs.readFileSync(a[1], 'utf8'));
  if (!Array.isArray(input)) {
    await z.wtns.calculate(input, circuit, a[2]);       // throws "Assert Failed" like circom_runtime
    return 0;
  }
  const wc = new z.WitnessCalculator(circuit);
  const r = await wc.calculateBatch(input);
  const stem = a[2].replace(/\.wtns$/, '');
  let failed = 0;
  for (let i = 0; i < input.length; ++i) {
    if (r.status[i] !== 0) { console.error(`email ${i}: Error: Assert Failed (status ${r.status[i]})`); ++failed; continue; }
    fs.writeFileSync(`${stem}_${i}.wtns`, circuit.wtnsFromBin(r.wtns[i]));
  }
  return failed ? 1 : 0;
}
main().then((rc) => process.exit(rc), (e) => { console.error(String(e && e.message ? 'Error: ' + e.message : e)); process.exit(1); });
