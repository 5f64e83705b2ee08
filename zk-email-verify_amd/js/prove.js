#!/usr/bin/env node
'use strict';
// Counterpart of `snarkjs groth16 fullprove input.json circuit.wasm circuit.zkey proof.json public.json` (what
// packages/helpers/src/chunked-zkey.ts:80-84 does through the API) with witness AND proof computed on the device:
//
//     node prove.js <circuit> <input.json> <circuit.zkey> <proofs.json> [device]
//     node prove.js <circuit> <input.json> <circuit.zkey> <circuit.r1cs> <nRows> <proofs.json> [device]
//
// <circuit> as for generate_witness.js (EmailVerifier(1024,1536,121,17,0,0,0,0) or a JSON of zkwg.Circuit options).  First form: the
// zkey alone, as `fullProve(input, wasm, zkey)` takes it -- its section 4 is the constraint system.  Second form: the system from
// <circuit.r1cs> over the circuit's witness layout with the nPublic + 1 rows snarkjs appends to A (python -m zkwg.r1cs --public-rows 1
// -o circuit.r1cs), <nRows> its constraint count.  <input.json>: one CircuitInput object or an array; <proofs.json> receives
// [{status, proof, publicSignals}] per email.
const fs = require('fs');
const z = require('./zkwg.js');

async function main() {
  const a = process.argv.slice(2);
  if (a.length < 4) { console.error('Usage: node prove.js <circuit> <input.json> <circuit.zkey> [<circuit.r1cs> <nRows>] <proofs.json> [device]'); process.exit(2); }
  const withR1cs = a.length >= 6;
  const outPath = withR1cs ? a[5] : a[3], devArg = withR1cs ? a[6] : a[4];
  const m = /^EmailVerifier\(([\d\s,]+)\)$/.exec(a[0].trim());
  let opts;
  if (m) {
    const p = m[1].split(',').map((x) => Number(x.trim()));
    opts = { mainKind: z.MAIN_EMAIL_VERIFIER, maxHeader: p[0], maxBody: p[1], n: p[2], k: p[3], ignoreBodyHashCheck: p[4], enableHeaderMasking: p[5], enableBodyMasking: p[6], removeSoftLineBreaks: p[7] };
  } else opts = JSON.parse(fs.existsSync(a[0]) ? fs.readFileSync(a[0], 'utf8') : a[0]);
  const circuit = new z.Circuit(opts, devArg === undefined ? 0 : Number(devArg));
  let inputs = JSON.parse(fs.readFileSync(a[1], 'utf8'));
  if (!Array.isArray(inputs)) inputs = [inputs];
  const zkey = fs.readFileSync(a[2]);
  const prover = withR1cs ? new z.Prover(circuit, fs.readFileSync(a[3]), Number(a[4]), zkey, 16) : new z.Prover(circuit, zkey, 16);
  const calc = new z.WitnessCalculator(circuit);
  const r = await prover.proveBatch(inputs);
  // public signals: w[1 .. nPublic] of each witness (groth16.fullProve returns them beside the proof)
  const nPublic = prover.nPublic;
  const out = [];
  for (let i = 0; i < inputs.length; ++i) {
    let publicSignals = null;
    if (r.status[i] === 0) publicSignals = (await calc.calculateWitness(inputs[i])).slice(1, 1 + nPublic).map((x) => x.toString());
    out.push({ status: r.status[i], proof: r.proofs[i], publicSignals });
  }
  fs.writeFileSync(outPath, JSON.stringify(out));
  const bad = out.filter((o) => o.status !== 0).length;
  console.log(`zkwg prove: ${inputs.length - bad} proof(s), ${bad} failed email(s)`);
  process.exit(bad ? 1 : 0);
}
main().catch((e) => { console.error(String(e && e.stack || e)); process.exit(3); });
