#!/usr/bin/env python3
"""Times `checkConstraints` on the device for the production circuit: EmailVerifier(1024,1536) witnesses from the
witness kernels against the complete constraint system exported by zkwg.r1cs (1.8 M constraints)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "zk-email-verify_amd", "py"))
import torch  # noqa: E402
import zkwg  # noqa: E402
from zkwg import r1cs as zr, synth  # noqa: E402

N, M, n = 1024, 1536, int(sys.argv[1]) if len(sys.argv) > 1 else 64
c = zkwg.Circuit(zkwg.MAIN_EMAIL_VERIFIER, max_header=N, max_body=M, device=0)
t0 = time.time()
data = zr.email_verifier_r1cs(c.symbols(), N, M)
t1 = time.time()
r = zkwg.R1cs(data, device=0)
t2 = time.time()
recs, _ = synth.packed_batch(c, seed=9, n=n, body_len=1024)
d_in = torch.frombuffer(bytearray(recs), dtype=torch.uint8).to("cuda:0")
d_out = torch.empty(n * c.witness_bytes, dtype=torch.uint8, device="cuda:0")
d_status = torch.zeros(n, dtype=torch.int32, device="cuda:0")
d_scr = torch.empty(c.scratch_bytes(n), dtype=torch.uint8, device="cuda:0")
c.calculate_batch_device(d_in, n, d_out, d_status, d_scr)
torch.cuda.synchronize()
assert d_status.cpu().tolist() == [0] * n
bad = r.first_violations_device(d_out, n, c.witness_bytes)      # warm-up + result
assert bad == [None] * n, bad[:4]
torch.cuda.synchronize()
ts = time.time()
for _ in range(3):
    r.first_violations_device(d_out, n, c.witness_bytes)
torch.cuda.synchronize()
dt = (time.time() - ts) / 3
print(f"r1cs: {r.n_constraints} constraints, {r.n_wires} wires; derive+serialise {t1 - t0:.1f} s, load {t2 - t1:.1f} s")
print(f"checkConstraints of {n} device witnesses: {dt * 1e3:.1f} ms  ({n / dt:.0f} witnesses/s, {n * r.n_constraints / dt / 1e9:.2f} G constraints/s), all satisfied")
