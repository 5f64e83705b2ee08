#!/usr/bin/env python3
"""Times the device-side prover (zkwg.prover: witness -> A.w | B.w | C.w -> H evaluations -> five multi-exponentiations -> proof
assembly) per email, stage by stage.  The bases are fixed-base multiples of RANDOM scalars (zkwg_fixed_base_device): timing needs
points of the right shape, not a valid key -- validity is tests/test_prove.py's business (pinned verifier).  Prints one JSON line.

    python tools/bench_prove.py [--max-header 576 --max-body 192] [--emails 8]
"""
import argparse
import json
import os
import random
import sys
import time

# Proofs in flight live on separate HIP streams; the runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (4 by default),
# and a multi-exponentiation's serial tail holds its queue: 16 queues took the same code from 43 to 85 proofs/s
# (profiles/r05/r05_i_bench_prove.json).  Must be set before the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "zk-email-verify_amd", "py"))


def main(argv=None, quiet=False):
    ap = argparse.ArgumentParser()
    ap.add_argument("--max-header", type=int, default=576)
    ap.add_argument("--max-body", type=int, default=192)
    ap.add_argument("--emails", type=int, default=8)
    ap.add_argument("--slots", type=int, default=32, help="proofs in flight (one stream each)")
    ap.add_argument("--proofs", type=int, default=96, help="proofs timed in the batched run")
    args = ap.parse_args(argv)
    import torch
    import zkwg
    from zkwg import prover, synth
    from zkwg import r1cs as zr
    N, M, n = args.max_header, args.max_body, args.emails
    R = prover.R
    t0 = time.time()
    c = zkwg.Circuit(zkwg.MAIN_EMAIL_VERIFIER, max_header=N, max_body=M, device=0)
    sym = c.symbols()
    n_public = 20
    full = zr.append_public_rows(zr.email_verifier_constraints(sym, N, M), n_public)
    data = zr.write_r1cs(len(sym), full, n_pub_out=3, n_pub_in=17, n_prv_in=N + 1 + 17 + 1 + 32 + M + 1)
    power = max(1, (len(full) - 1).bit_length())
    rng = random.Random(1)
    # random bases: a few thousand distinct points repeated (fixed-base work is not what is timed)
    pool = [rng.randrange(1, R) for _ in range(4096)]
    rep = lambda k: [pool[(7 * i + 3) % 4096] for i in range(k)]
    pk = prover.ProvingKey.from_scalars(0, n_public, power, rep(c.W), rep(c.W), rep(c.W - n_public - 1), rep(1 << power), 5, 7, 11)
    pv = prover.Prover(c, data, len(full), pk)
    t_setup = time.time() - t0
    recs, _ = synth.packed_batch(c, seed=9, n=n, body_len=min(100, M - 80))
    dev = torch.device("cuda", 0)
    d_in = torch.frombuffer(bytearray(recs), dtype=torch.uint8).to(dev)
    d_status = torch.zeros(n, dtype=torch.int32, device=dev)
    d_scratch = torch.empty(c.scratch_bytes(n), dtype=torch.uint8, device=dev)
    c.prepare_device(d_in, n, d_status, d_scratch)
    torch.cuda.synchronize()
    assert d_status.tolist() == [0] * n
    pv.prove_prepared(d_in, n, d_scratch, 0, 3, 4)       # warm-up (builds the per-stage plans)
    torch.cuda.synchronize()
    # stage timings on one email
    def timed(f, reps=3):
        torch.cuda.synchronize(); t = time.time()
        for _ in range(reps): f()
        torch.cuda.synchronize(); return (time.time() - t) / reps * 1e3
    wit = pv.d_wit.data_ptr()
    st = {
        "witness_ms": timed(lambda: c.expand_device(d_in, n, d_scratch, 0, 1, pv.d_wit)),
        "abc_ms": timed(lambda: c.expand_abc_device(d_in, n, d_scratch, 0, 1, pv.d_abc, montgomery=True)),
        "h_evaluations_ms": timed(lambda: pv.ntt.h_evaluations_device(pv.d_abc, c.abc_bytes, pv.m, 1, pv.d_ntt_work, pv.d_h)),
        "msm_a_ms": timed(lambda: pv.msm_a.run(wit, False, True, pv.d_msm_work)),
        "msm_b1_ms": timed(lambda: pv.msm_b1.run(wit, False, True, pv.d_msm_work)),
        "msm_b2_g2_ms": timed(lambda: pv.msm_b2.run(wit, False, True, pv.d_msm_work)),
        "msm_c_ms": timed(lambda: pv.msm_c.run(wit + 32 * (n_public + 1), False, True, pv.d_msm_work)),
        "msm_h_ms": timed(lambda: pv.msm_h.run(pv.d_h.data_ptr(), True, False, pv.d_msm_work)),
    }
    torch.cuda.synchronize()
    t = time.time()
    for e in range(min(n, 4)):
        pv.prove_prepared(d_in, n, d_scratch, e, 3 + e, 4 + e)
    torch.cuda.synchronize()
    per_single = (time.time() - t) / min(n, 4)
    # several proofs in flight
    idx = [e % n for e in range(args.proofs)]
    bl = [(3 + e, 4 + e) for e in range(args.proofs)]
    pv.prove_batch(d_in, n, d_scratch, idx[:args.slots], bl[:args.slots], slots=args.slots)     # buffers, first touch
    torch.cuda.synchronize()
    t = time.time()
    pv.prove_batch(d_in, n, d_scratch, idx, bl, slots=args.slots)
    torch.cuda.synchronize()
    per = (time.time() - t) / args.proofs
    out = {"circuit": f"EmailVerifier({N},{M},121,17,0,0,0,0)", "W": c.W, "constraints_with_public_rows": len(full), "domain_log2": power,
           "emails": n, "proofs_per_s": round(1 / per, 2), "ms_per_proof": round(per * 1e3, 2), "proofs_in_flight": args.slots, "proofs_timed": args.proofs, "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"),
           "msm_layout": "precomputed windows" if os.environ.get("ZKWG_MSM_PRECOMP", "1") != "0" else "classic",
           "one_at_a_time_ms_per_proof": round(per_single * 1e3, 2), "stages": {k: round(v, 2) for k, v in st.items()},
           "setup_s": round(t_setup, 1), "key": "random bases (timing only; validity: tests/test_prove.py under the pinned verifier)"}
    if not quiet:
        print(json.dumps(out))
    return out


if __name__ == "__main__":
    main()
