#!/usr/bin/env python3
"""Rate of prover stage 2 (zkwg_h_evaluations_device): 3 inverse + 3 forward transforms on the 2^20
domain of EmailVerifier(576,192)'s constraint system and a b - c, for a batch of emails, against the multiplier-issue roofline
(a Montgomery product = 128 v_mad_u64_u32 at the issue rate tools/mulbench.hip MEASURED on gfx950 -- 34.4 T lane-ops/s, i.e.
263 G products/s; rounds 2-4 assumed a quarter-rate multiplier, 76.8 G: profiles/r05/r05_b_mulbench.json)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "zk-email-verify_amd", "py"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2n", type=int, default=20)
    ap.add_argument("--constraints", type=int, default=753807)
    ap.add_argument("--emails", type=int, default=16)
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    import torch
    import zkwg
    dev = torch.device("cuda", 0)
    L, n, m, E = args.log2n, 1 << args.log2n, args.constraints, args.emails
    plan = zkwg.Ntt(L)
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    d_abc = torch.randint(0, 1 << 62, (E, 12 * m), dtype=torch.int64, device=dev, generator=g)   # random 62-bit limbs: valid Montgomery-form residues are not needed for timing
    d_abc[:, 3::4] >>= 4                                                                           # (top limb below the modulus)
    d_work = torch.empty(plan.work_bytes(E), dtype=torch.uint8, device=dev)
    d_out = torch.empty(E * 32 * n, dtype=torch.uint8, device=dev)
    run = lambda: plan.h_evaluations_device(d_abc.view(torch.uint8), 96 * m, m, E, d_work, d_out)
    run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    sec = e0.elapsed_time(e1) * 1e-3 / args.reps
    passes = (L + 6) // 7 - 1                       # column passes (each closes / opens with one twiddle product per element)
    per_transform = n * L // 2 + passes * n
    products = 6 * per_transform + 3 * n + 2 * n    # + the coset scaling of the three inverse transforms + a b (x 2^266) of the join
    peak = 263.0e9      # measured v_mad_u64_u32 issue rate / 128 (tools/mulbench.hip)
    rate = 162.9e9      # the product the butterflies run since round 6: 9 x 29-bit limbs, values kept in limb form (zkwg_fr29.h), in a pure product loop (profiles/r05/r05_g_mulbench.txt); 139.0 G/s behind the 4 x 64-bit interface
    print(json.dumps({"stage": "H evaluations (3 ifft, coset shift, 3 fft, a b - c)", "log2_domain": L, "constraints": m, "emails": E,
                      "emails_per_s": round(E / sec, 1), "ms_per_email": round(sec / E * 1e3, 3),
                      "montgomery_products_per_email": products, "products_per_s": round(E * products / sec),
                      "issue_roofline_products_per_s": round(peak), "frac_of_issue_roofline": round(E * products / sec / peak, 4),
                      "frac_of_limb_form_product_rate": round(E * products / sec / rate, 4), "frac_of_139G": round(E * products / sec / 139.0e9, 4),
                      "hbm_bytes_per_email": (2 * passes + 2) * 2 * 3 * 36 * n + 4 * 32 * n,
                      "hbm_GBps": round(E * ((2 * passes + 2) * 2 * 3 * 36 * n + 4 * 32 * n) / sec / 1e9, 1)}))


if __name__ == "__main__":
    main()
