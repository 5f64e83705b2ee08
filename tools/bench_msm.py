#!/usr/bin/env python3
"""Times the G1 multi-exponentiation at the H stage's sizes: n = 2^log2 bases (multiples of 256 random points),
random 254-bit scalars in Montgomery form as the transform stage leaves them, --emails vectors per launch series.  Prints one JSON line;
the bound is multiplier issue: K * n mixed additions x 10 products + the slices' and bit planes' full additions, against the 139 G
products/s the 9 x 29-bit product sustains behind the 4 x 64-bit interface (162.9 G/s in the lazy form the sums run; tools/mulbench.hip).

    python tools/bench_msm.py [--log2 20] [--window 0] [--reps 3] [--emails 4] [--slice0 64]
"""
import argparse
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "zk-email-verify_amd", "py"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2", type=int, default=20)
    ap.add_argument("--window", type=int, default=0)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--witness", type=int, default=0, help="1: witness-like scalars (90 %% bits, bytes, a few field elements), ones_apart")
    ap.add_argument("--emails", type=int, default=1, help="scalar vectors per launch series")
    ap.add_argument("--slice0", type=int, default=0, help="bucket entries per lane at the first slice level (0: 64 for full-size scalars, 16 for a witness)")
    args = ap.parse_args()
    import torch
    import zkwg
    from zkwg import prover
    n = 1 << args.log2
    dev = torch.device("cuda", 0)
    # bases: 256 fixed-base multiples of the generator made on the device (zkwg_fixed_base_device), repeated -- timing needs points of the
    # right shape, not distinct ones
    rng = random.Random(1)
    d_base = prover.fixed_base(0, 1, [rng.randrange(1, prover.R) for _ in range(256)])
    E = args.emails
    m = prover._DeviceMsm(d_base.repeat(n // 256), 1, 0, window_bits=args.window, slice0=args.slice0 or (16 if args.witness else 64))
    # scalars: random bytes with the top bits cleared (< 2^253 < r), declared to be in Montgomery form
    d_s = torch.randint(0, 256, (E * n, 32), dtype=torch.uint8, device=dev)
    d_s[:, 31] &= 0x1F
    mont = True
    if args.witness:
        kind = torch.rand(E * n, device=dev)
        d_s[kind < 0.97, 1:] = 0                       # bytes
        d_s[kind < 0.90, 0] &= 1                       # bits
        mont = False
    d_w = torch.empty(m.work_bytes(E) + 256, dtype=torch.uint8, device=dev)
    off = (-d_w.data_ptr()) % 256
    m.run_batch(d_s.data_ptr(), 32 * n, E, mont, bool(args.witness), d_w[off:])
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(args.reps):
        p = m.run_batch(d_s.data_ptr(), 32 * n, E, mont, bool(args.witness), d_w[off:])[0]
    torch.cuda.synchronize()
    ms = (time.time() - t0) / args.reps * 1e3 / E
    c = m.lib.zkwg_msm_window_bits(m.h)
    pt = prover.point_from_montgomery(p)        # affine integers, None = infinity
    Q = 21888242871839275222246405745257275088696311157297823662689037894645226208583
    K = (254 + c) // c
    s0 = args.slice0 or (16 if args.witness else 64)
    products = K * n * 10 + (K * n // s0 + (1 << (c - 1)) * (3 + c // 2)) * 14 + n
    print(json.dumps({"n": n, "emails_per_series": E, "window_bits": c, "windows": K, "ms_per_sum": round(ms, 3), "msm_per_s": round(1e3 / ms, 2),
                      "G_products_per_s": round(products / ms / 1e6, 2), "frac_of_139G_product_rate": round(products / ms / 1e6 / 139.0, 4),
                      "on_curve": pt is None or (pt[1] * pt[1] - pt[0] ** 3 - 3) % Q == 0}))


if __name__ == "__main__":
    main()
