#!/usr/bin/env python3
"""Pipeline granularity on ONE placed output ring: emails per prepare launch (and prepare streams) against the headline rate.
    python tools/prep_sweep.py [--steps 4]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "zk-email-verify_amd", "py"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=4)
    args = ap.parse_args()
    import torch
    import zkwg
    import bench
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    c = zkwg.Circuit(zkwg.MAIN_EMAIL_VERIFIER, max_header=1024, max_body=1536, device=0)
    _, d_in, _ = bench.resident_inputs(torch, c, dev, 0x5A4B, 512, 4096, 1024)
    base = bench.Pipeline(torch, c, dev, d_in, 4096, 512, 1024)
    print(json.dumps({"ring_placement": base.placement}), flush=True)
    for prep, streams, ring in ((1024, 1, 2), (512, 1, 2), (2048, 1, 2), (4096, 1, 2), (512, 2, 4), (1024, 2, 3)):
        pl = bench.Pipeline(torch, c, dev, d_in, 4096, 512, prep, ring=ring, prep_streams=streams, place=False)
        pl.d_out = base.d_out
        for _ in range(2):
            pl.step()
        torch.cuda.synchronize()
        assert int(pl.d_status.abs().sum().item()) == 0
        c.set_timing(True)
        dt = bench.timed(torch, pl.step, steps=args.steps, warmup=0)
        _, avg, n, gbs = bench.expand_roofline(c, 512)
        c.set_timing(False)
        print(json.dumps({"prep": prep, "prep_streams": streams, "ring": ring, "witnesses_per_s": round(4096 * args.steps / dt, 1),
                          "zk_expand_ms": round(avg, 4), "frac": round(gbs / 8000, 4)}), flush=True)
        del pl
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
