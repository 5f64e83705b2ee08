#!/usr/bin/env python3
"""Separates what tools/beside.py confounds: the prepare kernels that run beside zk_expand, the ALLOCATION the witnesses are
written to, and run-to-run noise.  For each of `--allocs` freshly allocated pipelines (new output ring / scratch buffers) the
configurations none / all / only zk_poseidon9 / alone are each run `--reps` times on the SAME buffers; every zk_expand launch
is timed with its own pair of events on the expand stream, and min / median / max are reported beside the mean.

    python tools/beside2.py [--allocs 3] [--reps 2] [--steps 2] [--out gpurun_out/beside2.json]
"""
import argparse
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "zk-email-verify_amd", "py"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--allocs", type=int, default=3)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--tile", type=int, default=512)
    ap.add_argument("--prep-batch", type=int, default=1024)
    ap.add_argument("--pad-mib", type=int, default=0, help="allocate (and keep) this many MiB more before every new pipeline: moves the buffers")
    ap.add_argument("--configs", default="none,all,pos,alone")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    import torch
    import zkwg
    import bench
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    c = zkwg.Circuit(zkwg.MAIN_EMAIL_VERIFIER, max_header=1024, max_body=1536, device=0)
    _, d_in, _ = bench.resident_inputs(torch, c, dev, 0x5A4B, 512, args.batch, 1024)
    bpe = 32 * c.W + c.in_stride
    CFG = {"none": (0, False), "all": (0xFFFFFFFF, False), "pos": (32, False), "rsa": (16, False), "alone": (0xFFFFFFFF, True)}
    rows = []
    pads = []
    for a in range(args.allocs):
        if args.pad_mib:
            pads.append(torch.empty(args.pad_mib << 20, dtype=torch.uint8, device=dev))
        pl = bench.Pipeline(torch, c, dev, d_in, args.batch, args.tile, args.prep_batch, place=False)
        addr = [int(t.data_ptr()) for t in pl.d_out]
        inner = pl.expand
        evs = []

        def timed_expand(*aa, **kw):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(pl.s_exp)
            inner(*aa, **kw)
            e1.record(pl.s_exp)
            evs.append((e0, e1))
        pl.expand = timed_expand
        c.set_prepare_mask(0xFFFFFFFF)
        for _ in range(2):
            pl.step()
        torch.cuda.synchronize()
        for rep in range(args.reps):
            for name in args.configs.split(","):
                mask, serial = CFG[name]
                pl.serial = serial
                c.set_prepare_mask(mask)
                pl.step()
                torch.cuda.synchronize()
                evs.clear()
                dt = bench.timed(torch, pl.step, steps=args.steps, warmup=0)
                ms = sorted(e0.elapsed_time(e1) for e0, e1 in evs)
                row = {"alloc": a, "rep": rep, "config": name, "d_out": [hex(x) for x in addr], "mean_ms": round(sum(ms) / len(ms), 4),
                       "min_ms": round(ms[0], 4), "median_ms": round(statistics.median(ms), 4), "max_ms": round(ms[-1], 4),
                       "frac_mean": round(bpe * args.tile / (sum(ms) / len(ms) * 1e-3) / 1e9 / bench.HBM_PEAK_GBS, 4),
                       "witnesses_per_s": round(args.batch * args.steps / dt, 1)}
                rows.append(row)
                print(json.dumps(row), flush=True)
        c.set_prepare_mask(0xFFFFFFFF)
        pl.serial = False
        del pl
        torch.cuda.empty_cache()
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        json.dump(rows, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
