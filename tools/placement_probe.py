#!/usr/bin/env python3
"""Does the HBM write rate depend on WHERE a buffer was placed?  (tools/beside2.py: the same kernel on the same images is
reproducible to 0.1 % on one allocation and differs by up to 15 % between allocations at the same virtual address.)

Allocates K output tiles of 512 witnesses (29 GB each), and for each one times
  * zk_expand3 writing the same 512 prepared emails into it (the product kernel),
  * torch.fill_ over the whole tile (a plain 16-byte-store fill),
  * torch.fill_ over each eighth of the tile (is a slow tile slow everywhere or in places?).

    python tools/placement_probe.py [--tiles 6] [--out gpurun_out/placement.json]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "zk-email-verify_amd", "py"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tiles", type=int, default=6)
    ap.add_argument("--tile", type=int, default=512)
    ap.add_argument("--reps", type=int, default=4)
    ap.add_argument("--one-block", type=int, default=0, help="1: carve the tiles out of ONE allocation instead of one allocation each")
    ap.add_argument("--variants", default="", help="comma list of ENV=V+ENV=V handle variants, e.g. ZKWG_XCD_REMAP=0,ZKWG_XCD_REMAP=217+ZKWG_X3_K=2")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    import torch
    import zkwg
    import bench
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    c = zkwg.Circuit(zkwg.MAIN_EMAIL_VERIFIER, max_header=1024, max_body=1536, device=0)
    n = args.tile
    _, d_in, _ = bench.resident_inputs(torch, c, dev, 0x5A4B, n, n, 1024)
    s = torch.cuda.current_stream()
    d_status = torch.zeros(n, dtype=torch.int32, device=dev)
    d_scr = torch.empty(c.scratch_bytes(n), dtype=torch.uint8, device=dev)
    c.prepare_device(d_in, n, d_status, d_scr, s)
    torch.cuda.synchronize()
    assert int(d_status.abs().sum().item()) == 0
    nbytes = n * c.witness_bytes
    if args.one_block:
        big = torch.empty(args.tiles * nbytes, dtype=torch.uint8, device=dev)
        tiles = [big[i * nbytes:(i + 1) * nbytes] for i in range(args.tiles)]
    else:
        tiles = [torch.empty(nbytes, dtype=torch.uint8, device=dev) for _ in range(args.tiles)]
    bpe = 32 * c.W + c.in_stride

    def t_ms(fn, reps):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    rows = []
    if args.variants:
        # the same tiles under other workgroup -> piece mappings / piece sizes (read from the environment at handle creation)
        for spec in args.variants.split(","):
            env = dict(kv.split("=") for kv in spec.split("+"))
            for k, v in env.items():
                os.environ[k] = v
            cv = zkwg.Circuit(zkwg.MAIN_EMAIL_VERIFIER, max_header=1024, max_body=1536, device=0)
            for k in env:
                del os.environ[k]
            ms = [round(t_ms(lambda: cv.expand_device(d_in, n, d_scr, 0, n, t, s), args.reps), 3) for t in tiles]
            row = {"variant": spec, "expand_ms_per_tile": ms, "frac_per_tile": [round(bpe * n / (x * 1e-3) / 1e9 / 8000, 3) for x in ms]}
            rows.append(row)
            print(json.dumps(row), flush=True)
            del cv
    for rnd in range(0 if args.variants else 2):
        for i, t in enumerate(tiles):
            ex = t_ms(lambda: c.expand_device(d_in, n, d_scr, 0, n, t, s), args.reps)
            v = t.view(torch.int32)
            fl = t_ms(lambda: v.fill_(1), args.reps)
            per = v.numel() // 8
            parts = [round(per * 4 / (t_ms(lambda k=k: v[k * per:(k + 1) * per].fill_(1), args.reps) * 1e-3) / 1e9) for k in range(8)]
            row = {"round": rnd, "tile": i, "addr": hex(t.data_ptr()), "expand_ms": round(ex, 4), "expand_GBps": round(bpe * n / (ex * 1e-3) / 1e9, 1),
                   "expand_frac": round(bpe * n / (ex * 1e-3) / 1e9 / 8000, 4), "fill_GBps": round(nbytes / (fl * 1e-3) / 1e9, 1), "fill_eighths_GBps": parts}
            rows.append(row)
            print(json.dumps(row), flush=True)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        json.dump(rows, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
