#!/usr/bin/env python3
"""Which prepare kernel costs zk_expand how much.

The headline pipeline (bench.py: EmailVerifier(1024,1536), batch 4096, tiles of 512, prepare sub-batches of 1024 on a second
stream) is run with subsets of the prepare kernels (zkwg_set_prepare_mask) and the average zk_expand launch duration (HIP
events on the expand stream) is recorded for each:

    all        every prepare kernel beside the stream (the bench configuration)
    alone      --no-overlap: zk_expand has the chip to itself
    without K  everything but kernel K         -> marginal cost of K in the real pipeline
    only K     kernel K alone beside the stream -> cost of K by itself

The images the expansions read stay valid throughout: the warm-up steps run the full prepare into every scratch buffer of
the ring, a masked prepare then rewrites a subset of the same values.  All configurations run on ONE pipeline (same output
ring, same scratch buffers): the store rate depends on the placement of the ring (zkwg.placement), not only on the neighbours.

    python tools/beside.py [--steps 3] [--out gpurun_out/beside.json]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "zk-email-verify_amd", "py"))

KERNELS = {"zk_sha_chain": 1, "zk_sha_trace": 2, "zk_misc_ev": 8, "zk_rsa": 16, "zk_poseidon9": 32}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--tile", type=int, default=512)
    ap.add_argument("--prep-batch", type=int, default=1024)
    ap.add_argument("--out", default=None)
    ap.add_argument("--modes", default="all,alone,without,only")
    args = ap.parse_args()
    import torch
    import zkwg
    import bench
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    c = zkwg.Circuit(zkwg.MAIN_EMAIL_VERIFIER, max_header=1024, max_body=1536, device=0)
    _, d_in, _ = bench.resident_inputs(torch, c, dev, 0x5A4B, 512, args.batch, 1024)
    bpe = 32 * c.W + c.in_stride
    res = {"workload": f"EmailVerifier(1024,1536) batch {args.batch}, tile {args.tile}, prepare sub-batch {args.prep_batch}", "rows": []}

    # ONE pipeline for every configuration: the rate HBM takes the stores at depends on where the output ring was placed
    # (zkwg.placement), so configurations are only comparable on the same buffers
    pl = bench.Pipeline(torch, c, dev, d_in, args.batch, args.tile, args.prep_batch)
    res["ring_placement"] = pl.placement
    for _ in range(2):
        pl.step()
    torch.cuda.synchronize()
    assert int(pl.d_status.abs().sum().item()) == 0

    def run(label, mask, serial=False):
        pl.serial = serial
        c.set_prepare_mask(mask)
        pl.step()                      # one step in the measured configuration before timing starts
        torch.cuda.synchronize()
        c.set_timing(True)
        dt = bench.timed(torch, pl.step, steps=args.steps, warmup=0)
        summ, avg, n, gbs = bench.expand_roofline(c, args.tile)
        c.set_timing(False)
        c.set_prepare_mask(0xFFFFFFFF)
        pl.serial = False
        row = {"config": label, "zk_expand_ms": round(avg, 4), "GBps": round(gbs, 1), "frac": round(gbs / bench.HBM_PEAK_GBS, 4),
               "witnesses_per_s": round(args.batch * args.steps / dt, 1),
               "prepare_ms_per_launch": {k: round(v[0] / max(v[1], 1), 3) for k, v in summ.items() if k != "zk_expand"}}
        res["rows"].append(row)
        print(json.dumps(row), flush=True)

    modes = args.modes.split(",")
    if "all" in modes:
        run("all", 0xFFFFFFFF)
    if "alone" in modes:
        run("alone (no overlap)", 0xFFFFFFFF, serial=True)
    if "without" in modes:
        for k, bit in KERNELS.items():
            run("without " + k, 0xFFFFFFFF & ~bit)
    if "only" in modes:
        for k, bit in KERNELS.items():
            run("only " + k, bit)
        run("no prepare kernels", 0)
    if "cumask" in modes:
        # the prepare kernels confined to a few compute units (CU-masked streams, zkwg_stream_create_masked), zk_expand on the
        # others -- on the SAME ring as the rows above (round 3's verdict on this knob, "not robust", was taken across allocations)
        import ctypes
        ncu = torch.cuda.get_device_properties(dev).multi_processor_count
        words = (ncu + 31) // 32
        keep_p, keep_e = pl.s_preps, pl.s_exp

        def mask(keep):
            m = [0] * words
            for i in range(ncu):
                if keep(i):
                    m[i // 32] |= 1 << (i % 32)
            return (ctypes.c_uint32 * words)(*m), words
        for n_cu, stride, share in ((32, 8, False), (64, 4, False), (32, 1, False), (16, 8, False), (32, 8, True), (64, 4, True)):
            sel = set(i * stride for i in range(n_cu) if i * stride < ncu)
            mp, w = mask(lambda i: i in sel)
            me, _ = mask(lambda i: share or i not in sel)        # share: zk_expand may use every CU, only the prepare side is confined
            sp = c.lib.zkwg_stream_create_masked(0, mp, w)
            se = c.lib.zkwg_stream_create_masked(0, me, w)
            assert sp and se
            pl.s_preps = [torch.cuda.ExternalStream(sp, device=dev)]
            pl.s_exp = torch.cuda.ExternalStream(se, device=dev)
            torch.cuda.synchronize()
            run(f"prepare on {n_cu} CUs (every {stride}th), zk_expand on {'all' if share else 'the others'}", 0xFFFFFFFF)
            torch.cuda.synchronize()
            pl.s_preps, pl.s_exp = keep_p, keep_e
            c.lib.zkwg_stream_destroy(sp)
            c.lib.zkwg_stream_destroy(se)
    if "all" in modes:
        run("all (again)", 0xFFFFFFFF)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        json.dump(res, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
