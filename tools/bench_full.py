"""Throughput of complete `--O0`-numbered witnesses (zkwg_circuit_create_full) for
EmailVerifier(576,192) -- or `bench_full.py 1024 1536` -- : the row kernels at the end of prepare, then zk_expand3_o0 (one pass
from per-wire descriptors), beside the kept-v1 witness of the same circuit.  ZKWG_BENCH_PRIO="expand,prepare" stream priorities,
ZKWG_BENCH_PREP = emails per prepare launch.  Needs artifacts/o0_ev_576_192.* (built by __graft_entry__.build() where /root/reference exists; travels to
the GPU box) -- a measurement aid, not part of the product."""
import gzip
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "zk-email-verify_amd", "py"))
sys.path.insert(0, ROOT)


def main():
    import torch
    import zkwg
    import bench
    N, M = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (576, 192)
    base = os.path.join(ROOT, "artifacts", f"o0_ev_{N}_{M}")
    meta = json.load(open(base + ".json"))
    sym = gzip.open(base + ".sym.gz", "rb").read()
    r1cs = gzip.open(base + ".r1cs.gz", "rb").read()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    out = {}
    for name, kw in (("kept-v1", {}), ("complete O0", dict(sym=sym, sym_alias=meta["alias"], r1cs=r1cs))):
        t0 = time.time()
        c = zkwg.Circuit(zkwg.MAIN_EMAIL_VERIFIER, max_header=N, max_body=M, device=0, **kw)
        t_create = time.time() - t0
        batch, tile = (2048, 256) if N <= 576 else (512, 128)
        _, d_in, _ = bench.resident_inputs(torch, c, dev, 0x5A4B + 9, 64, batch, 60 if M <= 192 else 1024)
        prio = [int(x) for x in os.environ.get("ZKWG_BENCH_PRIO", "-1,0").split(",")]   # (expand, prepare) stream priorities
        pl = bench.Pipeline(torch, c, dev, d_in, batch, tile, min(int(os.environ.get("ZKWG_BENCH_PREP", "1024")), batch), ring=2, exp_prio=prio[0], prep_prio=prio[1], serial=bool(int(os.environ.get('ZKWG_BENCH_SERIAL', '0'))))
        c.set_timing(True)
        dt = bench.timed(torch, pl.step, steps=3, warmup=1)
        summ = c.timing_summary()
        c.set_timing(False)
        assert int(pl.d_status.abs().sum().item()) == 0
        out[name] = {"witness_len": c.W, "witnesses_per_s": round(batch * 3 / dt, 1), "GBps_written": round(batch * 3 * c.witness_bytes / dt / 1e9, 1),
                     "create_s": round(t_create, 1), "kernel_ms": {k: round(v[0] / max(v[1], 1), 3) for k, v in summ.items()}}
        del pl, d_in, c
        torch.cuda.empty_cache()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
