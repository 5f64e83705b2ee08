"""Rate of the first Groth16 prover stage on device-resident witnesses (zkwg_r1cs_evaluate_device, DESIGN.md
section 15): EmailVerifier(576,192), the kept-v1 constraint system exported by zkwg.r1cs, witnesses written in
Montgomery form by the fused expand and never leaving the device."""
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
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    c = zkwg.Circuit(zkwg.MAIN_EMAIL_VERIFIER, max_header=576, max_body=192, device=0)
    t0 = time.time()
    cs = zkwg.WitnessCalculator(c).constraint_system()
    t_cs = time.time() - t0
    n = 256
    _, d_in, _ = bench.resident_inputs(torch, c, dev, 0x5A4B + 31, 64, n, 60)
    d_status = torch.zeros(n, dtype=torch.int32, device=dev)
    d_scr = torch.empty(c.scratch_bytes(n), dtype=torch.uint8, device=dev)
    d_wit = torch.empty(n * c.witness_bytes, dtype=torch.uint8, device=dev)
    s = torch.cuda.current_stream()
    c.prepare_device(d_in, n, d_status, d_scr, s)
    c.expand_montgomery_device(d_in, n, d_scr, 0, n, d_wit, s)
    torch.cuda.synchronize()
    assert int(d_status.abs().sum().item()) == 0
    cs.evaluate_device(d_wit, n, c.witness_bytes, s, montgomery=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        out = cs.evaluate_device(d_wit, n, c.witness_bytes, s, montgomery=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    res_image = {}
    del out, d_wit
    torch.cuda.empty_cache()
    # the same evaluations written from the compact image (zkwg_circuit_attach_r1cs + zkwg_expand_abc_device), as a
    # prepare / expand pipeline over resident inputs like bench.py's
    c2 = zkwg.Circuit(zkwg.MAIN_EMAIL_VERIFIER, max_header=576, max_body=192, device=0)
    t0 = time.time()
    c2.attach_r1cs(cs)
    t_attach = time.time() - t0
    batch, tile, prep = 2048, 256, 1024
    _, d_in2, _ = bench.resident_inputs(torch, c2, dev, 0x5A4B + 32, 64, batch, 60)
    for mont in (True, False):
        pl = bench.Pipeline(torch, c2, dev, d_in2, batch, tile, prep, ring=2, montgomery=mont, abc=True, serial=bool(int(os.environ.get('ZKWG_BENCH_SERIAL', '0'))))
        c2.set_timing(True)
        dtp = bench.timed(torch, pl.step, steps=3, warmup=1)
        summ = c2.timing_summary()
        c2.set_timing(False)
        assert int(pl.d_status.abs().sum().item()) == 0
        res_image["montgomery" if mont else "standard"] = {
            "witnesses_per_s": round(batch * 3 / dtp, 1), "GBps_written": round(batch * 3 * c2.abc_bytes / dtp / 1e9, 1),
            "kernel_ms": {k: round(v[0] / max(v[1], 1), 3) for k, v in summ.items()}}
        del pl
        torch.cuda.empty_cache()
    res_image["attach_s"] = round(t_attach, 1)
    res_image["image_bytes_per_email"] = c2.scratch_bytes(1)
    print(json.dumps({"from_image": res_image, "circuit": "EmailVerifier(576,192,121,17,0,0,0,0) kept-v1", "constraints": cs.n_constraints, "witnesses": n,
                      "ms_per_launch": round(dt * 1e3, 3), "witnesses_per_s": round(n / dt, 1),
                      "evaluations_per_s": round(3 * cs.n_constraints * n / dt, 1),
                      "GBps_written": round(96 * cs.n_constraints * n / dt / 1e9, 1), "r1cs_export_s": round(t_cs, 1)}))


if __name__ == "__main__":
    main()
