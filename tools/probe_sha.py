"""Tuning probe: time zk_sha_expand store-pattern variants (ZKWG_EXPAND_VARIANT) on one GPU."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "zk-email-verify_amd", "py"))
import torch
import zkwg

def run(variant, batch=1024, N=1024, iters=5):
    os.environ["ZKWG_EXPAND_VARIANT"] = str(variant)
    c = zkwg.Circuit(zkwg.MAIN_SHA256_BYTES, max_header=N, max_body=0, device=0)
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(1)
    recs = torch.randint(0, 256, (batch, c.in_stride), dtype=torch.uint8, generator=g)
    import struct
    off = c.lib.zkwg_input_offset(c.h, 6)
    recs[:, off:off+4] = torch.tensor(list(struct.pack("<I", N)), dtype=torch.uint8)
    d_in = recs.to(dev)
    d_out = torch.empty(batch * c.witness_bytes, dtype=torch.uint8, device=dev)
    d_status = torch.zeros(batch, dtype=torch.int32, device=dev)
    d_scr = torch.empty(c.scratch_bytes(batch), dtype=torch.uint8, device=dev)
    c.set_timing(True)
    st = torch.cuda.current_stream()
    best = {}
    for it in range(iters):
        c.calculate_batch_device(d_in, batch, d_out, d_status, d_scr, st)
        torch.cuda.synchronize()
        for k, (ms, slots) in c.kernel_times_ms().items():
            if k not in best or ms < best[k][0]:
                best[k] = (ms, slots)
    out = []
    for k, (ms, slots) in best.items():
        gbs = slots * 32 * batch / (ms * 1e-3) / 1e9 if slots else 0
        out.append(f"{k}: {ms:.3f} ms {gbs:.0f} GB/s")
    print(f"variant {variant} batch {batch} W={c.W}: " + " | ".join(out), flush=True)

if __name__ == "__main__":
    for v in [0]:
        run(v)
