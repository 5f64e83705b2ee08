"""Tuning probe: standalone zk_expand timing (no concurrent prepare) for several geometries."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "zk-email-verify_amd", "py"))
import torch
import zkwg
from zkwg import synth

def run(portion, threads, epw, n=512, iters=6):
    os.environ["ZKWG_PORTION"] = str(portion); os.environ["ZKWG_EXPAND_THREADS"] = str(threads); os.environ["ZKWG_EMAILS_PER_WG"] = str(epw)
    c = zkwg.Circuit(zkwg.MAIN_EMAIL_VERIFIER, max_header=1024, max_body=1536, device=0)
    dev = torch.device("cuda:0")
    recs, _ = synth.packed_batch(c, seed=1, n=32, body_len=1024)
    h = torch.frombuffer(bytearray(recs), dtype=torch.uint8).view(32, c.in_stride)
    d_in = h.repeat(n // 32, 1).contiguous().to(dev)
    d_out = torch.empty(n * c.witness_bytes, dtype=torch.uint8, device=dev)
    d_st = torch.zeros(n, dtype=torch.int32, device=dev)
    d_scr = torch.empty(c.scratch_bytes(n), dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream()
    c.prepare_device(d_in, n, d_st, d_scr, st)
    torch.cuda.synchronize()
    c.set_timing(True)
    for _ in range(iters):
        c.expand_device(d_in, n, d_scr, 0, n, d_out, st)
    torch.cuda.synchronize()
    ms, cnt, slots = c.timing_summary()["zk_expand"]
    ms /= cnt
    print(f"portion={portion} threads={threads} epw={epw}: {ms:.3f} ms  {32*c.W*n/ms/1e6:.0f} GB/s", flush=True)
    del d_out, d_scr

if __name__ == "__main__":
    cfgs = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]] or [(1024, 256, 1), (2048, 256, 1), (512, 256, 8), (256, 256, 8), (256, 256, 16)]
    for cfg in cfgs:
        run(*cfg)
