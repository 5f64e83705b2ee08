import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "zk-email-verify_amd", "py"))
import torch, zkwg
from zkwg import synth
c = zkwg.Circuit(zkwg.MAIN_EMAIL_VERIFIER, max_header=1024, max_body=1536, device=0)
n = 512
dev = torch.device("cuda:0")
d_out = torch.empty(n * c.witness_bytes, dtype=torch.uint8, device=dev)
def t(fn, name, nbytes, iters=5):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / iters
    print(f"{name}: {ms:.3f} ms {nbytes/ms/1e6:.0f} GB/s", flush=True)
t(lambda: d_out.zero_(), "torch zero_ on the 29 GB witness buffer", d_out.numel())
v = d_out.view(torch.int64)
t(lambda: v.fill_(1), "torch fill_(int64 1)", d_out.numel())
recs, _ = synth.packed_batch(c, seed=1, n=32, body_len=1024)
h = torch.frombuffer(bytearray(recs), dtype=torch.uint8).view(32, c.in_stride)
d_in = h.repeat(n // 32, 1).contiguous().to(dev)
d_st = torch.zeros(n, dtype=torch.int32, device=dev)
d_scr = torch.empty(c.scratch_bytes(n), dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream()
c.prepare_device(d_in, n, d_st, d_scr, st)
t(lambda: c.expand_device(d_in, n, d_scr, 0, n, d_out, st), "zk_expand (both kernels)", d_out.numel())
