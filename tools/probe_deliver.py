"""Host-delivered rate of the C-ABI host path (zkwg_calculate_batch): inputs in host memory, witnesses
written to (pinned or pageable) host memory -- PCIe-bound, never the bench `value`."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "zk-email-verify_amd", "py"))
import zkwg
from zkwg import synth

c = zkwg.Circuit(zkwg.MAIN_EMAIL_VERIFIER, max_header=1024, max_body=1536, device=0)
n, tile = 256, 64
recs, _ = synth.packed_batch(c, seed=3, n=32, body_len=1024)
recs = recs * (n // 32)
status = (C.c_int32 * n)()
for pinned in (True, False):
    nbytes = n * c.witness_bytes
    if pinned:
        ptr = c.lib.zkwg_alloc_pinned(nbytes)
        out = C.cast(ptr, C.POINTER(C.c_uint8))
    else:
        buf = (C.c_uint8 * nbytes)()
        out = buf
    for it in range(2):
        t0 = time.perf_counter()
        rc = c.lib.zkwg_calculate_batch(c.h, recs, n, out, c.witness_bytes, status, tile)
        dt = time.perf_counter() - t0
    assert rc == 0 and sum(status) == 0
    print(f"{'pinned' if pinned else 'pageable'} host buffer: {n / dt:.0f} witnesses/s delivered, {nbytes / dt / 1e9:.1f} GB/s over PCIe", flush=True)
    if pinned:
        c.lib.zkwg_free_pinned(ptr)
