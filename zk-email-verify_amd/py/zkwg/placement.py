"""Where the output ring lives in HBM ("placement").

Measured on MI355X (tools/placement_probe.py, tools/placebench.hip; profiles/r04_*placement*): the rate at which a
buffer takes zk_expand's stores depends on WHICH physical memory the driver handed out for it.  On every box of the pool
a contiguous part of the 288 GB (a quarter to a third of it) takes 32 KiB-per-workgroup store streams 12-20 % slower than
the rest (zk_expand 5.1-5.3 ms against 4.3-4.7 ms per 512 witnesses; a pure fill of the same shape 6.2 against 7.2 TB/s),
reproducibly to 0.1 % for a given allocation and independent of the kernel beside it; a 4 KiB-per-workgroup fill does not
feel it, which is why `torch.fill_` reports the same 6.6 TB/s everywhere.  The ring of output tiles is a long-lived
allocation of a witness service, and it needs 58 GB of the 288: so the pipeline allocates more candidate tiles than it
needs, times the real kernel on each, keeps the fastest and frees the others -- once, at set-up.
"""


def choose_tiles(torch, dev, tile_bytes, want, run, max_candidates=7, reserve_bytes=24 << 30, reps=2):
    """Allocate up to `max_candidates` buffers of `tile_bytes` (as many as free HBM minus `reserve_bytes` allows, at least
    `want`), time `run(buffer)` (a callable that launches the expansion of one tile into `buffer` on the current stream) on
    each, keep the `want` fastest and free the rest.  -> (list of kept tensors, report dict)."""
    free_b, _ = torch.cuda.mem_get_info(dev)
    n = int(max(want, min(max_candidates, (free_b - reserve_bytes) // max(tile_bytes, 1))))
    cands = []
    for _ in range(n):
        try:
            cands.append(torch.empty(tile_bytes, dtype=torch.uint8, device=dev))
        except RuntimeError:          # out of memory: work with what we have
            break
    if len(cands) < want:
        raise RuntimeError(f"could not allocate {want} output tiles of {tile_bytes} bytes")
    ms = []
    for t in cands:
        run(t)                        # first touch (page tables, caches)
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            run(t)
        e1.record()
        torch.cuda.synchronize(dev)
        ms.append(e0.elapsed_time(e1) / reps)
    order = sorted(range(len(cands)), key=lambda i: ms[i])
    keep = sorted(order[:want])
    kept = [cands[i] for i in keep]
    report = {"candidates": len(cands), "ms_per_tile": [round(x, 3) for x in ms], "kept": keep,
              "kept_ms": [round(ms[i], 3) for i in keep], "tile_bytes": tile_bytes}
    cands = t = None
    torch.cuda.empty_cache()
    return kept, report


class _Chunked:
    """owner of one zkwg_device_alloc_chunked buffer, exposed to torch through __cuda_array_interface__"""

    def __init__(self, lib, device, nbytes, chunk_bytes=0, extra=0):
        import ctypes as C
        self.lib, self.nbytes = lib, nbytes
        p = C.c_void_p()
        rates = (C.c_float * 512)()
        nr = C.c_uint32(0)
        rc = lib.zkwg_device_alloc_chunked_ex(device, nbytes, chunk_bytes, extra, C.byref(p), rates, 512, C.byref(nr))
        self.rates = [round(rates[i], 0) for i in range(nr.value)]
        if rc != 0:
            raise RuntimeError(f"zkwg_device_alloc_chunked({nbytes}) failed: {lib.zkwg_strerror(rc).decode()}")
        self.ptr = p.value
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (self.ptr, False), "version": 2}

    def __del__(self):
        if getattr(self, "ptr", None):
            self.lib.zkwg_device_free_chunked(self.ptr)
            self.ptr = None


def chunked_tensor(torch, dev, nbytes, chunk_bytes=0, extra=0):
    """a uint8 tensor of nbytes on `dev` mapped from physical chunks (include/zkwg.h zkwg_device_alloc_chunked): what the output
    ring is made of since round 5 -- every tile takes zk_expand's stores at the rate of round 4's best candidates, with no candidates"""
    from . import _lib
    owner = _Chunked(_lib.load(), dev.index if dev.index is not None else 0, int(nbytes), chunk_bytes, extra)
    t = torch.as_tensor(owner, device=dev)
    t._zkwg_owner = owner          # the tensor keeps the mapping alive
    return t
