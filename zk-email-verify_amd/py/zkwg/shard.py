"""Multi-GPU sharding of a batch: one process per GPU, emails are independent
units, rank r owns a contiguous range, no data-path collective.  The only exchange is the
gather of the small per-email result table {status, pubkeyHash, shaHi, shaLo} (100 bytes per
email) to rank 0 -- over RCCL/xGMI on GPUs (backend "nccl"), gloo in the CPU tests.  The bulk
`.wtns` bytes stay on the GPU that produced them (or leave through that GPU's own PCIe link).
"""
import torch

TABLE_BYTES = 4 + 3 * 32  # i32 status + three 32-byte public outputs


def shard_range(n_total, rank, world):
    """Contiguous range [lo, hi) of emails owned by `rank` (first n_total % world ranks get one more)."""
    base, extra = divmod(n_total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def result_table(status, witness_rows):
    """status: int32[n]; witness_rows: uint8[n, >=128] (first 128 bytes of each witness: w[0..3]).
    Returns uint8[n, TABLE_BYTES]."""
    n = status.shape[0]
    t = torch.empty((n, TABLE_BYTES), dtype=torch.uint8, device=status.device)
    t[:, :4] = status.view(torch.uint8).view(n, 4)
    t[:, 4:] = witness_rows[:, 32:128]
    return t


def gather_table(dist, local_table, n_total, rank, world):
    """Gather every rank's table on rank 0 (returns uint8[n_total, TABLE_BYTES] there, None elsewhere).
    Shards may differ by one row, so rows are padded to the largest shard for the collective."""
    sizes = [shard_range(n_total, r, world)[1] - shard_range(n_total, r, world)[0] for r in range(world)]
    mx = max(sizes)
    pad = torch.zeros((mx, TABLE_BYTES), dtype=torch.uint8, device=local_table.device)
    pad[:local_table.shape[0]] = local_table
    if world == 1:
        return local_table
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == 0 else None
    dist.gather(pad, bufs, dst=0)
    if rank != 0:
        return None
    return torch.cat([bufs[r][:sizes[r]] for r in range(world)], dim=0)


def gather_rows(dist, local_rows, n_total, rank, world):
    """gather_table for rows of any width (uint8[n_local, width]): the prover's status + proof rows (4 + 256 bytes per email)"""
    sizes = [shard_range(n_total, r, world)[1] - shard_range(n_total, r, world)[0] for r in range(world)]
    if world == 1:
        return local_rows
    pad = torch.zeros((max(sizes), local_rows.shape[1]), dtype=torch.uint8, device=local_rows.device)
    pad[:local_rows.shape[0]] = local_rows
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == 0 else None
    dist.gather(pad, bufs, dst=0)
    if rank != 0:
        return None
    return torch.cat([bufs[r][:sizes[r]] for r in range(world)], dim=0)


def gather_witnesses(dist, local_wtns, rank, world, chunk_bytes=1 << 30, sink=None):
    """Optional bulk exchange (BASELINE.json config C4): gather `local_wtns`
    (uint8[k, witness_bytes], the same k on every rank) on rank 0 over RCCL/xGMI (gloo in the CPU tests), in
    fixed chunks of at most `chunk_bytes` per rank so that the root needs only world x chunk of receive space.
    `sink(src_rank, byte_offset, chunk_tensor)` is called on rank 0 for every received chunk (e.g. an async
    D2H copy); without a sink rank 0 returns uint8[world, k * witness_bytes], other ranks None.
    The production path keeps witnesses on the GPU that made them (one PCIe link per GPU beats the root's
    single link); this exists to measure and to serve callers that want one buffer."""
    flat = local_wtns.reshape(-1)
    total = flat.numel()
    if world == 1:
        if sink is not None:
            sink(0, 0, flat)
            return None
        return flat.view(1, total)
    out = None
    if rank == 0 and sink is None:
        out = torch.empty((world, total), dtype=torch.uint8, device=flat.device)
    ring = [torch.empty(min(chunk_bytes, total), dtype=torch.uint8, device=flat.device) for _ in range(world)] if rank == 0 else None
    for off in range(0, total, chunk_bytes):
        n = min(chunk_bytes, total - off)
        piece = flat[off:off + n].contiguous()
        bufs = [b[:n] for b in ring] if rank == 0 else None
        dist.gather(piece, bufs, dst=0)
        if rank == 0:
            for r in range(world):
                if sink is not None:
                    sink(r, off, bufs[r])
                else:
                    out[r, off:off + n] = bufs[r]
    return out
