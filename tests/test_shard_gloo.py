"""N>1 path on CPU: 2 processes, gloo backend -- shard ranges and the result-table gather."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, q):
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "zk-email-verify_amd", "py"))
    from zkwg import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard.shard_range(n_total, rank, world)
    n = hi - lo
    status = torch.tensor([4 if (i % 5 == 0) else 0 for i in range(lo, hi)], dtype=torch.int32)
    rows = torch.zeros((n, 128), dtype=torch.uint8)
    for k, i in enumerate(range(lo, hi)):
        rows[k, 32:128] = torch.tensor([(i * 7 + j) % 251 for j in range(96)], dtype=torch.uint8)
    table = shard.result_table(status, rows)
    full = shard.gather_table(dist, table, n_total, rank, world)
    # bulk witness gather in chunks (3 fake witnesses of 1000 bytes per rank, 256-byte chunks)
    wt = torch.tensor([[(rank * 31 + k * 7 + j) % 253 for j in range(1000)] for k in range(3)], dtype=torch.uint8)
    allw = shard.gather_witnesses(dist, wt, rank, world, chunk_bytes=256)
    got = []
    shard.gather_witnesses(dist, wt, rank, world, chunk_bytes=512, sink=lambda r, off, t: got.append((r, off, t.clone())))
    # the prover's rows (status + 256-byte proof per email) through gather_rows: ragged shards, order by global index
    prow = torch.tensor([[(i * 13 + j) % 249 for j in range(260)] for i in range(lo, hi)], dtype=torch.uint8).view(n, 260)
    allp = shard.gather_rows(dist, prow, n_total, rank, world)
    if rank == 0:
        assert allp.shape == (n_total, 260) and all(allp[i].tolist() == [(i * 13 + j) % 249 for j in range(260)] for i in range(n_total))
    else:
        assert allp is None
    if rank == 0:
        assert allw.shape == (world, 3000)
        for r in range(world):
            assert allw[r].tolist() == [(r * 31 + k * 7 + j) % 253 for k in range(3) for j in range(1000)]
        assert sorted((r, off) for r, off, _ in got) == [(r, off) for r in range(world) for off in range(0, 3000, 512)]
        assert all(t.tolist() == allw[r, off:off + 512].tolist() for r, off, t in got)
        q.put(full.numpy().tobytes())
    else:
        assert allw is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather_is_ordered_and_complete():
    from zkwg import shard
    n_total, world = 11, 2
    assert shard.shard_range(n_total, 0, 2) == (0, 6) and shard.shard_range(n_total, 1, 2) == (6, 11)
    assert [shard.shard_range(32768, r, 8) for r in range(8)][-1] == (28672, 32768)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    blob = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert len(blob) == n_total * shard.TABLE_BYTES
    for i in range(n_total):
        row = blob[i * shard.TABLE_BYTES:(i + 1) * shard.TABLE_BYTES]
        assert int.from_bytes(row[:4], "little") == (4 if i % 5 == 0 else 0)
        assert list(row[4:]) == [(i * 7 + j) % 251 for j in range(96)]
