"""Multi-GPU entry of the C-ABI (include/zkwg.h zkwg_multi_*): contiguous shards, one
handle + host thread per device, RCCL gather of the 100-byte result table on devices[0].
CPU: the sharding arithmetic (the same ranges as the torch.distributed path, zkwg.shard.shard_range) and
the error path of a handle without GPUs.  GPU: n_dev = 1 parity with the single-device entry points
(8-GPU runs are the driver's; the RCCL branch is exercised there through bench.py / this API)."""
import ctypes as C
import hashlib
import json
import os

import pytest

from conftest import ROOT


def test_shard_ranges_tile_the_batch_like_the_distributed_path():
    import zkwg
    from zkwg import shard
    for n in (0, 1, 7, 256, 4096, 32768, 32771):
        for world in (1, 2, 3, 8):
            pos = 0
            sizes = []
            for r in range(world):
                first, count = zkwg.shard_range(n, world, r)
                assert first == pos
                assert (first, first + count) == shard.shard_range(n, r, world)
                pos += count
                sizes.append(count)
            assert pos == n and max(sizes) - min(sizes) <= 1
    assert zkwg.shard_range(10, 2, 5) == (0, 0)      # out-of-range shard index: empty


def test_multi_create_without_a_gpu_fails_cleanly():
    import torch
    import zkwg
    if torch.cuda.is_available():
        pytest.skip("needs a box without GPUs")
    with pytest.raises(zkwg.ZkwgError):
        zkwg.MultiCircuit([0, 1], main_kind=zkwg.MAIN_EMAIL_VERIFIER, max_header=576, max_body=192)


@pytest.mark.gpu
def test_multi_one_device_equals_single_device_path_and_table():
    import zkwg
    case = json.load(open(os.path.join(ROOT, "tests", "golden", "ev_576_192_case.json")))
    N, M = case["maxHeader"], case["maxBody"]
    mc = zkwg.MultiCircuit([0], main_kind=zkwg.MAIN_EMAIL_VERIFIER, max_header=N, max_body=M)
    assert mc.n_devices == 1
    c = zkwg.Circuit(zkwg.MAIN_EMAIL_VERIFIER, max_header=N, max_body=M, device=0)
    good = c.pack(case["input"])
    bad_inp = dict(case["input"], emailHeader=list(case["input"]["emailHeader"]))
    bad_inp["emailHeader"][10] = str(int(bad_inp["emailHeader"][10]) ^ 1)
    recs = good * 3 + c.pack(bad_inp) + good * 2
    wit, status, rows = mc.calculate_batch_host(recs, max_tile=2)       # several tiles
    wit1, status1 = c.calculate_batch_host(recs)
    assert status == status1 == [0, 0, 0, 4, 0, 0]
    wb = c.witness_bytes
    for i in (0, 1, 2, 4, 5):
        assert wit[i * wb:(i + 1) * wb] == wit1[i * wb:(i + 1) * wb]
        assert hashlib.sha256(wit[i * wb:(i + 1) * wb]).hexdigest() == case["witnessSha256"]
        assert rows[i] == (0, int(case["pubkeyHash"]), int(case["shaHi"]), int(case["shaLo"]))
    assert rows[3][0] == 4
    # table only (no witnesses delivered)
    _, st2, rows2 = mc.calculate_batch_host(recs, want_witness=False)
    assert st2 == status and rows2[0] == rows[0]
