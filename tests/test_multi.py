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


def _stub_lib():
    """tests/native/librccl_stub.so: the six RCCL entry points over hipMemcpyAsync (built by __graft_entry__.build())."""
    import subprocess
    so = os.path.join(ROOT, "tests", "native", "librccl_stub.so")
    src = os.path.join(ROOT, "tests", "native", "rccl_stub.cpp")
    if not os.path.exists(so) or os.path.getmtime(src) > os.path.getmtime(so):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-shared", "-fPIC", "-x", "hip",
                               "--offload-arch=gfx950", src, "-o", so])
    return so


@pytest.mark.gpu
@pytest.mark.parametrize("n", [11, 1, 2])
def test_multi_two_shards_on_one_gpu_gather_through_the_rccl_entry_points(n, monkeypatch):
    """The n_dev > 1 branch of zkwg_calculate_batch_multi (a host thread per shard, ncclGroupStart / ncclRecv /
    ncclSend / ncclGroupEnd gather of the result table on devices[0]) with devices = [0, 0] and ZKWG_RCCL_LIB
    pointing at the hipMemcpyAsync stand-in: uneven shards (11 -> 6 + 5), an empty shard (1 -> 1 + 0) and
    one email each.  Witnesses, statuses and the gathered table equal the single-device path."""
    import zkwg
    so = _stub_lib()
    monkeypatch.setenv("ZKWG_RCCL_LIB", so)
    case = json.load(open(os.path.join(ROOT, "tests", "golden", "ev_576_192_case.json")))
    N, M = case["maxHeader"], case["maxBody"]
    mc = zkwg.MultiCircuit([0, 0], main_kind=zkwg.MAIN_EMAIL_VERIFIER, max_header=N, max_body=M)
    assert mc.n_devices == 2
    c = zkwg.Circuit(zkwg.MAIN_EMAIL_VERIFIER, max_header=N, max_body=M, device=0)
    good = c.pack(case["input"])
    bad_inp = dict(case["input"], emailHeader=list(case["input"]["emailHeader"]))
    bad_inp["emailHeader"][10] = str(int(bad_inp["emailHeader"][10]) ^ 1)
    bad = c.pack(bad_inp)
    # tampered emails on both sides of the shard boundary (n = 11: shards [0,6) and [6,11))
    recs = b"".join(bad if i in (2, 7) else good for i in range(n))
    expect = [4 if i in (2, 7) else 0 for i in range(n)]
    stub = C.CDLL(so)
    stub.zk_stub_pairs.restype = C.c_ulong
    pairs0 = stub.zk_stub_pairs()
    wit, status, rows = mc.calculate_batch_host(recs, max_tile=4)
    wit1, status1 = c.calculate_batch_host(recs)
    assert status == status1 == expect
    wb = c.witness_bytes
    for i in range(n):
        assert rows[i][0] == expect[i]
        if expect[i] == 0:
            assert wit[i * wb:(i + 1) * wb] == wit1[i * wb:(i + 1) * wb]
            assert hashlib.sha256(wit[i * wb:(i + 1) * wb]).hexdigest() == case["witnessSha256"]
            assert rows[i] == (0, int(case["pubkeyHash"]), int(case["shaHi"]), int(case["shaLo"]))
    # the rows of shard 1 really travelled through the send/recv pair (none for an empty second shard)
    assert stub.zk_stub_pairs() - pairs0 == (1 if n >= 2 else 0)
    # table only
    _, st2, rows2 = mc.calculate_batch_host(recs, want_witness=False)
    assert st2 == status and rows2 == rows


def test_multi_create_reports_a_missing_rccl_library(monkeypatch):
    """n_dev > 1 with ZKWG_RCCL_LIB naming a file that does not exist: a clean error, no crash (CPU: creation
    already fails for lack of a GPU; both are ZkwgError)."""
    import zkwg
    monkeypatch.setenv("ZKWG_RCCL_LIB", "/nonexistent/librccl.so")
    with pytest.raises(zkwg.ZkwgError):
        zkwg.MultiCircuit([0, 0], main_kind=zkwg.MAIN_EMAIL_VERIFIER, max_header=576, max_body=192)


@pytest.mark.parametrize("world", [2, 8])
def test_bench_launches_its_own_ranks_without_torchrun(world):
    """`python bench.py --gpus N` (N = 2, and 8: the driver's scaling run) with no launcher on the command line and no WORLD_SIZE in the environment: bench.py
    starts two ranks under torch.distributed.run itself (free port, 127.0.0.1) -- checked here on CPU with the
    rendezvous-only mode (gloo); the GPU test below runs the real workload the same way."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--launch-check"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    assert json.loads(line) == {"launch_check": world, "rank_sum": world * (world + 1) // 2, "local_rank_env": 0}
    # the proving workload launches the same way (the flags travel through the self-launch)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--prove", "1", "--prove-batch", "6", "--launch-check"], cwd=ROOT,
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])["launch_check"] == world
    if world != 2:
        return
    # a launcher whose world size disagrees with --gpus is still refused, with a message that says what to do
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"], cwd=ROOT,
                       env=dict(env, WORLD_SIZE="1", RANK="0"), capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "launches the N ranks itself" in r.stderr


@pytest.mark.gpu
def test_bench_two_ranks_on_one_gpu_over_gloo():
    """bench.py's N > 1 path end to end on one GPU: two torch.distributed ranks forced onto cuda:0, gloo instead of
    RCCL (two ranks cannot share a GPU under RCCL), result-table gather + 2 gathered witnesses per rank and step
    inside the timed region.  The driver's 8-GPU run must not be the first execution of this code."""
    import subprocess
    import sys
    # plain `python bench.py --gpus 2` (what the driver's recorded N = 1 command line looks like with N changed): bench.py
    # re-launches itself under torch.distributed.run, so this covers the launcher and the ranks' code path at once
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(ZKWG_BENCH_FORCE_DEVICE="0", ZKWG_BENCH_BACKEND="gloo")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "512", "--steps", "1",
           "--warmup", "1", "--gather-wtns", "2", "--distinct", "64", "--cpu-sample", "0",
           "--place-ring", "0"]      # (two ranks share ONE GPU here: 2 x 7 candidate tiles of 29 GB would not fit it)
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == 2 and res["steps"] == 1 and res["scaling"] == "weak"
    assert res["config"]["batch_per_gpu"] == 512
    assert res["value"] > 0
    assert res["gathered_table"] == {"rows": 1024, "status_nonzero": 0, "rows_with_outputs": 1024}
    assert "2 wtns/rank/step gathered" in res["config"]["parallelism"]


@pytest.mark.gpu
def test_bench_proofs_shard_across_ranks_and_equal_the_single_rank_ones():
    """`bench.py --gpus 2 --prove 1`: proofs shard like witnesses -- rank r proves its contiguous range of the job, every
    rank uploads the key once, rank 0 gathers status + proof rows.  Two ranks forced onto cuda:0 (gloo) with 2 emails each give the SAME
    gathered rows (sha256) as one rank with 4: inputs and blinding are functions of the global email index."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--prove", "1", "--max-header", "576", "--max-body", "192", "--body-len", "100", "--steps", "1",
            "--warmup", "0", "--prove-slots", "2"]
    out = {}
    for world, pb in ((1, 4), (2, 2)):
        e = dict(env)
        if world > 1:
            e.update(ZKWG_BENCH_FORCE_DEVICE="0", ZKWG_BENCH_BACKEND="gloo")
        r = subprocess.run(base + ["--gpus", str(world), "--prove-batch", str(pb)], cwd=ROOT, env=e, capture_output=True, text=True, timeout=1200)
        assert r.returncode == 0, r.stderr[-2000:]
        out[world] = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    for world in (1, 2):
        assert out[world]["n_gpus"] == world and out[world]["gathered_rows"] == 4 and out[world]["nonzero_status"] == 0 and out[world]["value"] > 0
    assert out[1]["proofs_sha256"] == out[2]["proofs_sha256"]


@pytest.mark.gpu
def test_resident_pipeline_below_the_boundary_matches_the_host_path():
    """zkwg_calculate_batch_resident: records in, statuses + the result table out, witnesses into the
    handle's placed two-tile ring; the consumer callback sees every tile on the expansion stream.  Statuses and table rows
    equal the host-buffer path's; the tiles the consumer copies out on that stream are the host path's witnesses byte for byte."""
    import ctypes as C
    import zkwg
    from zkwg import synth
    c = zkwg.Circuit(zkwg.MAIN_EMAIL_VERIFIER, max_header=576, max_body=192, device=0)
    n = 23
    recs, _ = synth.packed_batch(c, seed=0x77, n=n, body_len=70)
    recs = bytearray(recs)
    off = c.lib.zkwg_input_offset(c.h, zkwg._lib.IN_SIGNATURE)
    for bad in (4, 17):                       # tampered signatures on both sides of a tile boundary
        recs[bad * c.in_stride + off] ^= 1
    recs = bytes(recs)
    wit, st_host = c.calculate_batch_host(recs)
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    hip.hipStreamSynchronize.argtypes = [C.c_void_p]
    wb = c.witness_bytes
    seen, got = [], bytearray(n * wb)
    buf = (C.c_uint8 * len(got)).from_buffer(got)

    def consumer(dev, d_tile, stride, first, count, stream):
        assert dev == 0 and stride == wb
        seen.append((first, count))
        # a stream-ordered consumer: copy the tile out ON THE EXPANSION STREAM (2 = hipMemcpyDeviceToHost)
        assert hip.hipMemcpyAsync(C.addressof(buf) + first * wb, d_tile, count * wb, 2, stream) == 0
        assert hip.hipStreamSynchronize(stream) == 0
    status, table = c.calculate_batch_resident(recs, tile=8, prep=16, consumer=consumer)
    assert status == st_host and [i for i, x in enumerate(status) if x] == [4, 17]
    assert seen == [(0, 8), (8, 8), (16, 7)]
    for i in range(n):
        row = table[100 * i:100 * i + 100]
        assert int.from_bytes(row[:4], "little", signed=True) == status[i]
        if status[i] == 0:
            assert row[4:] == wit[i * wb + 32:i * wb + 128]
            assert bytes(got[i * wb:(i + 1) * wb]) == wit[i * wb:(i + 1) * wb]
    pl = c.resident_placement()
    assert pl.get("mode", "").startswith("chunked") or (len(pl["ms_per_tile"]) >= 2 and all(k >= 0 for k in pl["kept"]))
    # a second call reuses the ring; a larger tile re-places it
    status2, _ = c.calculate_batch_resident(recs, tile=16, prep=16, want_table=False)
    assert status2 == status
    # zkwg_resident_release: the handle's buffers go back to the device, the next call allocates them again
    import torch
    free0 = torch.cuda.mem_get_info(0)[0]
    c.release_resident()
    assert torch.cuda.mem_get_info(0)[0] > free0
    c.release_resident()                                    # (idempotent)
    status3, table3 = c.calculate_batch_resident(recs, tile=8, prep=16)
    assert status3 == status and table3 == table
