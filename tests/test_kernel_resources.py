"""Compile-time properties of the streaming kernels that the roofline depends on (checked with the cross-compiler, no GPU):
no scratch memory in any kernel of zkwg_kernels_expand3.hip -- a struct passed by reference to an out-of-line function once
cost the Montgomery kernels 1.39 x their algorithmic HBM writes -- and the register budgets that give
the standard-form kernels 8 wavefronts per SIMD."""
import os
import re
import shutil
import subprocess

import pytest

from conftest import ROOT

CSRC = os.path.join(ROOT, "zk-email-verify_amd", "csrc")


_SOURCES = ("zkwg_kernels_rslb.hip", "zkwg_kernels_expand3.hip", "zkwg_kernels_msm.hip")
_compiles = {}


def _start_compiles(tmpdir):
    """the three cross-compilations of this module run side by side (each is one hipcc process of 30-60 s)"""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    for src in _SOURCES:
        if src not in _compiles:
            err = open(os.path.join(tmpdir, src + ".err"), "w+")
            _compiles[src] = (subprocess.Popen([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-c", os.path.join(CSRC, src), "-o",
                                                os.path.join(tmpdir, src + ".o"), "-Rpass-analysis=kernel-resource-usage"], stdout=subprocess.DEVNULL, stderr=err), err)


def _resource_usage(src, obj):
    import tempfile
    if src not in _compiles:
        _start_compiles(tempfile.mkdtemp(prefix="zkwg_res_"))
    proc, err = _compiles[src]
    assert proc.wait(timeout=900) == 0
    err.seek(0)
    stderr = err.read()
    info, cur = {}, None
    for line in stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = m.group(1)
            info[cur] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
        if m and cur:
            info[cur][m.group(1).strip()] = int(m.group(2))
    return info


@pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")), reason="hipcc not available")
def test_soft_line_break_chunk_hashes_use_no_scratch_memory(tmp_path):
    # zk_rslb_chunks (one Poseidon(16) per lane, 786 k per batch): the dense mixes hold the old state in registers (153 VGPRs) and
    # fetch their 153 table limbs 9 at a time; if the compiler hoists those scalar loads, or does not unroll the 17
    # multiply-accumulates of an output, the state lands in scratch (624 B .. 6 KB per lane were seen on the way)
    info = _resource_usage("zkwg_kernels_rslb.hip", tmp_path / "rslb.o")
    ks = [v for name, v in info.items() if "zk_rslb_chunks" in name]
    assert len(ks) == 8, sorted(info)               # the evaluator's variants (ZKWG_RSLB_V)
    for k in ks:
        assert k.get("ScratchSize") == 0, k
        assert k["LDS Size"] <= 40 * 1024, k       # 4 wavefronts per CU
    # the default (6: dense mixes through the staging area) leaves zk_expand six wavefronts per SIMD beside it
    v6 = next(v for name, v in info.items() if "zk_rslb_chunks_v6" in name)
    assert v6["VGPRs"] + v6.get("AGPRs", 0) <= 128, v6
    assert next(v for name, v in info.items() if "zk_rslb_scan" in name).get("ScratchSize") == 0
    # round 6: the merge chain one lane per email through the same evaluator (t = 3), the constant-chunk passes
    for frag in ("zk_rslb_merge1", "zk_rslb_classify", "zk_rslb_fill_const"):
        assert next(v for name, v in info.items() if frag in name).get("ScratchSize") == 0, frag


@pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")), reason="hipcc not available")
def test_streaming_kernels_use_no_scratch_memory_and_keep_their_occupancy(tmp_path):
    info = _resource_usage("zkwg_kernels_expand3.hip", tmp_path / "x3.o")
    kernels = {k: v for k, v in info.items() if "zk_" in k}
    assert len(kernels) >= 20, sorted(info)
    for k, v in kernels.items():
        assert v.get("ScratchSize") == 0, (k, v)
    by = lambda frag: next(v for k, v in kernels.items() if frag in k)
    # the headline kernel: <= 64 VGPRs = 8 wavefronts per SIMD; the descriptor-driven one: below
    for frag in ("13zk_expand3_k4", "13zk_expand3_k2"):
        assert by(frag)["VGPRs"] <= 64 and by(frag)["Occupancy"] == 8, (frag, by(frag))
    # the descriptor-driven kernel at its default (K = 1, software-pipelined over the emails of its group): 8 wavefronts per SIMD
    assert by("17zk_expand3_o0p_k1")["VGPRs"] <= 64 and by("17zk_expand3_o0p_k1")["Occupancy"] == 8, by("17zk_expand3_o0p_k1")
    assert by("13zk_o0_rows_fr")["Occupancy"] >= 4, by("13zk_o0_rows_fr")


@pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")), reason="hipcc not available")
def test_multi_exponentiation_kernels_use_no_scratch_memory(tmp_path):
    """every zk_msm_* kernel of the prover, G1 and G2 (the round-5 G2 kernels -- one lane per point, Fq2 products as
    function calls -- sat at 256 VGPRs + 1,168 bytes of scratch, occupancy 1).  Since round 6 G2 runs on lane pairs in lazy limb form:
    no scratch anywhere, the kernels of the mixed additions (level-0 slices, the ones' sums) at 4 wavefronts per SIMD for G1 and 3 for G2
    (no software prefetch of the next base: same or better measured rate, profiles/r06/r06_l, r06_s).  zk_msm_table (once per key: K shifted copies of the bases, canonical-word arithmetic with
    one inversion per copy) is exempt."""
    info = _resource_usage("zkwg_kernels_msm.hip", tmp_path / "msm.o")
    ks = {n: v for n, v in info.items() if "zk_msm_" in n and "zk_msm_table" not in n}
    assert len(ks) >= 30, sorted(info)
    for n, v in ks.items():
        assert v.get("ScratchSize") == 0, (n, v)
    hot1 = [v for n, v in ks.items() if "ZkEcG1" in n and ("slice_sumI6ZkEcG1Lb1" in n or "zk_msm_onesI" in n)]
    hot2 = [v for n, v in ks.items() if "ZkEcG2" in n and ("slice_sumI6ZkEcG2Lb1" in n or "zk_msm_onesI" in n)]
    assert len(hot1) == 2 and len(hot2) == 2
    assert all(v["Occupancy"] >= 4 for v in hot1), hot1
    assert all(v["Occupancy"] >= 3 for v in hot2), hot2
    assert all(v["Occupancy"] >= 2 for n, v in ks.items()), ks
