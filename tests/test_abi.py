"""The C-ABI shared library: loads, exports every symbol include/pmaf.h
declares, validates arguments, and fails loudly (no CPU fallback) when no HIP
device is present. No compute calls here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import conftest


def _declared_symbols():
    hdr = open(os.path.join(conftest.ROOT, "include", "pmaf.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(pmaf_[a-z0-9_]+)\s*\(", hdr)))


def test_every_declared_symbol_is_exported_and_bound(pmaf, hip_lib):
    declared = _declared_symbols()
    assert len(declared) >= 30
    raw = C.CDLL(pmaf.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), "libpmaf_hip.so does not export %s" % name
    assert sorted(pmaf.SYMBOLS) == declared, "planner.py binding table out of sync with include/pmaf.h"
    assert hip_lib.pmaf_abi_version() == 7


def test_params_struct_matches_header(pmaf):
    # 8 int32 + 6 doubles + 9 pointers, naturally aligned
    assert C.sizeof(pmaf.planner.PmafParams) == 8 * 4 + 6 * 8 + 9 * 8


def test_argument_validation_messages(pmaf, hip_lib, scenes):
    h = C.c_void_p()
    rc = hip_lib.pmaf_create(None, C.byref(h))
    assert rc == -1 and b"NULL" in hip_lib.pmaf_last_error()
    prm = pmaf.planner.PmafParams()
    prm.abi_version = 99
    assert hip_lib.pmaf_create(C.byref(prm), C.byref(h)) == -1
    assert b"ABI version" in hip_lib.pmaf_last_error()
    prm.abi_version = hip_lib.pmaf_abi_version()
    prm.n_populations, prm.n_agents, prm.n_obstacles, prm.max_prediction_steps = 1, 4, 0, 10
    assert hip_lib.pmaf_create(C.byref(prm), C.byref(h)) == -1
    assert b"obstacle" in hip_lib.pmaf_last_error()  # empty obstacle list (reference underflows)
    assert hip_lib.pmaf_destroy(None) == 0
    assert hip_lib.pmaf_start(None) == -1


@pytest.mark.skipif(conftest.has_gpu(), reason="checks the no-GPU failure mode")
def test_no_device_fails_loudly_without_cpu_fallback(pmaf, scenes):
    sc = scenes.config_scene("C1")
    with pytest.raises(pmaf.PmafError) as e:
        pmaf.PmafPlanner(sc)
    assert e.value.code == -2 and "no CPU fallback" in str(e.value)


def test_product_sources_do_not_reference_the_oracle():
    """only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may touch oracle/"""
    pkg = os.path.join(conftest.ROOT, "predictive-multi-agent-framework_amd")
    inc = os.path.join(conftest.ROOT, "include")
    for base in (pkg, inc):
        for dp, _, files in os.walk(base):
            for f in files:
                if f.endswith((".py", ".hpp", ".h", ".hip", ".cpp", ".sh")):
                    txt = open(os.path.join(dp, f), errors="ignore").read()
                    for line in txt.splitlines():
                        if re.search(r"^\s*(from|import)\s+oracle|#include\s+\"[^\"]*oracle|libpmaf_oracle|orc_[a-z_]+\(", line):
                            raise AssertionError("%s references the oracle: %s" % (f, line.strip()))


def test_throughput_kernels_keep_two_waves_per_simd(pmaf):
    """k_rollout_grp hides its latencies with a second wave on the SIMD, i.e. it
    must stay within 256 VGPRs (it sits just below; csrc/build.sh records every
    kernel's resources). A variant that tips over runs at half the rate."""
    rec = os.path.join(os.path.dirname(pmaf.LIB_PATH), "resource_usage.txt")
    if not os.path.exists(rec):
        pytest.skip("no resource record next to the library (built by another recipe)")
    kernels = {}
    name = None
    for line in open(rec):
        line = line.strip()
        if line.startswith("Function Name:"):
            name = line.split(":", 1)[1].strip()
            kernels[name] = {}
        elif name and ":" in line:
            k, v = line.rsplit(":", 1)
            kernels[name][k.strip()] = v.strip()
    grp = [k for k in kernels if re.match(r"_Z13k_rollout_grpILi(8|16|32)ELi[12]ELi2EE", k)]
    assert len(grp) == 6, sorted(kernels)
    for k in grp:
        assert int(kernels[k]["Occupancy [waves/SIMD]"]) >= 2, (k, kernels[k])
        assert int(kernels[k]["ScratchSize [bytes/lane]"]) == 0, (k, kernels[k])
    for k in kernels:
        if re.match(r"_Z13k_rollout_w64ILi[12]ELi2ELb[01]ELb[01]EE", k):
            # (reserved stack only; the next test makes sure no instruction uses it)
            # the one-slot kernel holds two loop versions per heuristic (with / without code for the repulsive
            # obstacle); the compiler reserves a few stack slots for it that no instruction uses (next test)
            # (68 bytes per loop version in rounds 1-2; the PLAIN instantiations of round 3 reserve up to 148)
            assert int(kernels[k]["ScratchSize [bytes/lane]"]) <= 192, (k, kernels[k])
    # the waves of k_rollout_mw have a SIMD each: occupancy is not a constraint, scratch is (next test)
    mw = [k for k in kernels if re.match(r"_Z12k_rollout_mwILi[234]ELi[123]ELb[01]ELb[01]EE", k)]
    assert len(mw) == 36, sorted(kernels)


def test_no_kernel_touches_scratch_memory(pmaf, tmp_path):
    """a rollout kernel that spills to scratch runs at a fraction of its speed
    (seen twice while tuning): disassemble the gfx950 code object of every
    kernel translation unit (csrc/build.sh keeps the objects in lib/obj) and make
    sure there is not a single scratch instruction in the library"""
    import glob
    import subprocess
    llvm = "/opt/rocm/lib/llvm/bin"
    tools = [os.path.join(llvm, t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-objdump")]
    if not all(os.path.exists(t) for t in tools):
        pytest.skip("ROCm LLVM tools not available")
    objs = sorted(glob.glob(os.path.join(os.path.dirname(pmaf.LIB_PATH), "obj", "k_*.o")))
    if not objs:
        pytest.skip("no kernel objects next to the library (built by another recipe)")
    assert len(objs) == 12, objs   # w64: m0, m1, m2 (t1 + tn), m3; grp: m0, m2, m3; mw: m1, m2, m3; misc
    n_add = 0
    for k, obj in enumerate(objs):
        fat = str(tmp_path / ("fatbin%d.bin" % k))
        co = str(tmp_path / ("gfx950_%d.o" % k))
        subprocess.run([tools[0], "--dump-section", ".hip_fatbin=" + fat, obj], check=True, capture_output=True)
        subprocess.run([tools[1], "--unbundle", "--type=o", "--input=" + fat,
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], check=True, capture_output=True)
        dis = subprocess.run([tools[2], "-d", co], check=True, capture_output=True, text=True).stdout
        n_add += dis.count("v_add_f64")
        assert dis.count("scratch_") == 0, obj
    assert n_add > 1000          # it is the kernels' code


def test_library_links_rccl_and_the_hip_runtime(pmaf, hip_lib):
    """the multi-GPU entry points are product code: libpmaf_hip.so itself links
    librccl (ncclAllGather is called from inside the library)"""
    import subprocess
    out = subprocess.run(["readelf", "-d", pmaf.LIB_PATH], check=True, capture_output=True, text=True).stdout
    assert "librccl.so" in out and "libamdhip64.so" in out


def test_select_best_is_evaluate_agents_rule(pmaf, hip_lib):
    # CfManager::evaluateAgents, B/src/cf_manager.cpp:336-353
    assert pmaf.select_best([3.0, 2.0, 2.0, 5.0]) == 1                 # first minimum wins ties
    assert pmaf.select_best([3.0, 2.0, 1.9, 5.0], prev_best=1) == 1    # 1.9 !< 0.9 * 2.0
    assert pmaf.select_best([3.0, 2.0, 1.7, 5.0], prev_best=1) == 2
    assert pmaf.select_best([np.inf, np.inf]) == 0


def test_host_communicator_roundtrip_single_rank(pmaf, hip_lib):
    """pmaf_comm_init_host + pmaf_comm_allgather with a one-rank Python transport (no GPU involved)"""
    calls = []

    def ag(send):
        calls.append(send.size)
        return send
    c = pmaf.PmafComm.host(1, 0, ag)
    assert c.world == 1 and c.rank == 0
    a = np.arange(12.0).reshape(4, 3)
    np.testing.assert_array_equal(c.allgather(a), a[None])
    assert calls == [96]
    c.close()
