"""The sanitizer and debug-bounds variants of the library build (SURVEY.md section 5 "race detection / sanitizers";
tools/asan.sh): host side with -fsanitize=address,undefined (lib_asan/), kernels with -DPMAF_DEBUG_BOUNDS
(lib_bounds/). Running them needs the GPU box (`gpurun -- bash tools/asan.sh run`, result in profiles/r3_asan.txt);
this test only makes sure both variants keep compiling and export the whole C-ABI."""
import ctypes as C
import os
import subprocess

import conftest


def test_asan_and_debug_bounds_variants_build(pmaf):
    subprocess.run(["bash", os.path.join(conftest.ROOT, "tools", "asan.sh"), "build"], check=True, capture_output=True, timeout=900)
    pkg = os.path.join(conftest.ROOT, "predictive-multi-agent-framework_amd")
    nm = subprocess.run(["nm", "-D", "--undefined-only", os.path.join(pkg, "lib_asan", "libpmaf_hip.so")],
                        check=True, capture_output=True, text=True).stdout
    assert "__asan_" in nm and "__ubsan_" in nm            # the host objects really are instrumented
    lib = C.CDLL(os.path.join(pkg, "lib_bounds", "libpmaf_hip.so"))
    for name in pmaf.SYMBOLS:
        assert hasattr(lib, name), name
    dis = subprocess.run(["strings", "-n", "6", os.path.join(pkg, "lib_bounds", "libpmaf_hip.so")], capture_output=True, text=True).stdout
    assert "k_rollout_w64" in dis
