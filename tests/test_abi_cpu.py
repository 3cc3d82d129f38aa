"""CPU: the C-ABI library builds, loads, and exports exactly what include/vqa_hip.h declares."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "vqa_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vqa_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_bound_and_exported():
    from dl_vqa_amd import _lib, build
    build.build_library(verbose=False)
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 30
    assert set(names) == set(_lib.PROTOTYPES), set(names) ^ set(_lib.PROTOTYPES)
    for n in names:
        assert hasattr(lib, n), f"{n} not exported by libvqa_hip.so"
    assert lib.vqa_abi_version() == _lib.header_abi_version() >= 3


def test_argument_validation_without_gpu():
    """Host-side checks run before any HIP call, so they are testable on a CPU-only box."""
    from dl_vqa_amd import _lib
    lib = _lib.load()
    rc = lib.vqa_gemm(None, 4, 0, None, 4, 1, None, 4, 4, 4, 4, None, None, None, 0, 1, 0, 0, 0, None, None, 0, 0, None)
    assert rc == 1 and b"null operand" in lib.vqa_last_error()
    rc = lib.vqa_conv3x3_relu_pool_fwd(16, 16, 16, 16, 16, 1, 8, 8, 6, 8, 1, 0, None)   # CiP = 6
    assert rc == 1 and b"multiples of 4" in lib.vqa_last_error()
    rc = lib.vqa_att_score_fwd(16, 0, 16, 8, 16, 16, 1, 4, 8, 9, 0.0, 0, None, None)     # G = 9
    assert rc == 1 and b"glimpses" in lib.vqa_last_error()
    assert lib.vqa_gemm_workspace_bytes(256, 1024, 2560) > 0       # split-K plan for a skinny GEMM
    assert lib.vqa_gemm_workspace_bytes(4096, 4096, 64) == 0


def test_missing_library_fails_loudly(monkeypatch):
    from dl_vqa_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libvqa_hip.so")
    with pytest.raises(_lib.VqaHipError):
        _lib.load()


def test_no_kernel_carries_a_large_private_segment():
    """a workgroup-barrier kernel with 512 bytes of scratch per lane deadlocked the GPU beyond ~2 000 resident
    waves (the round-2 hang, reproduced and bisected in round 3).  Read every gfx950 kernel's private segment size from the
    code objects of the built library: none may exceed the 256 bytes per lane the launch-time guard (ensure_dyn_smem) allows."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from kernel_scratch import kernel_scratch
    from dl_vqa_amd import _lib, build
    build.build_library(verbose=False)
    ks = kernel_scratch(_lib.LIB_PATH)
    assert len(ks) > 200
    worst = max(ks.items(), key=lambda kv: kv[1][0])
    print(f"largest private segment: {worst[1][0]} bytes per lane ({worst[0][:80]})")
    big = {k: v for k, v in ks.items() if v[0] > 256}
    assert not big, big
