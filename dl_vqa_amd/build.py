"""Build libvqa_hip.so (gfx950) in-tree with hipcc.  `python -m dl_vqa_amd.build [--force]`."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libvqa_hip.so")
# a source, or (source, object suffix, extra flags): gemm.hip and gemm_x3.hip are compiled in parts (host side + kernels by tile
# shape / operand layout: the fused epilogue's straight-line variants compile slowly), the slowest units first
SOURCES = ([("gemm.hip", f"_p{k}", [f"-DVQA_GEMM_PART={k}"]) for k in (2, 3, 4, 5)]
           + [("gemm_x3.hip", f"_p{k}", [f"-DVQA_GEMM_PART={k}"]) for k in (1, 2, 3, 4)]
           + ["bf16.hip", ("gemm.hip", "_p1", ["-DVQA_GEMM_PART=1"]), "conv.hip", "conv_x3.hip", "conv_bf16.hip", "conv_patch_bf16.hip", "conv_patch_f32.hip", "gemm_tall_bf16.hip", "gemm.hip",
              "gemm_x3.hip", "conv0.hip", "lstm.hip", "elementwise.hip", "conv_generic.hip"])
HEADERS = ["common.hpp", "gemm_core.hpp", "gemm_epilogue.hpp", "bf16_core.hpp", "x3_core.hpp", "conv_device.inc", "conv_host.inc", "conv_bf16.inc", os.path.join("..", "..", "include", "vqa_hip.h")]
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wno-unused-result"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def _deps(src: str, seen=None) -> list:
    """The quoted #include closure of a source (csrc/ and include/): an object is rebuilt when one of THESE changes, not when
    any header of the library does."""
    import re
    seen = set() if seen is None else seen
    out = []
    try:
        text = open(src).read()
    except OSError:
        return out
    for inc in re.findall(r'^\s*#\s*include\s+"([^"]+)"', text, flags=re.M):
        path = os.path.normpath(os.path.join(os.path.dirname(src), inc))
        if path not in seen and os.path.exists(path):
            seen.add(path)
            out.append(path)
            out += _deps(path, seen)
    return out


def build_library(force: bool = False, verbose: bool = True, diag: bool = False) -> str:
    """Compile every .hip source for gfx950 and link the shared library. Returns its path.
    diag=True builds libvqa_hip_diag.so with -DVQA_DIAG (in-kernel s_memtime stamps; tools/ only)."""
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs = []
    jobs = []
    lib = LIB.replace(".so", "_diag.so") if diag else LIB
    flags = FLAGS + (["-DVQA_DIAG"] if diag else [])
    for entry in SOURCES:
        src, suffix, extra = entry if isinstance(entry, tuple) else (entry, "", [])
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace(".hip", suffix + ("_diag.o" if diag else ".o")))
        objs.append(o)
        if force or _stale(o, [s] + (_deps(s) or hdrs)):
            jobs.append([_hipcc()] + flags + extra + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + r.stdout + r.stderr)
        return r

    with ThreadPoolExecutor(max_workers=8) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(lib, objs):
        run([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    return lib


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, diag="--diag" in sys.argv))
