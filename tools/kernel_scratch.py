#!/usr/bin/env python3
"""Private-segment (scratch) bytes per lane of every gfx950 kernel in a shared library or object file, read from the code
objects' AMDGPU metadata (no GPU needed):
    python tools/kernel_scratch.py [dl_vqa_amd/libvqa_hip.so]
Used by tests/test_abi_cpu.py: workgroup-barrier kernels with a large private segment hang the GPU."""
import os
import re
import struct
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def kernel_scratch(path):
    """{kernel name: (private_segment_fixed_size, vgpr_count, vgpr_spill_count)}"""
    out = {}
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", path, os.path.join(td, "copy")],
                       check=True, capture_output=True)
        data = open(fat, "rb").read()
        magic = b"__CLANG_OFFLOAD_BUNDLE__"
        for bi, m in enumerate(re.finditer(re.escape(magic), data)):
            p = m.start()
            n = struct.unpack_from("<Q", data, p + 24)[0]
            off = p + 32
            for _ in range(n):
                eoff, esize, tlen = struct.unpack_from("<QQQ", data, off)
                off += 24
                triple = data[off:off + tlen].decode()
                off += tlen
                if "gfx950" not in triple or esize == 0:
                    continue
                co = os.path.join(td, f"co{bi}.elf")
                open(co, "wb").write(data[p + eoff:p + eoff + esize])
                notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
                for k in re.finditer(r"\.name:\s+(\S+)\n(.*?)\.wavefront_size", notes, flags=re.S):
                    body = k.group(2)
                    g = lambda key: int(re.search(key + r":\s+(\d+)", body).group(1))
                    out[k.group(1)] = (g(r"\.private_segment_fixed_size"), g(r"\.vgpr_count"), g(r"\.vgpr_spill_count"))
    return out


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                             "dl_vqa_amd", "libvqa_hip.so")
    ks = kernel_scratch(lib)
    print(f"{len(ks)} kernels in {lib}")
    for name, (scr, vg, sp) in sorted(ks.items(), key=lambda kv: -kv[1][0]):
        if scr:
            print(f"{scr:6d} B/lane  vgpr {vg:3d}  spilled {sp:3d}  {name[:110]}")
