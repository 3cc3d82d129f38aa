#!/usr/bin/env python3
"""Diagnostic build only: the two wave roles of gemm_persistent_kernel (loader waves / MFMA waves) must execute the same number
of workgroup barriers.  Build with `python -m dl_vqa_amd.build --diag`, then
    VQA_LIB=dl_vqa_amd/libvqa_hip_diag.so VQA_PERSISTENT=1 python tools/diag_barriers.py
runs persistent GEMMs at several tile counts (one tile per workgroup, several tiles per workgroup, a grid larger than the
resident slots) and asserts barriers-per-wave(loader) == barriers-per-wave(MFMA) == 1 + tiles * K-steps."""
import ctypes
import faulthandler
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
faulthandler.dump_traceback_later(60, exit=True)
import torch  # noqa: E402
from dl_vqa_amd import _lib, ops  # noqa: E402

lib = _lib.load()
setbuf = lib.vqa_diag_barrier_buffer
setbuf.argtypes = [ctypes.c_void_p]
buf = torch.zeros(4, dtype=torch.int64, device="cuda")
setbuf(buf.data_ptr())
for (M, N, K, div) in ((1352, 1024, 256, 676), (5408, 1024, 256, 676), (173056, 1024, 256, 676), (4096, 4096, 512, 4096)):
    buf.zero_()
    A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda")
    rg = torch.randn((M + div - 1) // div, N, device="cuda")
    C = torch.empty(M, N, device="cuda")
    ops.gemm(A, W, C, M, N, K, rowgroup=rg, rg_div=div, relu=True)
    torch.cuda.synchronize()
    lb, lw, mb, mw = [int(x) for x in buf.tolist()]
    assert lw > 0 and mw > 0, "no persistent launch was counted (VQA_PERSISTENT=1?)"
    assert lb * mw == mb * lw, f"barrier counts differ: loader {lb}/{lw} waves, MFMA {mb}/{mw} waves"
    print(f"M={M} N={N} K={K}: {lb // lw} barriers per loader wave == {mb // mw} per MFMA wave (waves {lw} / {mw}) ok", flush=True)
