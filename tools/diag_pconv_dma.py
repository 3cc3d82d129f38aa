#!/usr/bin/env python3
"""Diagnostic: does the patch DMA run faster when a pixel's K-slice is contiguous with its neighbours'?  Forward patch
kernel on 223x223 maps with Cin = 16 (a 16-channel slice IS the whole pixel: patch rows are contiguous runs) against
Cin = 64 / 128 (32-byte fragments of 128- / 256-byte pixels), per STAGE, with VQA_PCONV_DBG from the environment."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dl_vqa_amd import ops  # noqa: E402

dev = "cuda:0"
B, H, Co = 512, 223, 128
for Ci in (16, 64, 128):
    x = torch.randn(B, H, H, Ci, device=dev).to(torch.bfloat16)
    w = torch.randn(Co, Ci, 3, 3, device=dev) * (9 * Ci) ** -0.5
    b = torch.zeros(Co, device=dev)
    wf, _ = ops.pconv_pack_weights(w, need_wd=False)
    ops.pconv_fwd(x, wf, b, Co)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(3):
        ops.pconv_fwd(x, wf, b, Co)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 3
    stages = B * 14 * 7 * (Ci // 16) / 256
    print(f"Cin={Ci:4d}: {ms:7.3f} ms, {stages:6.0f} stages per workgroup, {ms * 1e3 / stages:6.2f} us per stage", flush=True)
    del x
