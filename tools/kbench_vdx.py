#!/usr/bin/env python3
"""Times the v_conv backward-data product of the bf16 path (dx' [B*P][mid] bf16 x W [mid][C] bf16 -> fp32 [B*P][C]) at the
configs[3] shape; run once with VQA_GEMM_WIDE=0 and once without to compare the 128 x 128 and 128 x 256 tilings."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dl_vqa_amd import ops

M, C, mid = 512 * 2916, 256, 1024
dev = "cuda:0"
dx = torch.randn(M, mid, device=dev).to(torch.bfloat16)
w = torch.randn(mid, C, device=dev).to(torch.bfloat16)
out = torch.empty(M, C, device=dev)
fn = lambda: ops.gemm_bf16(dx, w, out, M, C, mid, transB=False, lda=mid, ldb=C)
fn(); torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10):
    fn()
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / 10
print(f"v_conv dX bf16 VQA_GEMM_WIDE={os.environ.get('VQA_GEMM_WIDE', '1')}: {ms:.3f} ms  {2.0 * M * C * mid / ms / 1e9:.0f} TF/s  "
      f"{(dx.numel() * 2 + out.numel() * 4) / ms / 1e9:.2f} TB/s compulsory")
