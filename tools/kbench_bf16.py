#!/usr/bin/env python3
"""Per-kernel timing of the bf16 convolution / GEMM kernels at the BASELINE configs[3] shapes (diagnostic).
usage: python tools/kbench_bf16.py [B] [S]     (defaults 256 224; configs[3] is 512 448)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dl_vqa_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
S = int(sys.argv[2]) if len(sys.argv) > 2 else 224
PEAK = 2500.0
dev = "cuda:0"


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def run(name, flops, fn, bytes_=0):
    t = timeit(fn)
    print(f"{name:28s}{t*1e3:9.3f} ms {flops/t/1e12:9.1f} TF/s {100*flops/t/1e12/PEAK:6.1f}% of bf16 peak"
          + (f"   {bytes_/t/1e12:5.2f} TB/s compulsory" if bytes_ else ""), flush=True)


H0 = (S - 2) // 2                     # pooled size after conv0
shapes = [("conv1", 64, 128, H0), ("conv2", 128, 256, (H0 - 2) // 2)]
for name, Ci, Co, H in shapes:
    Hp = (H - 2) // 2
    x = (torch.randn(B, H, H, Ci, device=dev) * 0.5).to(torch.bfloat16)
    w = torch.randn(Co, Ci, 3, 3, device=dev) / (3 * Ci ** 0.5)
    bias = torch.zeros(Co, device=dev)
    wfT, wdT = ops.conv_pack_weights_bf16(w, Ci)
    flops = 2.0 * B * (H - 2) ** 2 * Co * 9 * Ci
    pooled, am = ops.conv_fwd_bf16(x, wfT, bias, 1)
    dp = (torch.randn(B, Hp, Hp, Co, device=dev)).to(torch.bfloat16)
    dw, db = torch.empty_like(w), torch.empty_like(bias)
    xb, pb = x.numel() * 2, pooled.numel() * 3
    run(f"{name}_fwd_bf16", flops, lambda: ops.conv_fwd_bf16(x, wfT, bias, 1), xb + pb)
    run(f"{name}_dgrad_bf16", flops, lambda: ops.conv_dgrad_bf16(dp, am, wdT, x.shape, 1), xb + pb)
    run(f"{name}_wgrad_bf16", flops, lambda: ops.conv_wgrad_bf16(x, dp, am, dw, db, 1), xb + pb)
    del x, pooled, am, dp
for (M, N, K) in [(B * ((shapes[1][3] - 2) // 2) ** 2, 1024, 256), (4096, 4096, 4096)]:
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    W = torch.randn(N, K, device=dev).to(torch.bfloat16)
    C = torch.empty(M, N, device=dev)
    run(f"gemm_bf16 {M}x{N}x{K} NT", 2.0 * M * N * K, lambda: ops.gemm_bf16(A, W, C, M, N, K))
    if M == 4096:
        run(f"gemm_bf16 {M}x{N}x{K} TN", 2.0 * M * N * K, lambda: ops.gemm_bf16(A, W, C, M, N, K, transA=True, transB=False))
