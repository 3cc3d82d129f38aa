#!/usr/bin/env python3
"""Micro-benchmark of the tall bf16 GEMM (csrc/gemm_tall_bf16.hip) at the attention image projection's shape of a bench
configuration: x = relu(v . Wv^T + q'), M = batch * positions, N = attention hidden, K = last conv channels.
    python tools/kbench_tall.py [--batch 512 --size 448] [--dbg 0,1,2,4]      (--dbg needs tools/build_tall_diag.sh's library)"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dl_vqa_amd import ops  # noqa: E402


def timeit(fn, iters):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--size", type=int, default=448)
    ap.add_argument("--mid", type=int, default=1024)
    ap.add_argument("--channels", type=int, default=256)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--dbg", default="")
    args = ap.parse_args()
    g = args.size
    for _ in range(3):
        g = (g - 2) // 2
    B, P, K, N, dev = args.batch, g * g, args.channels, args.mid, "cuda:0"
    M = B * P
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    W = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
    rg = torch.randn(B, N, device=dev)
    C = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    flops = 2.0 * M * N * K
    gbytes = (M * K * 2 + M * N * 2) / 1e9
    for d in ([int(x) for x in args.dbg.split(",")] if args.dbg else [0]):
        os.environ["VQA_TALL_DBG"] = str(d)
        ms = timeit(lambda: ops.gemm_tall_bf16(A, W, C, M, N, K, rowgroup=rg, rg_div=P, rg_op=0, relu=True), args.iters)
        print(f"tall GEMM {M}x{N}x{K} dbg={d}: {ms:8.3f} ms  {flops / ms / 1e9:8.1f} TF/s  {gbytes / ms * 1e3:8.1f} GB/s (algorithmic)", flush=True)
    os.environ["VQA_TALL_DBG"] = "0"


if __name__ == "__main__":
    main()
