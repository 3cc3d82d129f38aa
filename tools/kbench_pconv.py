#!/usr/bin/env python3
"""Micro-benchmark of the bf16 convolution kernels on one MI355X: the patch kernels (csrc/conv_patch_bf16.hip) beside
the implicit-GEMM kernels (csrc/conv_bf16.inc) on the conv1 / conv2 shapes of a bench configuration.
    python tools/kbench_pconv.py [--batch 512 --size 448] [--iters 5]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dl_vqa_amd import ops  # noqa: E402

PEAK = 2500.0


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
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--skip-old", action="store_true")
    ap.add_argument("--dbg", default="", help="comma list of VQA_PCONV_DBG values timed beside the plain run, e.g. 0,1,2,3,4")
    args = ap.parse_args()
    B, dev = args.batch, "cuda:0"
    h1 = (args.size - 2) // 2
    h2 = (h1 - 2) // 2

    dbgs = [int(d) for d in args.dbg.split(",")] if args.dbg else [0]

    def run(name, flops, fn):
        # timing experiments in ONE process on one device (VQA_PCONV_DBG is read at every launch): 1 = no epilogue stores,
        # 2 = no DMA after the first stage, 4 = no MFMA
        parts = []
        for d in dbgs:
            os.environ["VQA_PCONV_DBG"] = str(d)
            ms = timeit(fn, args.iters)
            parts.append((d, ms))
        os.environ["VQA_PCONV_DBG"] = "0"
        ms = parts[0][1]
        tf = flops / ms / 1e9
        extra = "  ".join(f"dbg{d}={m:.3f}" for d, m in parts[1:])
        print(f"{name:28s} {ms:9.3f} ms  {tf:8.1f} TF/s  {100 * tf / PEAK:5.1f}% of bf16 MFMA peak  {extra}", flush=True)

    for l, (Hin, Ci, Co) in enumerate(((h1, 64, 128), (h2, 128, 256)), 1):
        x = torch.randn(B, Hin, Hin, Ci, device=dev).to(torch.bfloat16)
        w = torch.randn(Co, Ci, 3, 3, device=dev) * (9 * Ci) ** -0.5
        b = torch.zeros(Co, device=dev)
        Ho = Hin - 2
        flops = 2.0 * B * Ho * Ho * Co * 9 * Ci
        last = l == 2
        od = torch.float32 if last else torch.bfloat16
        wf, wd = ops.pconv_pack_weights(w)
        xc = ops.to_c16(x)
        pooled, am = ops.pconv_fwd(xc, wf, b, Co, out_dtype=od, tag=l)
        run(f"conv{l} pconv fwd", flops, lambda: ops.pconv_fwd(xc, wf, b, Co, out_dtype=od, tag=l))
        dp = torch.randn(am.shape, device=dev).to(torch.bfloat16)          # C16, like the arg-max bytes
        run(f"conv{l} pconv dgrad", flops * (Hin * Hin) / (Ho * Ho),
            lambda: ops.pconv_dgrad(dp, am, wd, x.shape, out_c16=l > 1, tag=l))
        if ops.pconv_wgrad_supported(Hin, Hin, Ci, Co):
            dw, db = torch.empty_like(w), torch.empty_like(b)
            run(f"conv{l} pconv wgrad", flops, lambda: ops.pconv_wgrad(xc, dp, am, dw, db, tag=l))
        if not args.skip_old:
            wfT, wdT = ops.conv_pack_weights_bf16(w, Ci)
            run(f"conv{l} implicit-GEMM fwd", flops, lambda: ops.conv_fwd_bf16(x, wfT, b, 1, out_dtype=od, tag=l))
            dpn, amn = ops.from_c16(dp), ops.from_c16(am)
            run(f"conv{l} implicit-GEMM dgrad", flops, lambda: ops.conv_dgrad_bf16(dpn, amn, wdT, x.shape, 1, tag=l))
            dw, db = torch.empty_like(w), torch.empty_like(b)
            run(f"conv{l} implicit-GEMM wgrad", flops, lambda: ops.conv_wgrad_bf16(x, dpn, amn, dw, db, 1, tag=l))
        del x, xc, dp, pooled, am
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
