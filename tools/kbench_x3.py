#!/usr/bin/env python3
"""conv1 / conv2 forward, dgrad, wgrad at the headline shape: fp32 MFMA kernels against the 3 x bf16 split kernels, the
latter with fp32 operands (split in the loaders) and with x3-packed operands (HIP events on the launch stream;
fp32-equivalent TFLOP/s, % of the fp32 MFMA peak 157.3 and of 2500/6 = 416.7).  --gemm: the three v_conv products."""
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
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--size", type=int, default=224)
    args = ap.parse_args()
    B, dev = args.batch, "cuda:0"
    chans = [64, 128, 256]
    Hin = (args.size - 2) // 2
    x = torch.randn(B, Hin, Hin, 64, device=dev)
    for l in (1, 2):
        Ci, Co = chans[l - 1], chans[l]
        w = torch.randn(Co, Ci, 3, 3, device=dev) * (1.0 / (9 * Ci) ** 0.5)
        b = torch.zeros(Co, device=dev)
        wf, wd = ops.conv_pack_weights(w, Ci)
        wfx, wdx = ops.x3_split(wf), ops.x3_split(wd)
        Ho = x.shape[1] - 2
        flops = 2.0 * B * Ho * Ho * Co * 9 * Ci
        pooled, am = ops.conv_fwd(x, wf, b, 1)
        dp = torch.randn_like(pooled)
        dw, db = torch.empty_like(w), torch.empty_like(b)
        dx = torch.empty_like(x)
        xp = ops.x3_pack(x)
        dpp = ops.x3_pack(dp)
        t_pack = timeit(lambda: ops.x3_pack(x), args.iters)
        print(f"conv{l} input x3_pack {t_pack:.3f} ms", flush=True)
        for name, fn in (("fwd", lambda x3: ops.conv_fwd(x, wfx if x3 else wf, b, 1, x3=x3)),
                         ("dgrad", lambda x3: ops.conv_dgrad(dp, am, wdx if x3 else wd, x.shape, 1, out=dx, x3=x3)),
                         ("wgrad", lambda x3: ops.conv_wgrad(x, dp, am, dw, db, 1, x3=x3))):
            t = {x3: timeit(lambda: fn(x3), args.iters) for x3 in (False, True)}
            tf = {k: flops / v / 1e9 for k, v in t.items()}
            fp = {"fwd": lambda: ops.conv_fwd(xp, wfx, b, 1, x3=True),
                  "wgrad": lambda: ops.conv_wgrad(xp, dp, am, dw, db, 1, x3=True, dpooled_packed=dpp),
                  "dgrad": lambda: ops.conv_dgrad(dpp, am, wdx, x.shape, 1, out=dx, x3=True)}[name]
            tp = timeit(fp, args.iters)     # operands split once per tensor (x3-packed): what the engine runs
            print(f"conv{l}_{name:6s} packed input: {tp:7.3f} ms {flops / tp / 1e9:6.1f} TF   x{t[False] / tp:.2f}", flush=True)
            print(f"conv{l}_{name:6s} fp32-MFMA {t[False]:7.3f} ms {tf[False]:6.1f} TF ({100 * tf[False] / 157.3:4.1f}%)   "
                  f"3xbf16 {t[True]:7.3f} ms {tf[True]:6.1f} TF ({100 * tf[True] / 416.7:4.1f}% of 416.7)   x{t[False] / t[True]:.2f}",
                  flush=True)
        x = pooled


if __name__ == "__main__" and "--gemm" not in sys.argv:
    main()


def gemms():
    """the three v_conv products at the headline shape: fp32 MFMA vs split kernel"""
    dev = "cuda:0"
    Bn, Pn, C, mid = 256, 676, 256, 1024
    M = Bn * Pn
    v = torch.randn(M, C, device=dev)
    w = torch.randn(mid, C, device=dev) * 0.05
    qp = torch.randn(Bn, mid, device=dev)
    xs = torch.empty(M, mid, device=dev)
    dxp = torch.randn(M, mid, device=dev)
    dw = torch.empty(mid, C, device=dev)
    dv = torch.empty(M, C, device=dev)
    fl = 2.0 * M * C * mid
    for name, fn in (("v_conv fwd", lambda x3: ops.gemm(v, w, xs, M, mid, C, rowgroup=qp, rg_div=Pn, relu=True, x3=x3)),
                     ("v_conv dW", lambda x3: ops.gemm(dxp, v, dw, mid, C, M, transA=True, transB=False, lda=mid, ldb=C, x3=x3)),
                     ("v_conv dX", lambda x3: ops.gemm(dxp, w, dv, M, C, mid, transB=False, lda=mid, ldb=C, x3=x3))):
        t = {x3: timeit(lambda: fn(x3), 5) for x3 in (False, True)}
        print(f"{name:12s} fp32-MFMA {t[False]:7.3f} ms {fl / t[False] / 1e9:6.1f} TF   3xbf16 {t[True]:7.3f} ms {fl / t[True] / 1e9:6.1f} TF   x{t[False] / t[True]:.2f}", flush=True)


if __name__ == "__main__" and "--gemm" in sys.argv:
    gemms()
