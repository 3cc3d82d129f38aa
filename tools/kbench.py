#!/usr/bin/env python3
"""Per-kernel micro-benchmark on one MI355X: times each contraction of the train step in isolation
(HIP events on the launch stream) and prints achieved fp32 TFLOP/s against the 157.3 TF MFMA peak.

    python tools/kbench.py [--batch 256] [--only conv1_wgrad,...] [--iters 5]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dl_vqa_amd import ops  # noqa: E402

PEAK = 157.3


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
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    B = args.batch
    dev = "cuda:0"
    only = set(filter(None, args.only.split(",")))
    res = []

    def run(name, flops, fn):
        if only and name not in only:
            return
        ms = timeit(fn, args.iters)
        tf = flops / ms / 1e9
        res.append((name, ms, tf))
        print(f"{name:22s} {ms:9.3f} ms  {tf:7.1f} TF/s  {100 * tf / PEAK:5.1f}% of fp32 MFMA peak", flush=True)

    # conv layers of the reference architecture at 224x224
    H = 224
    chans = [3, 64, 128, 256]
    img = torch.randn(B, 3, H, H, device=dev)
    w0 = torch.randn(64, 3, 3, 3, device=dev) * 0.2
    b0 = torch.zeros(64, device=dev)
    f0 = 2.0 * B * 222 * 222 * 64 * 27
    p0, a0 = ops.conv0_fwd(img, w0, b0)
    run("conv0_fast_fwd", f0, lambda: ops.conv0_fwd(img, w0, b0))
    dp0, dw0, db0 = torch.randn_like(p0), torch.empty_like(w0), torch.empty_like(b0)
    run("conv0_fast_wgrad", f0, lambda: ops.conv0_wgrad(img, dp0, a0, dw0, db0))
    x = ops.nchw_to_nhwc4(img)
    for l in range(3):
        Ci, Co = chans[l], chans[l + 1]
        CiP = x.shape[3]
        w = torch.randn(Co, Ci, 3, 3, device=dev) * (1.0 / (9 * Ci) ** 0.5)
        b = torch.zeros(Co, device=dev)
        wf, wd = ops.conv_pack_weights(w, CiP)
        Hin = x.shape[1]
        Ho = Hin - 2
        flops = 2.0 * B * Ho * Ho * Co * 9 * Ci
        pooled, am = ops.conv_fwd(x, wf, b, 1, tag=l)
        run(f"conv{l}_fwd", flops, lambda: ops.conv_fwd(x, wf, b, 1, tag=l))
        dp = torch.randn_like(pooled)
        dw, db = torch.empty_like(w), torch.empty_like(b)
        run(f"conv{l}_wgrad", flops, lambda: ops.conv_wgrad(x, dp, am, dw, db, 1, tag=l))
        if l > 0:
            dx = torch.empty_like(x)
            run(f"conv{l}_dgrad", flops, lambda: ops.conv_dgrad(dp, am, wd, x.shape, 1, tag=l, out=dx))
        x = pooled
    # attention / LSTM / classifier GEMM shapes
    P, C, mid, Hh, E, T = 26 * 26, 256, 1024, 1024, 300, 14
    M = B * P
    vn = torch.randn(M, C, device=dev)
    wv = torch.randn(mid, C, device=dev)
    qp = torch.randn(B, mid, device=dev)
    xs = torch.empty(M, mid, device=dev)
    run("v_conv_fwd", 2.0 * M * C * mid, lambda: ops.gemm(vn, wv, xs, M, mid, C, rowgroup=qp, rg_div=P, relu=True))
    dwv = torch.empty(mid, C, device=dev)
    run("v_conv_wgrad", 2.0 * M * C * mid, lambda: ops.gemm(xs, vn, dwv, mid, C, M, transA=True, transB=False, lda=mid, ldb=C))
    dvn = torch.empty(M, C, device=dev)
    run("v_conv_dgrad", 2.0 * M * C * mid, lambda: ops.gemm(xs, wv, dvn, M, C, mid, transB=False, lda=mid, ldb=C))
    h = torch.randn(B, Hh, device=dev)
    whh = torch.randn(4 * Hh, Hh, device=dev)
    hg = torch.empty(B, 4 * Hh, device=dev)
    run("lstm_step_fwd", 2.0 * B * Hh * 4 * Hh, lambda: ops.gemm(h, whh, hg, B, 4 * Hh, Hh))
    dg = torch.randn(B, 4 * Hh, device=dev)
    dh = torch.zeros(B, Hh, device=dev)
    run("lstm_step_bwd", 2.0 * B * Hh * 4 * Hh, lambda: ops.gemm(dg, whh, dh, B, Hh, 4 * Hh, transB=False, lda=4 * Hh, ldb=Hh, accumulate=True))
    dga = torch.randn(T * B, 4 * Hh, device=dev)
    ha = torch.randn(T * B, Hh, device=dev)
    dwhh = torch.empty(4 * Hh, Hh, device=dev)
    run("lstm_whh_grad", 2.0 * T * B * Hh * 4 * Hh, lambda: ops.gemm(dga, ha, dwhh, 4 * Hh, Hh, T * B, transA=True, transB=False, lda=4 * Hh, ldb=Hh))
    xe = torch.randn(T * B, E, device=dev)
    wih = torch.randn(4 * Hh, E, device=dev)
    xg = torch.empty(T * B, 4 * Hh, device=dev)
    run("lstm_xg", 2.0 * T * B * E * 4 * Hh, lambda: ops.gemm(xe, wih, xg, T * B, 4 * Hh, E))
    comb = torch.randn(B, 2560, device=dev)
    w1 = torch.randn(1024, 2560, device=dev)
    h1 = torch.empty(B, 1024, device=dev)
    run("lin1_fwd", 2.0 * B * 2560 * 1024, lambda: ops.gemm(comb, w1, h1, B, 1024, 2560))
    # square GEMMs: the engine's ceiling with the plain loaders (no conv addressing)
    S = 4096
    a = torch.randn(S, S, device=dev)
    bm = torch.randn(S, S, device=dev)
    cm = torch.empty(S, S, device=dev)
    run("sq4096_nt", 2.0 * S ** 3, lambda: ops.gemm(a, bm, cm, S, S, S))
    run("sq4096_nn", 2.0 * S ** 3, lambda: ops.gemm(a, bm, cm, S, S, S, transB=False, lda=S, ldb=S))
    run("sq4096_tn", 2.0 * S ** 3, lambda: ops.gemm(a, bm, cm, S, S, S, transA=True, transB=False, lda=S, ldb=S))
    tot = sum(r[1] for r in res)
    print(f"sum of listed kernels: {tot:.3f} ms")


if __name__ == "__main__":
    main()
