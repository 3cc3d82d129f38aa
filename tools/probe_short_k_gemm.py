"""Probe: the v_conv forward GEMM (M = B*676, N = 1024, K = 256) with and without its fused epilogue, persistent or not."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dl_vqa_amd import ops, _lib

def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters

dev = "cuda:0"
B, P, C, mid = 256, 676, 256, 1024
M = B * P
vn = torch.randn(M, C, device=dev); wv = torch.randn(mid, C, device=dev); qp = torch.randn(B, mid, device=dev)
xs = torch.empty(M, mid, device=dev)
fl = 2.0 * M * C * mid
for pers in ("1", "0"):
    os.environ["VQA_PERSISTENT"] = pers
    _lib.load().vqa_reload_knobs()
    for name, kw in (("plain", {}), ("rowgroup+relu", dict(rowgroup=qp, rg_div=P, relu=True)), ("relu", dict(relu=True))):
        ms = timeit(lambda: ops.gemm(vn, wv, xs, M, mid, C, **kw))
        print(f"persistent={pers} {name:14s} {ms:.3f} ms  {fl/ms/1e9:.1f} TF/s  {fl/ms/1e9/157.3*100:.1f}%", flush=True)
# K sweep at the same M, N: where does the efficiency come back?
os.environ["VQA_PERSISTENT"] = "1"; _lib.load().vqa_reload_knobs()
for K in (128, 256, 512, 1024):
    a = torch.randn(M, K, device=dev); w = torch.randn(mid, K, device=dev)
    ms = timeit(lambda: ops.gemm(a, w, xs, M, mid, K))
    f = 2.0 * M * K * mid
    print(f"K={K:5d} plain {ms:.3f} ms {f/ms/1e9/157.3*100:.1f}%", flush=True)
