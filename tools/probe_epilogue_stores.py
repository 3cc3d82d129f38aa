"""Probe: kernels whose epilogues moved to 16-byte stores through LDS (conv forward, conv2 backward-data, the v_conv GEMMs)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dl_vqa_amd import ops, _lib

def timeit(fn, iters=8):
    fn(); fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters

dev = "cuda:0"
B = 256
rows = []
def run(name, fl, fn):
    ms = timeit(fn)
    print(f"{name:28s} {ms:.3f} ms {fl / ms / 1e9 / 157.3 * 100:5.1f}%", flush=True)

x1 = torch.randn(B, 111, 111, 64, device=dev); w1 = torch.randn(128, 64, 3, 3, device=dev) / 24; b1 = torch.zeros(128, device=dev)
wf1, wd1 = ops.conv_pack_weights(w1, 64)
run("conv1_fwd", 2.0 * B * 109 * 109 * 128 * 576, lambda: ops.conv_fwd(x1, wf1, b1, 1, tag=1))
x2 = torch.randn(B, 54, 54, 128, device=dev); w2 = torch.randn(256, 128, 3, 3, device=dev) / 34; b2 = torch.zeros(256, device=dev)
wf2, wd2 = ops.conv_pack_weights(w2, 128)
p2, a2 = ops.conv_fwd(x2, wf2, b2, 1, tag=2)
run("conv2_fwd", 2.0 * B * 52 * 52 * 256 * 1152, lambda: ops.conv_fwd(x2, wf2, b2, 1, tag=2))
dp2 = torch.randn_like(p2); dx2 = torch.empty_like(x2)
run("conv2_dgrad", 2.0 * B * 52 * 52 * 256 * 1152, lambda: ops.conv_dgrad(dp2, a2, wd2, x2.shape, 1, tag=2, out=dx2))
P, C, mid = 676, 256, 1024
M = B * P
vn = torch.randn(M, C, device=dev); wv = torch.randn(mid, C, device=dev); qp = torch.randn(B, mid, device=dev)
xs = torch.empty(M, mid, device=dev); dvn = torch.empty(M, C, device=dev)
for pers in ("1", "0"):
    os.environ["VQA_PERSISTENT"] = pers
    _lib.load().vqa_reload_knobs()
    run(f"v_conv_fwd persistent={pers}", 2.0 * M * C * mid, lambda: ops.gemm(vn, wv, xs, M, mid, C, rowgroup=qp, rg_div=P, relu=True))
    T, E, Hh = 14, 300, 1024
    xe = torch.randn(T * B, E, device=dev); wih = torch.randn(4 * Hh, E, device=dev); xg = torch.empty(T * B, 4 * Hh, device=dev)
    bi = torch.randn(4 * Hh, device=dev)
    run(f"lstm_xg persistent={pers}", 2.0 * T * B * E * 4 * Hh, lambda: ops.gemm(xe, wih, xg, T * B, 4 * Hh, E, bias1=bi, bias2=bi))
os.environ.pop("VQA_PERSISTENT"); _lib.load().vqa_reload_knobs()
run("v_conv_dgrad", 2.0 * M * C * mid, lambda: ops.gemm(xs, wv, dvn, M, C, mid, transB=False, lda=mid, ldb=C))
p1, a1 = ops.conv_fwd(x1, wf1, b1, 1, tag=1)
dp1 = torch.randn_like(p1); dx1 = torch.empty_like(x1)
wimg = ops.pconvf_pack_weights(w1)
run("conv1_dgrad (pconvf)", 2.0 * B * 109 * 109 * 128 * 576, lambda: ops.pconvf_dgrad(dp1, a1, wimg, tuple(x1.shape), tag=1, out=dx1))
run("conv1_dgrad (implicit)", 2.0 * B * 109 * 109 * 128 * 576, lambda: ops.conv_dgrad(dp1, a1, wd1, x1.shape, 1, tag=1, out=dx1))
