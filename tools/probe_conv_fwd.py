"""Probe: isolated timings of the conv forward kernels and the v_conv forward GEMM (B = 256; PB overrides), repeated PN times.
Used with VQA_LIB=build_var/libvqa_<variant>.so for the timing-only builds (-DVQA_EXP_NOSTORE_AM / _ALL, -DVQA_EXP_NO_STAGED_POOL;
DESIGN 4.1).  The start-phase stagger experiment (a VQA_STAGGER knob delaying blocks 256..511) also ran through this script; the
knob was withdrawn with the experiment."""
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
B = int(os.environ.get("PB", "256"))
cases = []
x1 = torch.randn(B, 111, 111, 64, device=dev); w1 = torch.randn(128, 64, 3, 3, device=dev) / 24; b1 = torch.zeros(128, device=dev)
wf1, _ = ops.conv_pack_weights(w1, 64)
cases.append(("conv1_fwd", 2.0 * B * 109 * 109 * 128 * 576, lambda: ops.conv_fwd(x1, wf1, b1, 1, tag=1)))
x2 = torch.randn(B, 54, 54, 128, device=dev); w2 = torch.randn(256, 128, 3, 3, device=dev) / 34; b2 = torch.zeros(256, device=dev)
wf2, _ = ops.conv_pack_weights(w2, 128)
cases.append(("conv2_fwd", 2.0 * B * 52 * 52 * 256 * 1152, lambda: ops.conv_fwd(x2, wf2, b2, 1, tag=2)))
P, C, mid = 676, 256, 1024
M = B * P
vn = torch.randn(M, C, device=dev); wv = torch.randn(mid, C, device=dev); qp = torch.randn(B, mid, device=dev)
xs = torch.empty(M, mid, device=dev)
cases.append(("v_conv_fwd", 2.0 * M * C * mid, lambda: ops.gemm(vn, wv, xs, M, mid, C, rowgroup=qp, rg_div=P, relu=True)))
for rep in range(int(os.environ.get("PN", "3"))):
    row = [f"run {rep}"]
    for name, fl, fn in cases:
        ms = timeit(fn)
        row.append(f"{name} {ms:.3f} ms {fl / ms / 1e9 / 157.3 * 100:.1f}%")
    print("  ".join(row), flush=True)
