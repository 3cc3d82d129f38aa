#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "pconvf" > $O/r03_tests_y.log 2>&1; rc=$?; tail -12 $O/r03_tests_y.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 200 python - <<'PY' 2>&1 | grep -v amdgpu.ids
import torch, sys, os
sys.path.insert(0, os.getcwd())
from dl_vqa_amd import ops
def timeit(fn, iters=8):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
B = 256
for (H, Ci, Co) in ((111, 64, 128), (54, 128, 256)):
    x = torch.randn(B, H, H, Ci, device="cuda")
    w = torch.randn(Co, Ci, 3, 3, device="cuda") * (9 * Ci) ** -0.5
    b = torch.zeros(Co, device="cuda")
    wf, wd = ops.conv_pack_weights(w, Ci)
    pooled, am = ops.conv_fwd(x, wf, b, 1)
    dp = torch.randn_like(pooled)
    wdp = ops.pconvf_pack_weights(w)
    flops = 2.0 * B * (H - 2) ** 2 * Co * 9 * Ci
    for name, fn in (("implicit-GEMM dgrad", lambda: ops.conv_dgrad(dp, am, wd, tuple(x.shape), 1)),
                     ("patch dgrad", lambda: ops.pconvf_dgrad(dp, am, wdp, tuple(x.shape)))):
        ms = timeit(fn)
        print(f"{H}x{H} {Ci}->{Co} {name:22s} {ms:7.3f} ms  {flops / ms / 1e9:6.1f} TF/s  {100 * flops / ms / 1e9 / 157.3:5.1f} % of fp32 MFMA peak", flush=True)
PY
