#!/usr/bin/env python3
"""Workload for the PMC passes: 3 launches each of the conv1 forward / wgrad / dgrad kernels at the bench shape.
    rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_fetch -- python3 tools/pmc_conv1_run.py
    rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_write -- python3 tools/pmc_conv1_run.py
    python tools/pmc_traffic_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write > profiles/rNN_conv1_traffic.json"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dl_vqa_amd import ops

B, dev = 256, "cuda:0"
x = torch.randn(B, 111, 111, 64, device=dev)
w = torch.randn(128, 64, 3, 3, device=dev) * (9 * 64) ** -0.5
b = torch.zeros(128, device=dev)
wf, wd = ops.conv_pack_weights(w, 64)
pooled, am = ops.conv_fwd(x, wf, b, 1, tag=1)
dp = torch.randn_like(pooled)
dw, db = torch.empty_like(w), torch.empty_like(b)
dx = torch.empty_like(x)
for _ in range(3):
    ops.conv_fwd(x, wf, b, 1, tag=1)
    ops.conv_wgrad(x, dp, am, dw, db, 1, tag=1)
    ops.conv_dgrad(dp, am, wd, x.shape, 1, tag=1, out=dx)
torch.cuda.synchronize()
