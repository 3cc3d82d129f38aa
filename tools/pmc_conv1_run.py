#!/usr/bin/env python3
"""Workload for the PMC passes: 3 launches each of the conv1 and conv2 forward / wgrad / dgrad kernels at the
bench shapes (B = 256), always in the order layer 1, layer 2.
    rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_fetch -- python3 tools/pmc_conv1_run.py
    rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_write -- python3 tools/pmc_conv1_run.py
    python tools/pmc_traffic_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write > profiles/rNN_conv1_traffic.json"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dl_vqa_amd import ops

B, dev = 256, "cuda:0"
layers = []
for (Hin, Ci, Co) in ((111, 64, 128), (54, 128, 256)):
    x = torch.randn(B, Hin, Hin, Ci, device=dev)
    w = torch.randn(Co, Ci, 3, 3, device=dev) * (9 * Ci) ** -0.5
    layers.append(dict(x=x, w=w, b=torch.zeros(Co, device=dev)))
for l, L in enumerate(layers):          # set-up launches, same layer order as the measured ones
    L["wf"], L["wd"] = ops.conv_pack_weights(L["w"], L["x"].shape[3])
    L["pooled"], L["am"] = ops.conv_fwd(L["x"], L["wf"], L["b"], 1, tag=l + 1)
    L["dp"] = torch.randn_like(L["pooled"])
    L["dw"], L["db"], L["dx"] = torch.empty_like(L["w"]), torch.empty_like(L["b"]), torch.empty_like(L["x"])
    ops.conv_wgrad(L["x"], L["dp"], L["am"], L["dw"], L["db"], 1, tag=l + 1)
    ops.conv_dgrad(L["dp"], L["am"], L["wd"], L["x"].shape, 1, tag=l + 1, out=L["dx"])
for _ in range(3):
    for l, L in enumerate(layers):
        ops.conv_fwd(L["x"], L["wf"], L["b"], 1, tag=l + 1)
        ops.conv_wgrad(L["x"], L["dp"], L["am"], L["dw"], L["db"], 1, tag=l + 1)
        ops.conv_dgrad(L["dp"], L["am"], L["wd"], L["x"].shape, 1, tag=l + 1, out=L["dx"])
torch.cuda.synchronize()
