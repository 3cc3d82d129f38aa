#!/usr/bin/env python3
"""Workload for the per-layer PMC passes: 3 launches each of the conv1 and conv2 forward / wgrad / dgrad kernels at a
bench shape, always in the order layer 1, layer 2 (the summary splits a family's dispatches by that parity).
    rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_fetch -- python3 tools/pmc_conv_run.py [--dtype bf16 --batch 512 --size 448]
    rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_write -- python3 tools/pmc_conv_run.py [same arguments]
    python tools/pmc_traffic_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write > profiles/r03_conv_traffic_<dtype>_<size>_<batch>.json
bench.py reads that file for `roofline.traffic` / `kernels[].traffic`."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dl_vqa_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="fp32", choices=["fp32", "bf16"])
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--size", type=int, default=224)
args = ap.parse_args()
B, dev = args.batch, "cuda:0"
bf = args.dtype == "bf16"
h1 = (args.size - 2) // 2
h2 = (h1 - 2) // 2
layers = []
for (Hin, Ci, Co) in ((h1, 64, 128), (h2, 128, 256)):
    x = torch.randn(B, Hin, Hin, Ci, device=dev)
    w = torch.randn(Co, Ci, 3, 3, device=dev) * (9 * Ci) ** -0.5
    layers.append(dict(x=x.to(torch.bfloat16) if bf else x, w=w, b=torch.zeros(Co, device=dev), last=Ci == 128))


def fwd(L, l):
    if bf:      # the engine's bf16 path: patch kernels on channel-blocked (C16) activations
        return ops.pconv_fwd(L["xc"], L["wf"], L["b"], L["w"].shape[0], out_dtype=torch.float32 if L["last"] else torch.bfloat16, tag=l + 1)
    return ops.conv_fwd(L["x"], L["wf"], L["b"], 1, tag=l + 1)


def wgrad(L, l):
    if bf:
        return ops.pconv_wgrad(L["xc"], L["dp"], L["am"], L["dw"], L["db"], tag=l + 1)
    return ops.conv_wgrad(L["x"], L["dp"], L["am"], L["dw"], L["db"], 1, tag=l + 1)


def dgrad(L, l):
    if bf:
        return ops.pconv_dgrad(L["dp"], L["am"], L["wd"], L["x"].shape, out_c16=l > 0, tag=l + 1)
    return ops.conv_dgrad(L["dp"], L["am"], L["wd"], L["x"].shape, 1, tag=l + 1, out=L["dx"])


for l, L in enumerate(layers):          # set-up launches, same layer order as the measured ones
    if bf:
        L["wf"], L["wd"] = ops.pconv_pack_weights(L["w"])
        L["xc"] = ops.to_c16(L["x"])
    else:
        L["wf"], L["wd"] = ops.conv_pack_weights(L["w"], L["x"].shape[3])
    L["pooled"], L["am"] = fwd(L, l)
    L["dp"] = torch.randn(L["am"].shape, device=dev).to(torch.bfloat16 if bf else torch.float32)
    L["dw"], L["db"] = torch.empty_like(L["w"]), torch.empty_like(L["b"])
    L["dx"] = None if bf else torch.empty_like(L["x"])
    dgrad(L, l)
    wgrad(L, l)
for _ in range(3):
    for l, L in enumerate(layers):
        fwd(L, l)
        dgrad(L, l)
        wgrad(L, l)
torch.cuda.synchronize()
print(f"pmc_conv_run: dtype={args.dtype} batch={B} size={args.size} done")
