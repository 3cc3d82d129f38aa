#!/usr/bin/env python3
"""Where do the MFMA waves of the conv kernels spend their cycles?  Needs the diagnostic library:
    python -m dl_vqa_amd.build --diag && VQA_LIB=dl_vqa_amd/libvqa_hip_diag.so python tools/diag_mfma.py"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dl_vqa_amd import _lib, ops

lib = _lib.load()
rd = lib.vqa_diag_read
rd.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
buf = (ctypes.c_ulonglong * 8)()
B, dev = 256, "cuda:0"
x = ops.nchw_to_nhwc4(torch.randn(B, 3, 224, 224, device=dev))
chans = [3, 64, 128, 256]
for l in range(3):
    Ci, Co = chans[l], chans[l + 1]
    w = torch.randn(Co, Ci, 3, 3, device=dev) * (9 * Ci) ** -0.5
    b = torch.zeros(Co, device=dev)
    wf, wd = ops.conv_pack_weights(w, x.shape[3])
    pooled, am = ops.conv_fwd(x, wf, b, 1)
    dp = torch.randn_like(pooled)
    dw, db = torch.empty_like(w), torch.empty_like(b)
    for name, fn in (("fwd", lambda: ops.conv_fwd(x, wf, b, 1)),
                     ("wgrad", lambda: ops.conv_wgrad(x, dp, am, dw, db, 1)),
                     ("dgrad", (lambda: ops.conv_dgrad(dp, am, wd, x.shape, 1)) if l else None)):
        if fn is None:
            continue
        rd(buf)
        fn()
        rd(buf)
        mma, bar, waves, tot, lst, lis, lba, lw = [int(v) for v in buf]
        if lw:
            lt = lst + lis + lba
            print(f"   loader waves: finish+ds_write (incl. load wait) {100*lst/lt:5.1f} %, issue {100*lis/lt:5.1f} %, "
                  f"barrier wait {100*lba/lt:5.1f} %  ({lt/lw/1e3:.1f} kcyc in the K loop)")
        if waves:
            print(f"conv{l}_{name}: per MFMA wave: {tot/waves/1e3:8.1f} kcyc total, MFMA block {100*mma/tot:5.1f} %, "
                  f"barrier wait {100*bar/tot:5.1f} %, other {100*(tot-mma-bar)/tot:5.1f} %  ({waves} waves)")
    x = pooled
