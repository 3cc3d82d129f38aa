"""Tensor-level wrappers over the C ABI (one Python function per entry point of include/vqa_hip.h).

torch supplies device memory and the current HIP stream; every computation happens in the HIP
library.  Nothing here falls back to torch arithmetic.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import call, ptr, stream

_ws = {}


def workspace(nbytes: int, device) -> torch.Tensor:
    """A scratch buffer that only grows (split-K slabs, reduction partials), one per (device, stream):
    launch sequences that run concurrently on different HIP streams must not share slabs."""
    key = (str(device), torch.cuda.current_stream(device).cuda_stream)
    t = _ws.get(key)
    if t is None or t.numel() * 4 < nbytes:
        t = torch.empty(max(nbytes // 4 + 1024, 1 << 20), dtype=torch.float32, device=device)
        _ws[key] = t
    return t


def _chk(t: torch.Tensor, dtype=torch.float32):
    assert t.is_cuda and t.dtype == dtype, (t.device, t.dtype)


def gemm(A: torch.Tensor, B: torch.Tensor, C: torch.Tensor, M: int, N: int, K: int, *, transA=False,
         transB=True, lda=None, ldb=None, ldc=None, bias1=None, bias2=None, rowgroup=None, rg_div=1,
         rg_op=0, relu=False, accumulate=False, aux=None, tag=0, x3=False) -> torch.Tensor:
    """C[M,N] = act(op(A) op(B) (op) rowgroup + bias) (+C).  Leading dims default to the stored row length.
    x3: the contraction on the bf16 matrix cores with exact 3 x bf16 operand splits (csrc/x3_core.hpp)."""
    lib = _lib.load()
    lda = lda if lda is not None else (M if transA else K)
    ldb = ldb if ldb is not None else (K if transB else N)
    ldc = ldc if ldc is not None else N
    nbytes = (lib.vqa_gemm_x3_workspace_bytes if x3 else lib.vqa_gemm_workspace_bytes)(M, N, K)
    ws = workspace(nbytes, A.device) if nbytes else None
    call("vqa_gemm_x3" if x3 else "vqa_gemm", ptr(A), lda, int(transA), ptr(B), ldb, int(transB), ptr(C), ldc, M, N, K,
         ptr(bias1), ptr(bias2), ptr(rowgroup), (rowgroup.stride(0) if rowgroup is not None else 0),
         rg_div, rg_op, int(relu), int(accumulate), ptr(aux), ptr(ws), (ws.numel() * 4 if ws is not None else 0),
         tag, stream())
    return C


def nchw_to_nhwc4(x: torch.Tensor) -> torch.Tensor:
    B, Cc, H, W = x.shape
    y = torch.empty(B, H, W, 4, dtype=torch.float32, device=x.device)
    call("vqa_nchw_to_nhwc4", ptr(x), ptr(y), B, Cc, H, W, stream())
    return y


def conv_pack_weights(w: torch.Tensor, CiP: int, need_wd: bool = True):
    Co, Ci = w.shape[0], w.shape[1]
    wf = torch.empty(9 * CiP, Co, dtype=torch.float32, device=w.device)
    wd = torch.empty(9 * Co, CiP, dtype=torch.float32, device=w.device) if need_wd else None
    call("vqa_conv_pack_weights", ptr(w), ptr(wf), ptr(wd), Co, Ci, CiP, stream())
    return wf, wd


def conv_out_hw(H: int, W: int, stride: int) -> Tuple[int, int]:
    return ((H - 3) // stride + 1) // 2, ((W - 3) // stride + 1) // 2


def convk_out_hw(H: int, W: int, ks: int, stride: int) -> Tuple[int, int, int, int]:
    """(Ho, Wo, Hp, Wp) of Conv2d(kernel_size=ks, stride, pad=0) -> MaxPool2d(2,2) (models/model.py:80-82)."""
    Ho, Wo = (H - ks) // stride + 1, (W - ks) // stride + 1
    return Ho, Wo, Ho // 2, Wo // 2


def convk_pack_weights(w: torch.Tensor, CiP: int) -> torch.Tensor:
    """torch [Co][Ci][ks][ks] -> wk [Co][ks*ks*CiP], K index (ky*ks + kx)*CiP + ci (the im2col matrix's)."""
    Co, Ci, ks, ks2 = w.shape
    assert ks == ks2
    wk = torch.empty(Co, ks * ks * CiP, dtype=torch.float32, device=w.device)
    call("vqa_convk_pack_weights", ptr(w), ptr(wk), Co, Ci, CiP, ks, stream())
    return wk


def convk_unpack_wgrad(dwk: torch.Tensor, dw: torch.Tensor, CiP: int) -> torch.Tensor:
    Co, Ci, ks, _ = dw.shape
    call("vqa_convk_unpack_wgrad", ptr(dwk), ptr(dw), Co, Ci, CiP, ks, stream())
    return dw


def convk_chunk(B: int, H: int, W: int, CiP: int, Co: int, ks: int, stride: int, limit_bytes: int = 1 << 31) -> int:
    """Images per im2col chunk: the matrix [rows][ks*ks*CiP] (and the GEMM's [rows][Co] result) stay below limit_bytes."""
    Ho, Wo, _, _ = convk_out_hw(H, W, ks, stride)
    per_img = Ho * Wo * max(ks * ks * CiP, Co) * 4
    return max(1, min(B, limit_bytes // per_img))


def convk_fwd(x: torch.Tensor, wk: torch.Tensor, bias: torch.Tensor, ks: int, stride: int = 1, tag: int = 0, chunk: int = 0):
    """Conv block with kernel_size != 3, materialised form (csrc/conv_generic.hip): x NHWC [B,H,W,CiP] ->
    (pooled [B,Hp,Wp,Co], arg-max bytes).  im2col of a batch chunk, vqa_gemm with the bias in its epilogue, ReLU + pool."""
    B, H, W, CiP = x.shape
    Co, K = wk.shape
    assert K == ks * ks * CiP and x.dtype == torch.float32 and x.is_contiguous()
    Ho, Wo, Hp, Wp = convk_out_hw(H, W, ks, stride)
    assert Hp > 0 and Wp > 0, "image too small for conv + pool"
    pooled = torch.empty(B, Hp, Wp, Co, dtype=torch.float32, device=x.device)
    amax = torch.empty(B, Hp, Wp, Co, dtype=torch.uint8, device=x.device)
    nb = chunk or convk_chunk(B, H, W, CiP, Co, ks, stride)
    cols = torch.empty(nb * Ho * Wo, K, dtype=torch.float32, device=x.device)
    y = torch.empty(nb * Ho * Wo, Co, dtype=torch.float32, device=x.device)
    for b0 in range(0, B, nb):
        n = min(nb, B - b0)
        call("vqa_convk_im2col", ptr(x[b0:]), ptr(cols), n, H, W, CiP, ks, stride, stream())
        gemm(cols, wk, y, n * Ho * Wo, Co, K, transB=True, bias1=bias, tag=tag)
        call("vqa_convk_relu_pool", ptr(y), ptr(pooled[b0:]), ptr(amax[b0:]), n, Ho, Wo, Co, stream())
    return pooled, amax


def convk_bwd(x: torch.Tensor, dpooled: torch.Tensor, amax: torch.Tensor, wk: torch.Tensor, dw: torch.Tensor,
              dbias: torch.Tensor, ks: int, stride: int = 1, need_dx: bool = True, tag: int = 0, chunk: int = 0):
    """Backward of convk_fwd: dw [Co][Ci][ks][ks] and dbias are written, dX NHWC is returned (None if not need_dx)."""
    B, H, W, CiP = x.shape
    Co, K = wk.shape
    Ho, Wo, Hp, Wp = convk_out_hw(H, W, ks, stride)
    assert dpooled.shape == (B, Hp, Wp, Co) and dpooled.dtype == torch.float32 and dpooled.is_contiguous()
    nb = chunk or convk_chunk(B, H, W, CiP, Co, ks, stride)
    cols = torch.empty(nb * Ho * Wo, K, dtype=torch.float32, device=x.device)
    dy = torch.empty(nb * Ho * Wo, Co, dtype=torch.float32, device=x.device)
    dcols = torch.empty(nb * Ho * Wo, K, dtype=torch.float32, device=x.device) if need_dx else None
    dwk = torch.empty(Co, K, dtype=torch.float32, device=x.device)
    dx = torch.empty(B, H, W, CiP, dtype=torch.float32, device=x.device) if need_dx else None
    for b0 in range(0, B, nb):
        n = min(nb, B - b0)
        rows = n * Ho * Wo
        call("vqa_convk_route", ptr(dpooled[b0:]), ptr(amax[b0:]), ptr(dy), n, Ho, Wo, Co, stream())
        call("vqa_convk_im2col", ptr(x[b0:]), ptr(cols), n, H, W, CiP, ks, stride, stream())
        # dWk [Co][K] (+)= dY^T . cols over the chunk's rows
        gemm(dy, cols, dwk, Co, K, rows, transA=True, transB=False, accumulate=b0 > 0, tag=tag)
        if need_dx:
            gemm(dy, wk, dcols, rows, K, Co, transA=False, transB=False, tag=tag)
            call("vqa_convk_col2im", ptr(dcols), ptr(dx[b0:]), n, H, W, CiP, ks, stride, stream())
    convk_unpack_wgrad(dwk, dw, CiP)
    # windows whose ReLU was dead (arg-max byte 4) pass no gradient to the bias either
    colsum(dpooled, B * Hp * Wp, Co, dbias, mask=amax)
    return dx


def x3_split(x: torch.Tensor) -> torch.Tensor:
    """The exact three-way bf16 split of the fp32x3 kernels: x (fp32, numel % 4 == 0) -> planes [3, *x.shape] bf16
    (hi, mid, lo) with x == hi + mid + lo."""
    out = torch.empty((3,) + tuple(x.shape), dtype=torch.bfloat16, device=x.device)
    call("vqa_x3_split", ptr(x), ptr(out[0]), ptr(out[1]), ptr(out[2]), x.numel(), stream())
    return out


def x3_pack(x: torch.Tensor) -> torch.Tensor:
    """fp32 activation [..., C] (C % 4 == 0) -> the x3-packed form [..., C/4, 3, 4] bf16: every four channels as
    hi[4] mid[4] lo[4] (include/vqa_hip.h vqa_x3_pack); accepted as the input of conv_fwd / conv_wgrad with x3=True."""
    assert x.dtype == torch.float32 and x.shape[-1] % 4 == 0 and x.is_contiguous()
    out = torch.empty(tuple(x.shape[:-1]) + (x.shape[-1] // 4, 3, 4), dtype=torch.bfloat16, device=x.device)
    call("vqa_x3_pack", ptr(x), ptr(out), x.numel(), stream())
    return out


def nhwc_shape(x) -> tuple:
    """(B, H, W, C) of an NHWC activation, fp32 / bf16 [B,H,W,C] or x3-packed [B,H,W,C/4,3,4]."""
    if x.dim() == 6:
        return (x.shape[0], x.shape[1], x.shape[2], x.shape[3] * 4)
    return tuple(x.shape)


def x3_pack_pooled_grad(dpooled: torch.Tensor, amax: torch.Tensor, dbias: torch.Tensor) -> torch.Tensor:
    """x3_pack(dpooled) and, from the same read, dbias = sum of dpooled over the windows whose ReLU was alive."""
    Co = dpooled.shape[-1]
    out = torch.empty(tuple(dpooled.shape[:-1]) + (Co // 4, 3, 4), dtype=torch.bfloat16, device=dpooled.device)
    ws = workspace(_lib.load().vqa_x3_pack_pooled_grad_workspace_bytes(Co), dpooled.device)
    call("vqa_x3_pack_pooled_grad", ptr(dpooled), ptr(amax), ptr(out), ptr(dbias), dpooled.numel() // Co, Co, ptr(ws),
         ws.numel() * 4, stream())
    return out


def _x3_input(x):
    """(B, H, W, CiP, packed) of a conv input that is fp32 NHWC or x3-packed."""
    if x.dtype == torch.bfloat16 and x.dim() == 6:
        return x.shape[0], x.shape[1], x.shape[2], x.shape[3] * 4, 1
    assert x.dtype == torch.float32 and x.dim() == 4
    return x.shape[0], x.shape[1], x.shape[2], x.shape[3], 0


def conv_x3_supported(H: int, W: int, CiP: int, Co: int, stride: int) -> bool:
    return bool(_lib.load().vqa_conv3x3_x3_supported(H, W, CiP, Co, stride))


def conv_fwd(x: torch.Tensor, wf: torch.Tensor, bias: torch.Tensor, stride: int = 1, tag: int = 0, x3: bool = False,
             out_packed: bool = False):
    """x NHWC [B,H,W,CiP] -> (pooled [B,Hp,Wp,Co], argmax uint8 same shape).  x3: fp32 on the bf16 matrix cores
    (exact 3 x bf16 operand split, csrc/x3_core.hpp) instead of the fp32 MFMA; wf is then x3_split(packed weights)."""
    B, H, W, CiP, packed = _x3_input(x) if x3 else (*x.shape, 0)
    Co = wf.shape[-1]
    assert (wf.dtype == torch.bfloat16 and wf.dim() == 3) if x3 else wf.dtype == torch.float32
    Hp, Wp = conv_out_hw(H, W, stride)
    assert x3 or not out_packed
    pooled = (torch.empty(B, Hp, Wp, Co // 4, 3, 4, dtype=torch.bfloat16, device=x.device) if out_packed else
              torch.empty(B, Hp, Wp, Co, dtype=torch.float32, device=x.device))
    amax = torch.empty(B, Hp, Wp, Co, dtype=torch.uint8, device=x.device)
    if x3:
        call("vqa_conv3x3_relu_pool_fwd_x3", ptr(x), packed, ptr(wf), ptr(bias), ptr(pooled), int(out_packed), ptr(amax),
             B, H, W, CiP, Co, stride, tag, stream())
    else:
        call("vqa_conv3x3_relu_pool_fwd", ptr(x), ptr(wf), ptr(bias), ptr(pooled), ptr(amax), B, H, W, CiP, Co, stride,
             tag, stream())
    return pooled, amax


def conv_dgrad(dpooled, amax, wd, x_shape, stride: int = 1, tag: int = 0, out=None, x3: bool = False) -> torch.Tensor:
    """dpooled: fp32 [B,Hp,Wp,Co], or (x3 only) its x3-packed form (x3_pack)."""
    B, H, W, CiP = x_shape
    Co = nhwc_shape(dpooled)[3]
    assert (wd.dtype == torch.bfloat16 and wd.dim() == 3) if x3 else wd.dtype == torch.float32
    dx = out if out is not None else torch.empty(B, H, W, CiP, dtype=torch.float32, device=dpooled.device)
    if x3:
        call("vqa_conv3x3_dgrad_x3", ptr(dpooled), int(dpooled.dim() == 6), ptr(amax), ptr(wd), ptr(dx), B, H, W, CiP, Co,
             stride, tag, stream())
    else:
        call("vqa_conv3x3_dgrad", ptr(dpooled), ptr(amax), ptr(wd), ptr(dx), B, H, W, CiP, Co, stride, tag, stream())
    return dx


def conv_wgrad(x, dpooled, amax, dw: torch.Tensor, dbias: torch.Tensor, stride: int = 1, tag: int = 0, x3: bool = False,
               dpooled_packed=None):
    """x3: x may be x3-packed; dpooled_packed (optional) = x3_pack(dpooled), read by the contraction instead of dpooled;
    dbias None (x3 only): the bias gradient is not computed here (x3_pack_pooled_grad already produced it)."""
    lib = _lib.load()
    B, H, W, CiP, packed = _x3_input(x) if x3 else (*x.shape, 0)
    Co, Ci = dw.shape[0], dw.shape[1]
    nbytes = (lib.vqa_conv3x3_wgrad_x3_workspace_bytes if x3 else lib.vqa_conv3x3_wgrad_workspace_bytes)(B, H, W, CiP, Co, stride)
    ws = workspace(nbytes, x.device)
    if x3:
        call("vqa_conv3x3_wgrad_x3", ptr(x), packed, ptr(dpooled), ptr(dpooled_packed) if dpooled_packed is not None else None,
             ptr(amax), ptr(dw), ptr(dbias), B, H, W, CiP, Ci, Co, stride, ptr(ws), ws.numel() * 4, tag, stream())
        return
    call("vqa_conv3x3_wgrad", ptr(x), ptr(dpooled), ptr(amax), ptr(dw), ptr(dbias), B, H, W, CiP, Ci, Co, stride,
         ptr(ws), ws.numel() * 4, tag, stream())


def conv0_supported(Ci: int, H: int, W: int, Co: int, stride: int) -> bool:
    return bool(_lib.load().vqa_conv0_supported(Ci, H, W, Co, stride))


def conv0_fwd(x_nchw: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, out_dtype=torch.float32, bf16_mfma=False,
              out_packed=False, out_c16=False):
    """First conv block straight from the NCHW image: (pooled NHWC [B,Hp,Wp,Co] fp32 or bf16, argmax uint8).
    bf16_mfma (bf16 output only): image and weights rounded to bf16, bf16 MFMA.  out_packed: fp32 MFMA, the output
    written in the x3-packed form [B,Hp,Wp,Co/4,3,4] bf16 (see x3_pack).  out_c16 (with bf16_mfma): the pooled map
    channel-blocked [B,Co/16,Hp,Wp,16] for the patch convolutions; the arg-max bytes stay NHWC."""
    B, Ci, H, W = x_nchw.shape
    Co = w.shape[0]
    Hp, Wp = conv_out_hw(H, W, 1)
    assert not out_c16 or (bf16_mfma and out_dtype == torch.bfloat16 and not out_packed)
    pooled = (torch.empty(B, Hp, Wp, Co // 4, 3, 4, dtype=torch.bfloat16, device=x_nchw.device) if out_packed else
              torch.empty(B, Co // 16, Hp, Wp, 16, dtype=torch.bfloat16, device=x_nchw.device) if out_c16 else
              torch.empty(B, Hp, Wp, Co, dtype=out_dtype, device=x_nchw.device))
    amax = torch.empty(B, Hp, Wp, Co, dtype=torch.uint8, device=x_nchw.device)
    mode = 3 if out_packed else (4 if out_c16 else 2 if bf16_mfma else 1) if out_dtype == torch.bfloat16 else 0
    assert x_nchw.dtype in (torch.float32, torch.float16) and x_nchw.is_contiguous()
    call("vqa_conv0_relu_pool_fwd", ptr(x_nchw), int(x_nchw.dtype == torch.float16), ptr(w), ptr(bias), ptr(pooled), mode,
         ptr(amax), B, Ci, H, W, Co, stream())
    return pooled, amax


def conv0_wgrad(x_nchw, dpooled, amax, dw: torch.Tensor, dbias: torch.Tensor):
    lib = _lib.load()
    B, Ci, H, W = x_nchw.shape
    Co = dw.shape[0]
    ws = workspace(lib.vqa_conv0_wgrad_workspace_bytes(Co), x_nchw.device)
    call("vqa_conv0_wgrad", ptr(x_nchw), int(x_nchw.dtype == torch.float16), ptr(dpooled), ptr(amax), ptr(dw), ptr(dbias), B, Ci, H, W, Co, ptr(ws),
         ws.numel() * 4, stream())


def conv0_wgrad_bf16(x_nchw, dpooled16, amax, dw: torch.Tensor, dbias: torch.Tensor):
    """First block's weight / bias gradient on bf16 MFMA from the bf16 pooled gradient."""
    assert dpooled16.dtype == torch.bfloat16
    lib = _lib.load()
    B, Ci, H, W = x_nchw.shape
    Co = dw.shape[0]
    ws = workspace(lib.vqa_conv0_wgrad_workspace_bytes(Co), x_nchw.device)
    call("vqa_conv0_wgrad_bf16", ptr(x_nchw), int(x_nchw.dtype == torch.float16), ptr(dpooled16), ptr(amax), ptr(dw), ptr(dbias), B, Ci, H, W, Co, ptr(ws),
         ws.numel() * 4, stream())


def dropout(x: torch.Tensor, p: float, seed: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    y = out if out is not None else torch.empty_like(x)
    call("vqa_dropout", ptr(x), ptr(y), x.numel(), p, seed, stream())
    return y


def dropout_add(x: torch.Tensor, y: torch.Tensor, p: float, seed: int) -> torch.Tensor:
    """y += dropout_{p, seed}(x) in one pass."""
    call("vqa_dropout_add", ptr(x), ptr(y), x.numel(), p, seed, stream())
    return y


def l2norm_fwd(pooled: torch.Tensor, p: float, seed: int, drop2=None):
    """drop2 = (p2, seed2, dtype): also return dropout_{p2, seed2}(vn) in fp32 or bf16 (written in the same pass)."""
    C = pooled.shape[-1]
    rows = pooled.numel() // C
    vn = torch.empty_like(pooled)
    norm = torch.empty(rows, dtype=torch.float32, device=pooled.device)
    if drop2 is None:
        call("vqa_l2norm_fwd", ptr(pooled), ptr(vn), ptr(norm), rows, C, p, seed, None, 0, 0.0, 0, stream())
        return vn, norm
    p2, seed2, dtype = drop2
    vd = torch.empty(pooled.shape, dtype=dtype, device=pooled.device)
    call("vqa_l2norm_fwd", ptr(pooled), ptr(vn), ptr(norm), rows, C, p, seed, ptr(vd), int(dtype == torch.bfloat16), p2, seed2,
         stream())
    return vn, norm, vd


def l2norm_bwd(dvn, vn, norm, p: float, seed: int, out=None, out_dtype=torch.float32, c16_hw=None):
    """out_dtype=torch.bfloat16: the gradient is stored as bf16 (the bf16 path's pooled gradient of the last conv block);
    c16_hw=(Hp, Wp): bf16, channel-blocked [B, C/16, Hp, Wp, 16] (what the patch convolutions' backward kernels read)."""
    C = vn.shape[-1]
    rows = vn.numel() // C
    if c16_hw is not None:
        Hp, Wp = c16_hw
        assert out is None and rows % (Hp * Wp) == 0 and C % 16 == 0
        d = torch.empty(rows // (Hp * Wp), C // 16, Hp, Wp, 16, dtype=torch.bfloat16, device=vn.device)
        call("vqa_l2norm_bwd", ptr(dvn), ptr(vn), ptr(norm), ptr(d), 2, rows, Hp * Wp, C, p, seed, stream())
        return d
    d = out if out is not None else torch.empty(vn.shape, dtype=out_dtype, device=vn.device)
    call("vqa_l2norm_bwd", ptr(dvn), ptr(vn), ptr(norm), ptr(d), int(d.dtype == torch.bfloat16), rows, 0, C, p, seed, stream())
    return d


def l2norm_bwd_joined(dout, dout_ld, probs, dv_in, p_v: float, seed_v: int, vn, norm, p: float, seed: int,
                      out_dtype=torch.float32, c16_hw=None):
    """l2norm_bwd whose incoming gradient is joined in the kernel: sum_g probs[b,g,p] * dout[b, g*C:(g+1)*C] (what att_apply_bwd
    would have written) + dropout_{p_v, seed_v}-mask * dv_in (what dropout_add would have added)."""
    B, G, P = probs.shape
    C = vn.shape[-1]
    rows = vn.numel() // C
    assert rows == B * P and dv_in.numel() == rows * C and dv_in.dtype == torch.float32
    if c16_hw is not None:
        Hp, Wp = c16_hw
        assert Hp * Wp == P and C % 16 == 0
        d = torch.empty(B, C // 16, Hp, Wp, 16, dtype=torch.bfloat16, device=vn.device)
        mode = 2
    else:
        d = torch.empty(vn.shape, dtype=out_dtype, device=vn.device)
        mode = int(out_dtype == torch.bfloat16)
    call("vqa_l2norm_bwd_joined", ptr(dout), dout_ld, ptr(probs), G, ptr(dv_in), p_v, seed_v, ptr(vn), ptr(norm), ptr(d), mode,
         rows, P, C, p, seed, stream())
    return d


def embed_tanh_fwd(q: torch.Tensor, emb: torch.Tensor, p: float, seed: int,
                   bad_tokens: Optional[torch.Tensor] = None) -> torch.Tensor:
    """bad_tokens: optional device int32 [1], incremented once per token id outside [0, V)."""
    B, T = q.shape
    V, E = emb.shape
    x = torch.empty(T, B, E, dtype=torch.float32, device=emb.device)
    call("vqa_embed_tanh_fwd", ptr(q), ptr(emb), ptr(x), B, T, E, V, p, seed, ptr(bad_tokens), stream())
    return x


def embed_tanh_bwd(q, x, dx, demb, p: float, seed: int, binned: bool = True):
    """binned=False: the workspace-less scanning kernel (O(V * B*T)); same bits."""
    B, T = q.shape
    V, E = demb.shape
    if binned:
        ws = workspace(_lib.load().vqa_embed_tanh_bwd_workspace_bytes(B, T, V), q.device)
        call("vqa_embed_tanh_bwd", ptr(q), ptr(x), ptr(dx), ptr(demb), B, T, E, V, p, seed, ptr(ws), ws.numel() * 4, stream())
    else:
        call("vqa_embed_tanh_bwd", ptr(q), ptr(x), ptr(dx), ptr(demb), B, T, E, V, p, seed, None, 0, stream())


def lstm_cell_fwd(xg_t, hg, c_in, h_in, q_len, t, gates, c_out, h_out, c_final=None, cf_ld=0):
    B, H = c_in.shape
    call("vqa_lstm_cell_fwd", ptr(xg_t), ptr(hg), ptr(c_in), ptr(h_in), ptr(q_len), t, ptr(gates), ptr(c_out),
         ptr(h_out), ptr(c_final), cf_ld, B, H, stream())


def lstm_step_supported(H: int) -> bool:
    return bool(_lib.load().vqa_lstm_step_supported(H))


def _lstm_dirs(dirs):
    """[{w_hh, xg, gates, Hs, Cs, c_final, dgates, dh, dc, reverse}] -> ctypes array of vqa_lstm_dir_t."""
    arr = (_lib.LstmDir * len(dirs))()
    for k, d in enumerate(dirs):
        for f in ("w_hh", "xg", "gates", "Hs", "Cs", "c_final", "dgates", "dh", "dc"):
            setattr(arr[k], f, ptr(d.get(f)))
        arr[k].reverse = int(bool(d.get("reverse", False)))
    return arr


def lstm_seq_fwd(dirs, q_len, B: int, T: int, H: int, cf_ld: int = 0, use_graph: bool = True):
    """The whole recurrence, forward: one call = T launches (each covers every direction), optionally as a cached
    hipGraph.  dirs: list of dicts with tensors w_hh [4H,H], xg [T,B,4H], gates [T,B,4H], Hs / Cs [T+1,B,H]
    (initial slot zeroed by the caller), optional c_final [B, cf_ld], reverse."""
    import ctypes
    arr = _lstm_dirs(dirs)
    call("vqa_lstm_seq_fwd", ctypes.addressof(arr), len(dirs), ptr(q_len), B, T, H, cf_ld, int(use_graph), stream())


def lstm_seq_bwd(dirs, q_len, B: int, T: int, H: int, use_graph: bool = True):
    """BPTT of the recurrence: fills dgates [T,B,4H] per direction; dh (zeros) / dc (d loss / d c_n) are updated in
    place.  Needs gates, Hs, Cs of the forward pass and w_hh."""
    import ctypes
    arr = _lstm_dirs(dirs)
    call("vqa_lstm_seq_bwd", ctypes.addressof(arr), len(dirs), ptr(q_len), B, T, H, int(use_graph), stream())


def lstm_graph_stats():
    """(replays, builds, plain-launch fallbacks, cached graphs) of the library's LSTM-sequence hipGraph cache."""
    import ctypes
    a, b, c = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
    n = _lib.load().vqa_lstm_graph_stats(ctypes.byref(a), ctypes.byref(b), ctypes.byref(c))
    return a.value, b.value, c.value, n


def lstm_cell_bwd(gates, c_in, c_out, q_len, t, dh, dc, dgates):
    B, H = c_in.shape
    call("vqa_lstm_cell_bwd", ptr(gates), ptr(c_in), ptr(c_out), ptr(q_len), t, ptr(dh), ptr(dc), ptr(dgates),
         B, H, stream())


def att_score_fwd(xs, wx, bx, B, P, p: float, seed: int, qcat=None) -> torch.Tensor:
    """wx [G, xld] with xld = mid ('+', '*') or 2*mid ('|', qcat = q' [B, mid])."""
    G, xld = wx.shape[0], wx.shape[1]
    mid = xld // 2 if qcat is not None else xld
    score = torch.empty(B, G, P, dtype=torch.float32, device=xs.device)
    call("vqa_att_score_fwd", ptr(xs), int(xs.dtype == torch.bfloat16), ptr(wx), xld, ptr(bx), ptr(score), B, P, mid, G, p,
         seed, ptr(qcat), stream())
    return score


def att_score_bwd(dscore, wx, xs_inout, B, P, p: float, seed: int, mode: int = 0, vprime=None, qp=None):
    lib = _lib.load()
    G, xld = wx.shape[0], wx.shape[1]
    mid = xld // 2 if mode == 2 else xld
    RS = lib.vqa_att_row_splits(P)
    dwx_part = torch.empty(B * RS, G * xld, dtype=torch.float32, device=wx.device)
    dq_part = torch.empty(B * RS, mid, dtype=torch.float32, device=wx.device)
    call("vqa_att_score_bwd", ptr(dscore), ptr(wx), xld, ptr(xs_inout), int(xs_inout.dtype == torch.bfloat16), ptr(dwx_part),
         ptr(dq_part), B, P, mid, G, p, seed, mode, ptr(vprime), ptr(qp), stream())
    return dwx_part, dq_part, RS


def att_apply_fwd(score, vn, out, out_ld):
    B, G, P = score.shape
    C = vn.shape[-1]
    probs = torch.empty_like(score)
    call("vqa_att_apply_fwd", ptr(score), ptr(vn), ptr(probs), ptr(out), out_ld, B, P, C, G, stream())
    return probs


def att_apply_bwd(dout, dout_ld, probs, vn, dvn_out=None, rowsum=None, want_dvn=True):
    """rowsum: optional [B, G] output, sum over positions of dscore (per-sample x_conv bias gradient).
    want_dvn=False: the weighted-sum branch of d loss / d vn is not written (l2norm_bwd_joined recomputes it)."""
    B, G, P = probs.shape
    C = vn.shape[-1]
    dscore = torch.empty_like(probs)
    dvn = (dvn_out if dvn_out is not None else torch.empty_like(vn)) if want_dvn else None
    call("vqa_att_apply_bwd", ptr(dout), dout_ld, ptr(probs), ptr(vn), ptr(dscore), ptr(dvn), ptr(rowsum), B, P, C, G,
         stream())
    return dscore, dvn


def softce(logits, ld, a_idx, a_val, A, inv_batch, dlogits=None, dld=0):
    B = logits.shape[0]
    kmax = a_idx.shape[1]
    loss_rows = torch.empty(B, dtype=torch.float32, device=logits.device)
    score_rows = torch.empty(B, dtype=torch.float32, device=logits.device)
    call("vqa_softce_fwd_bwd", ptr(logits), ld, ptr(a_idx), ptr(a_val), kmax, B, A, inv_batch, ptr(loss_rows),
         ptr(score_rows), ptr(dlogits), dld, stream())
    return loss_rows, score_rows


def colsum(x: torch.Tensor, rows: int, cols: int, out: torch.Tensor, *, ld=None, mask=None, accumulate=False):
    lib = _lib.load()
    nbytes = lib.vqa_colsum_workspace_bytes(rows, cols)
    ws = workspace(nbytes, x.device)
    call("vqa_colsum", ptr(x), ld if ld is not None else cols, ptr(mask), rows, cols, ptr(out), int(accumulate),
         ptr(ws), ws.numel() * 4, stream())
    return out


def sum_bgp(x: torch.Tensor, out: torch.Tensor):
    B, G, P = x.shape
    call("vqa_sum_bgp", ptr(x), ptr(out), B, G, P, stream())
    return out


def sum_parts(part: torch.Tensor, out: torch.Tensor, batch: int, parts: int, cols: int):
    call("vqa_sum_parts", ptr(part), ptr(out), batch, parts, cols, stream())
    return out


def relu_drop_bwd(y, dy, dx, p: float, seed: int):
    call("vqa_relu_drop_bwd", ptr(y), ptr(dy), ptr(dx), y.numel(), p, seed, stream())
    return dx


def add2d(a, lda, b, ldb, out, ldo, rows, cols):
    """out[r, c] = a[r, c] + (b[r, c] if b is not None else 0) on strided row-major views."""
    call("vqa_add2d", ptr(a), lda, ptr(b), ldb, ptr(out), ldo, rows, cols, stream())
    return out


def add(a, b, out):
    return add2d(a, a.numel(), b, a.numel(), out, a.numel(), 1, a.numel())


def half_to_float(x: torch.Tensor) -> torch.Tensor:
    """fp16 image features -> fp32 on the device, layout unchanged (reference: host-side cast in
    preprocessing/data_preprocessing.py:167-176)."""
    assert x.dtype == torch.float16 and x.is_contiguous()
    y = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    call("vqa_half_to_float", ptr(x), ptr(y), x.numel(), stream())
    return y


# ---------------------------------------------------------------------------- bf16 path (BASELINE configs[3])
def to_bf16(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """fp32 -> bf16 copy (round to nearest even) by the library's converter."""
    assert x.dtype == torch.float32 and x.is_contiguous()
    y = out if out is not None else torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    call("vqa_f32_to_bf16", ptr(x), ptr(y), x.numel(), stream())
    return y


def dropout_to_bf16(x: torch.Tensor, p: float, seed: int) -> torch.Tensor:
    y = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    call("vqa_dropout_to_bf16", ptr(x), ptr(y), x.numel(), p, seed, stream())
    return y


def to_f32(x: torch.Tensor) -> torch.Tensor:
    assert x.dtype == torch.bfloat16 and x.is_contiguous()
    y = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    call("vqa_bf16_to_f32", ptr(x), ptr(y), x.numel(), stream())
    return y


def to_bf16_transposed(x: torch.Tensor) -> torch.Tensor:
    """[rows, cols] fp32 -> [cols, rows] bf16."""
    assert x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 2
    rows, cols = x.shape
    y = torch.empty(cols, rows, dtype=torch.bfloat16, device=x.device)
    call("vqa_f32_to_bf16_transpose", ptr(x), ptr(y), rows, cols, stream())
    return y


def gemm_bf16(A: torch.Tensor, B: torch.Tensor, C: torch.Tensor, M: int, N: int, K: int, *, transA=False, transB=True,
              lda=None, ldb=None, ldc=None, bias1=None, bias2=None, rowgroup=None, rg_div=1, rg_op=0, relu=False,
              accumulate=False, aux=None, tag=0) -> torch.Tensor:
    """vqa_gemm with bf16 A / B (fp32 accumulation); C fp32 or bf16 by its dtype."""
    lib = _lib.load()
    assert A.dtype == torch.bfloat16 and B.dtype == torch.bfloat16 and C.dtype in (torch.float32, torch.bfloat16)
    lda = lda if lda is not None else (M if transA else K)
    ldb = ldb if ldb is not None else (K if transB else N)
    ldc = ldc if ldc is not None else N
    nbytes = lib.vqa_gemm_bf16_workspace_bytes(M, N, K)
    ws = workspace(nbytes, A.device) if nbytes else None
    call("vqa_gemm_bf16", ptr(A), lda, int(transA), ptr(B), ldb, int(transB), ptr(C), ldc, int(C.dtype == torch.bfloat16),
         M, N, K, ptr(bias1), ptr(bias2), ptr(rowgroup), (rowgroup.stride(0) if rowgroup is not None else 0),
         rg_div, rg_op, int(relu), int(accumulate), ptr(aux), ptr(ws), (ws.numel() * 4 if ws is not None else 0),
         tag, stream())
    return C


def gemm_tall_bf16_supported(M: int, N: int, K: int, rg_div: int = 0, has_rowgroup: bool = False) -> bool:
    return bool(_lib.load().vqa_gemm_tall_bf16_supported(M, N, K, rg_div, int(has_rowgroup)))


def gemm_tall_bf16(A: torch.Tensor, W: torch.Tensor, C: torch.Tensor, M: int, N: int, K: int, *, rowgroup=None, rg_div=1,
                   rg_op=0, relu=False, tag=0) -> torch.Tensor:
    """C [M,N] bf16 = act(A [M,K] . W [N,K]^T (+|*) rowgroup[row // rg_div]) on persistent 256 x 128 tiles (short K)."""
    assert A.dtype == torch.bfloat16 and W.dtype == torch.bfloat16 and C.dtype == torch.bfloat16
    call("vqa_gemm_tall_bf16", ptr(A), A.stride(0), ptr(W), W.stride(0), ptr(C), C.stride(0), M, N, K, ptr(rowgroup),
         (rowgroup.stride(0) if rowgroup is not None else 0), rg_div, rg_op, int(relu), tag, stream())
    return C


def conv_pack_weights_bf16(w: torch.Tensor, CiP: int, need_wd: bool = True):
    """fp32 [Co,Ci,3,3] -> bf16 wfT [Co, 9*CiP] and wdT [CiP, 9*Co] (K index = (tap, channel))."""
    Co, Ci = w.shape[0], w.shape[1]
    wfT = torch.empty(Co, 9 * CiP, dtype=torch.bfloat16, device=w.device)
    wdT = torch.empty(CiP, 9 * Co, dtype=torch.bfloat16, device=w.device) if need_wd else None
    call("vqa_conv_pack_weights_bf16", ptr(w), ptr(wfT), ptr(wdT), Co, Ci, CiP, stream())
    return wfT, wdT


def conv_fwd_bf16(x: torch.Tensor, wfT: torch.Tensor, bias: torch.Tensor, stride: int = 1, out_dtype=torch.bfloat16,
                  tag: int = 0):
    """x NHWC bf16 [B,H,W,CiP] -> (pooled [B,Hp,Wp,Co] bf16 or fp32, argmax uint8)."""
    assert x.dtype == torch.bfloat16 and wfT.dtype == torch.bfloat16
    B, H, W, CiP = x.shape
    Co = wfT.shape[0]
    Hp, Wp = conv_out_hw(H, W, stride)
    pooled = torch.empty(B, Hp, Wp, Co, dtype=out_dtype, device=x.device)
    amax = torch.empty(B, Hp, Wp, Co, dtype=torch.uint8, device=x.device)
    call("vqa_conv3x3_relu_pool_fwd_bf16", ptr(x), ptr(wfT), ptr(bias), ptr(pooled), int(out_dtype == torch.bfloat16),
         ptr(amax), B, H, W, CiP, Co, stride, tag, stream())
    return pooled, amax


def conv_dgrad_bf16(dpooled, amax, wdT, x_shape, stride: int = 1, out_dtype=torch.bfloat16, tag: int = 0):
    assert dpooled.dtype == torch.bfloat16
    B, H, W, CiP = x_shape
    Co = dpooled.shape[3]
    dx = torch.empty(B, H, W, CiP, dtype=out_dtype, device=dpooled.device)
    call("vqa_conv3x3_dgrad_bf16", ptr(dpooled), ptr(amax), ptr(wdT), ptr(dx), int(out_dtype == torch.bfloat16), B, H, W,
         CiP, Co, stride, tag, stream())
    return dx


def conv_wgrad_bf16(x, dpooled, amax, dw: torch.Tensor, dbias: torch.Tensor, stride: int = 1, tag: int = 0):
    assert x.dtype == torch.bfloat16 and dpooled.dtype == torch.bfloat16
    lib = _lib.load()
    B, H, W, CiP = x.shape
    Co, Ci = dw.shape[0], dw.shape[1]
    ws = workspace(lib.vqa_conv3x3_wgrad_bf16_workspace_bytes(B, H, W, CiP, Co, stride), x.device)
    call("vqa_conv3x3_wgrad_bf16", ptr(x), ptr(dpooled), ptr(amax), ptr(dw), ptr(dbias), B, H, W, CiP, Ci, Co, stride,
         ptr(ws), ws.numel() * 4, tag, stream())


# ------------------------------------------------------------------ fp32 patch backward-data (csrc/conv_patch_f32.hip)
def pconvf_supported(H: int, W: int, Ci: int, Co: int, stride: int = 1) -> bool:
    return bool(_lib.load().vqa_pconvf_supported(H, W, Ci, Co, stride))


def pconvf_pack_weights(w: torch.Tensor) -> torch.Tensor:
    """fp32 [Co,Ci,3,3] -> the flipped, transposed, fragment-ordered fp32 image pconvf_dgrad reads."""
    Co, Ci = w.shape[0], w.shape[1]
    wd = torch.empty(9 * Ci * Co, dtype=torch.float32, device=w.device)
    call("vqa_pconvf_pack_weights", ptr(w), ptr(wd), Co, Ci, stream())
    return wd


def pconvf_dgrad(dpooled: torch.Tensor, amax: torch.Tensor, wd_img: torch.Tensor, x_shape, tag: int = 0, out=None) -> torch.Tensor:
    """dpooled fp32 NHWC [B,Hp,Wp,Co] + arg-max bytes NHWC -> dX fp32 NHWC; x_shape = (B, H, W, Ci) of the block's input."""
    B, H, W, Ci = x_shape
    Co = dpooled.shape[3]
    assert dpooled.dtype == torch.float32 and dpooled.is_contiguous() and amax.dtype == torch.uint8 and amax.is_contiguous()
    assert tuple(dpooled.shape[1:3]) == conv_out_hw(H, W, 1) and amax.shape == dpooled.shape
    dx = out if out is not None else torch.empty(B, H, W, Ci, dtype=torch.float32, device=dpooled.device)
    call("vqa_pconvf_dgrad", ptr(dpooled), ptr(amax), ptr(wd_img), ptr(dx), B, H, W, Ci, Co, tag, stream())
    return dx


# ------------------------------------------------------------------ bf16 patch convolutions (csrc/conv_patch_bf16.hip)
# Everything a patch is cut from lives in HBM channel-blocked, "C16" = [B][C/16][H][W][16] bf16 (a K-slice of a patch row is
# one contiguous run); pooled fp32 outputs, dX, pooled gradients and arg-max bytes stay NHWC.
def to_c16(x_nhwc: torch.Tensor) -> torch.Tensor:
    """NHWC -> C16 by torch (tests / tools; the kernels write C16 themselves)."""
    B, H, W, C = x_nhwc.shape
    return x_nhwc.view(B, H, W, C // 16, 16).permute(0, 3, 1, 2, 4).contiguous()


def from_c16(x_c16: torch.Tensor) -> torch.Tensor:
    B, Cb, H, W, _ = x_c16.shape
    return x_c16.permute(0, 2, 3, 1, 4).reshape(B, H, W, Cb * 16).contiguous()


def pconv_supported(H: int, W: int, Ci: int, Co: int, stride: int = 1) -> bool:
    return bool(_lib.load().vqa_pconv_supported(H, W, Ci, Co, stride))


def pconv_pack_weights(w: torch.Tensor, need_wd: bool = True):
    """fp32 [Co,Ci,3,3] -> fragment-ordered bf16 images (forward; flipped + transposed for backward-data)."""
    Co, Ci = w.shape[0], w.shape[1]
    n = 9 * Ci * Co
    wf = torch.empty(n, dtype=torch.bfloat16, device=w.device)
    wd = torch.empty(n, dtype=torch.bfloat16, device=w.device) if need_wd else None
    call("vqa_pconv_pack_weights", ptr(w), ptr(wf), ptr(wd), Co, Ci, stream())
    return wf, wd


def pconv_fwd(x: torch.Tensor, wf_img: torch.Tensor, bias: torch.Tensor, Co: int, out_dtype=torch.bfloat16, tag: int = 0):
    """x C16 bf16 [B,Ci/16,H,W,16] -> (pooled: bf16 C16 [B,Co/16,Hp,Wp,16] or fp32 NHWC [B,Hp,Wp,Co]; argmax uint8 C16)."""
    assert x.dtype == torch.bfloat16 and wf_img.dtype == torch.bfloat16 and x.is_contiguous() and x.dim() == 5
    B, Cb, H, W, _ = x.shape
    Hp, Wp = conv_out_hw(H, W, 1)
    if out_dtype == torch.bfloat16:
        pooled = torch.empty(B, Co // 16, Hp, Wp, 16, dtype=torch.bfloat16, device=x.device)
    else:
        pooled = torch.empty(B, Hp, Wp, Co, dtype=out_dtype, device=x.device)
    amax = torch.empty(B, Co // 16, Hp, Wp, 16, dtype=torch.uint8, device=x.device)
    call("vqa_pconv_fwd", ptr(x), ptr(wf_img), ptr(bias), ptr(pooled), int(out_dtype == torch.bfloat16), ptr(amax),
         B, H, W, Cb * 16, Co, tag, stream())
    return pooled, amax


def pconv_dgrad(dpooled: torch.Tensor, amax: torch.Tensor, wd_img: torch.Tensor, x_shape, out_dtype=torch.bfloat16,
                out_c16: bool = False, tag: int = 0):
    """Pooled gradient (bf16 C16 [B,Co/16,Hp,Wp,16]) + arg-max bytes (C16) -> dX of the block whose input is x_shape =
    (B, H, W, Ci): NHWC (bf16 or fp32), or bf16 C16 (the pooled gradient of the block below).  The pre-pool gradient is routed
    inside the kernel."""
    B, H, W, Ci = x_shape
    assert dpooled.dtype == torch.bfloat16 and dpooled.dim() == 5 and dpooled.is_contiguous() and dpooled.shape[0] == B
    assert amax.dtype == torch.uint8 and amax.shape == dpooled.shape and amax.is_contiguous()
    assert tuple(dpooled.shape[2:4]) == conv_out_hw(H, W, 1)
    Co = dpooled.shape[1] * 16
    if out_c16:
        assert out_dtype == torch.bfloat16
        dx = torch.empty(B, Ci // 16, H, W, 16, dtype=torch.bfloat16, device=dpooled.device)
    else:
        dx = torch.empty(B, H, W, Ci, dtype=out_dtype, device=dpooled.device)
    mode = 2 if out_c16 else int(out_dtype == torch.bfloat16)
    call("vqa_pconv_dgrad", ptr(dpooled), ptr(amax), ptr(wd_img), ptr(dx), mode, B, H, W, Ci, Co, tag, stream())
    return dx


def pconv_wgrad_supported(H: int, W: int, Ci: int, Co: int) -> bool:
    return bool(_lib.load().vqa_pconv_wgrad_supported(H, W, Ci, Co))


def pconv_wgrad(x, dpooled, amax, dw: torch.Tensor, dbias: torch.Tensor, tag: int = 0):
    """dw [Co,Ci,3,3], dbias [Co] (fp32) from x (C16 bf16), the pooled gradient (C16 bf16) and the arg-max bytes (C16)."""
    assert x.dtype == torch.bfloat16 and dpooled.dtype == torch.bfloat16 and amax.dtype == torch.uint8
    assert x.is_contiguous() and dpooled.is_contiguous() and amax.is_contiguous() and amax.shape == dpooled.shape
    lib = _lib.load()
    B, Cib, H, W, _ = x.shape
    Ci, Co = Cib * 16, dpooled.shape[1] * 16
    assert tuple(dpooled.shape[2:4]) == conv_out_hw(H, W, 1) and dpooled.shape[0] == B
    ws = workspace(lib.vqa_pconv_wgrad_workspace_bytes(B, H, W, Ci, Co), x.device)
    call("vqa_pconv_wgrad", ptr(x), ptr(dpooled), ptr(amax), ptr(dw), ptr(dbias), B, H, W, Ci, Co, ptr(ws), ws.numel() * 4, tag,
         stream())


def scale_by(x, scalar_dev):
    call("vqa_scale_by", ptr(x), x.numel(), ptr(scalar_dev), stream())
    return x


def adam(param, grad, exp_avg, exp_avg_sq, lr, step, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0):
    call("vqa_adam", ptr(param), ptr(grad), ptr(exp_avg), ptr(exp_avg_sq), param.numel(), lr, beta1, beta2, eps,
         step, grad_scale, stream())
