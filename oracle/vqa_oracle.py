"""CPU oracle for the VqaNet train-step hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, on the CPU, the arithmetic of the reference's hot path
(OmerShubi/DL_VQA ``models/model.py`` + the loss/metric/optimiser lines of
``train.py`` / ``utils/train_utils.py``).  It exists so the HIP kernels can be
checked against an independent implementation on the GPU box, where the
reference itself cannot travel.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it.  The product path (``dl_vqa_amd``) never does: it calls the
HIP kernels through the C ABI and fails loudly when the library is missing.

Parity pin: the reference publishes no tests or golden vectors,
so the oracle is pinned by fixtures under ``tests/golden/`` that were produced
by importing the reference's ``models/model.py`` in the build container
(``tests/golden/make_golden.py``).  ``tests/test_oracle_golden.py`` checks the
oracle against every one of them.

Every op is written out with basic tensor algebra (conv through ``F.conv2d``
is the only library contraction; the LSTM is an explicit masked loop rather
than ``nn.LSTM``) so that the intermediate quantities the HIP path saves for
backward (pool arg-max, gate activations, attention probabilities) have a
named counterpart here.  ``dtype=torch.float64`` gives a higher-precision
reference for error budgeting.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------
# image encoder  — reference models/model.py:72-84 (ImageNet2)
# --------------------------------------------------------------------------
def conv_relu_pool(x: Tensor, w: Tensor, b: Tensor, stride: int = 1) -> Tensor:
    """Conv2d(k, stride, pad=0) -> ReLU -> MaxPool2d(2,2)  (model.py:80-82)."""
    y = F.conv2d(x, w, b, stride=stride)
    y = torch.relu(y)
    return F.max_pool2d(y, 2, 2)


# Dropout (7 sites, model.py:84,156,185,186,194,201,204).  torch's RNG stream cannot be reproduced off-torch, so
# train mode is restated with the masks as DATA: `masks[site]` is the keep-scale tensor (0 or 1/(1-p)) of a site,
# multiplied in exactly where the reference applies nn.Dropout; a missing key / masks=None is eval mode.
#   "image" [B,C,g,g]   "text" [B,T,E]   "att_v" [B,C,g,g]   "att_q" [B,Q]   "att_x" [B,mid or 2*mid,g,g]
#   "cls1" [B,G*C+Q]    "cls2" [B,hid]
MASK_SITES = ("image", "text", "att_v", "att_q", "att_x", "cls1", "cls2")


def _drop(x: Tensor, masks: Optional[dict], site: str) -> Tensor:
    if masks is None or masks.get(site) is None:
        return x
    return x * masks[site].to(x.dtype)


# bf16 path (BASELINE configs[3]): the same oracle with the operands of the re-typed contractions rounded to
# bfloat16 where the HIP path stores / stages them as bf16 (round to nearest even; products of two bf16 values are
# exact in fp32, accumulation stays fp32): the image and the activations between conv blocks, the conv weights, the
# v_conv input and weight, the inputs and weights of q_lin / lin1 / lin2 (when their dimensions allow: fc16_dims_ok); in
# backward, the gradients the HIP path stores as bf16 (dP of the conv blocks >= 1, dx') or stages as bf16 (the output
# gradients of the three linear layers).
def rb(x: Tensor) -> Tensor:
    return x.to(torch.bfloat16).to(x.dtype)


class _RoundFB(torch.autograd.Function):
    """identity whose forward value and / or backward gradient are rounded to bf16"""

    @staticmethod
    def forward(ctx, x, fwd, bwd):
        ctx.bwd = bwd
        return rb(x) if fwd else x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return (rb(g) if ctx.bwd else g), None, None


class _FirstConvBf16(torch.autograd.Function):
    """First conv block of the bf16 path: both products run on bf16 MFMA with the image rounded to bf16 where it is
    consumed (forward: image and weight; weight gradient: image and the bf16-stored output gradient), csrc/conv0.hip."""

    @staticmethod
    def forward(ctx, x, w, b, stride):
        ctx.save_for_backward(x, w)
        ctx.stride = stride
        return F.conv2d(rb(x), rb(w), b, stride=stride)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gw = torch.nn.grad.conv2d_weight(rb(x), w.shape, gy, stride=ctx.stride)
        return None, gw, gy.sum(dim=(0, 2, 3)), None


def _rfb(x: Tensor, fwd: bool, bwd: bool, on: bool) -> Tensor:
    return _RoundFB.apply(x, fwd, bwd) if on else x


def _w16(w: Tensor, on: bool) -> Tensor:
    """bf16 copy of an fp32 master weight (straight-through: the gradient goes to the master weight unchanged)"""
    return w + (rb(w) - w).detach() if on else w


def fc16_dims_ok(sd: Dict[str, Tensor]) -> bool:
    """bf16 path: q_lin / lin1 / lin2 run on bf16 MFMA when every one of their dimensions is a multiple of 8 (the bf16 GEMM's
    16-byte operand rows); the same rule as dl_vqa_amd.engine.Engine.fc16, including its A/B switch VQA_FC16=0 (fp32 FC layers)."""
    import os
    if os.environ.get("VQA_FC16", "1") == "0":
        return False
    dims = list(sd["attention.q_lin.weight"].shape) + list(sd["classifier.lin1.weight"].shape) + \
        list(sd["classifier.lin2.weight"].shape)
    return all(int(d) % 8 == 0 for d in dims)


def _linear16(x: Tensor, w: Tensor, b: Tensor, on: bool) -> Tensor:
    """nn.Linear on the bf16 path: input and weight rounded to bf16 where the GEMM stages them, fp32 accumulation and fp32
    output; in backward the output gradient is rounded to bf16 before both products (dW = dy^T x, dx = dy W)."""
    y = F.linear(_rfb(x, fwd=True, bwd=False, on=on), _w16(w, on))
    return _rfb(y, fwd=False, bwd=True, on=on) + b          # the bias gradient is the fp32 column sum of the unrounded dy


def image_encoder(sd: Dict[str, Tensor], v: Tensor, stride: int = 1,
                  stages: Optional[dict] = None, masks: Optional[dict] = None, bf16: bool = False) -> Tensor:
    """ImageNet2.forward (model.py:79-84): the conv blocks, then image.drop on the last pooled map."""
    n = sum(1 for k in sd if k.startswith("image.conv") and k.endswith(".weight"))
    for i in range(n):
        w, b = sd[f"image.conv{i}.weight"], sd[f"image.conv{i}.bias"]
        if bf16 and i == 0:
            v = F.max_pool2d(torch.relu(_FirstConvBf16.apply(v, w, b, stride)), 2, 2)
        else:
            v = conv_relu_pool(v, _w16(w, bf16), b, stride)
        # bf16 path: pooled maps between blocks are stored as bf16 (the last one stays fp32); the gradient that
        # enters a block's backward is stored as bf16
        v = _rfb(v, fwd=i < n - 1, bwd=True, on=bf16)
        if stages is not None:
            stages[f"pool{i}"] = v
    return _drop(v, masks, "image")                                  # model.py:84


def l2_normalise(v: Tensor) -> Tensor:
    """v / (||v||_2 over channels + 1e-12)   (model.py:56)."""
    return v / (v.norm(p=2, dim=1, keepdim=True).expand_as(v) + 1e-12)


# --------------------------------------------------------------------------
# question encoder — reference models/model.py:134-166 (questionNet)
# --------------------------------------------------------------------------
def lstm16_ok(H: int, rows: int) -> bool:
    """bf16 path: the LSTM's non-recurrent products -- xg = x . W_ih^T, dW_hh = dgates^T . h, dW_ih = dgates^T . x,
    dx = dgates . W_ih -- run on bf16 MFMA (operands staged as bf16, fp32 accumulation) when the bf16 GEMM's 16-byte rows
    allow it; the same rule as dl_vqa_amd.engine.Engine (A/B switch VQA_LSTM16=0).  The recurrence h . W_hh^T, the cells and
    the gradient that flows back through time stay fp32."""
    import os
    return os.environ.get("VQA_LSTM16", "1") != "0" and H % 8 == 0 and rows % 8 == 0


class _HhProduct(torch.autograd.Function):
    """h . W_hh^T of one LSTM step; with `on` the WEIGHT gradient is taken from bf16-rounded operands (the HIP path stages
    dgates and the saved h as bf16 for that one product), the gradient w.r.t. h stays fp32."""

    @staticmethod
    def forward(ctx, h, w, on):
        ctx.save_for_backward(h, w)
        ctx.on = on
        return h @ w.t()

    @staticmethod
    def backward(ctx, gy):
        h, w = ctx.saved_tensors
        gw = rb(gy).t() @ rb(h) if ctx.on else gy.t() @ h
        return gy @ w, gw, None


def lstm_direction(x: Tensor, q_len: Tensor, w_ih: Tensor, w_hh: Tensor,
                   b_ih: Tensor, b_hh: Tensor, reverse: bool, whh16: bool = False) -> Tuple[Tensor, Tensor]:
    """One direction of nn.LSTM over a packed batch, as a masked loop.

    x [B,T,E]; sample b is updated only at steps t < q_len[b]; the reverse
    direction visits t = len-1 .. 0 per sample, which with masking is the same
    as visiting t = T-1 .. 0 and skipping t >= len.  Gate order i,f,g,o
    (torch.nn.LSTM; used by model.py:145-149, packed at model.py:159-162).
    Returns (h_n, c_n) [B,H].
    """
    B, T, _ = x.shape
    H = w_hh.shape[1]
    h = x.new_zeros(B, H)
    c = x.new_zeros(B, H)
    steps = range(T - 1, -1, -1) if reverse else range(T)
    for t in steps:
        # bf16 path (whh16): x and W_ih staged as bf16 for xg; the rounded output gradient feeds dW_ih and dx (as _linear16)
        xg = _rfb(F.linear(_rfb(x[:, t], fwd=True, bwd=False, on=whh16), _w16(w_ih, whh16)), fwd=False, bwd=True, on=whh16)
        gates = xg + _HhProduct.apply(h, w_hh, whh16) + b_ih + b_hh
        i, f, g, o = gates.split(H, dim=1)
        i, f, o = torch.sigmoid(i), torch.sigmoid(f), torch.sigmoid(o)
        g = torch.tanh(g)
        c_new = f * c + i * g
        h_new = o * torch.tanh(c_new)
        m = (q_len > t).to(x.dtype).unsqueeze(1)
        c = m * c_new + (1 - m) * c
        h = m * h_new + (1 - m) * h
    return h, c


def question_encoder(sd: Dict[str, Tensor], q: Tensor, q_len: Tensor,
                     bidirectional: bool = True, masks: Optional[dict] = None, bf16: bool = False) -> Tensor:
    """questionNet.forward: embedding(pad 0) -> dropout -> tanh -> LSTM -> c_n.

    Returns [B, 2H] = [c_fwd | c_bwd]  (model.py:164-166: c_n.transpose(0,1).flatten(1)).
    """
    emb = sd["text.embedding.weight"]
    # padding_idx=0 (model.py:138-140): row 0 is read as stored but receives no gradient
    x = torch.tanh(_drop(F.embedding(q, emb, padding_idx=0), masks, "text"))       # model.py:155-157
    outs = []
    w16 = bf16 and lstm16_ok(sd["text.lstm.weight_hh_l0"].shape[1], x.shape[0] * x.shape[1])
    _, c = lstm_direction(x, q_len, sd["text.lstm.weight_ih_l0"], sd["text.lstm.weight_hh_l0"],
                          sd["text.lstm.bias_ih_l0"], sd["text.lstm.bias_hh_l0"], False, w16)
    outs.append(c)
    if bidirectional:
        _, c = lstm_direction(x, q_len, sd["text.lstm.weight_ih_l0_reverse"],
                              sd["text.lstm.weight_hh_l0_reverse"],
                              sd["text.lstm.bias_ih_l0_reverse"],
                              sd["text.lstm.bias_hh_l0_reverse"], True, w16)
        outs.append(c)
    return torch.cat(outs, dim=1)


# --------------------------------------------------------------------------
# attention — reference models/model.py:169-195, 208-231
# --------------------------------------------------------------------------
def attention_scores(sd: Dict[str, Tensor], v: Tensor, q: Tensor, do_option: str = "+",
                     masks: Optional[dict] = None, bf16: bool = False) -> Tensor:
    """Attention.forward (model.py:183-195).  v [B,C,g,g], q [B,Q] -> [B,G,g,g]."""
    wv = _w16(sd["attention.v_conv.weight"], bf16)          # [mid, C, 1, 1], no bias (model.py:173)
    v_in = _rfb(_drop(v, masks, "att_v"), fwd=True, bwd=False, on=bf16)                          # model.py:185
    vv = _rfb(torch.einsum("bchw,mc->bmhw", v_in, wv[:, :, 0, 0]), fwd=False, bwd=True, on=bf16)
    qq = _linear16(_drop(q, masks, "att_q"), sd["attention.q_lin.weight"], sd["attention.q_lin.bias"],
                   bf16 and fc16_dims_ok(sd))                                                     # model.py:186
    qq = qq[:, :, None, None].expand_as(vv)     # tile_question_over_image (model.py:224-231)
    # bf16 path: x = relu(v' (+|*) q') is stored as bf16 (the v' half only for '|': the q' half is never materialised)
    if do_option == "*":
        x = _rfb(torch.relu(vv * qq), fwd=True, bwd=False, on=bf16)
    elif do_option == "+":
        x = _rfb(torch.relu(vv + qq), fwd=True, bwd=False, on=bf16)
    elif do_option == "|":
        x = torch.cat([_rfb(torch.relu(vv), fwd=True, bwd=False, on=bf16), torch.relu(qq)], dim=1)
    else:
        raise ValueError(do_option)
    wx = sd["attention.x_conv.weight"][:, :, 0, 0]
    x = _drop(x, masks, "att_x")                                                                     # model.py:194
    return torch.einsum("bmhw,gm->bghw", x, wx) + sd["attention.x_conv.bias"][None, :, None, None]


def image_question_attention(v: Tensor, att: Tensor) -> Tuple[Tensor, Tensor]:
    """softmax over positions per glimpse, weighted sum of v (model.py:208-221).

    Returns ([B, G*C] glimpse-major, probabilities [B,G,P])."""
    B, C = v.shape[:2]
    G = att.shape[1]
    vf = v.reshape(B, C, -1)
    p = torch.softmax(att.reshape(B, G, -1), dim=-1)
    out = torch.einsum("bgp,bcp->bgc", p, vf).reshape(B, G * C)
    return out, p


def classifier(sd: Dict[str, Tensor], x: Tensor, masks: Optional[dict] = None, bf16: bool = False) -> Tensor:
    """Classifier: Dropout -> Linear -> ReLU -> Dropout -> Linear (model.py:198-205)."""
    on = bf16 and fc16_dims_ok(sd)
    h = torch.relu(_linear16(_drop(x, masks, "cls1"), sd["classifier.lin1.weight"], sd["classifier.lin1.bias"], on))
    return _linear16(_drop(h, masks, "cls2"), sd["classifier.lin2.weight"], sd["classifier.lin2.bias"], on)


def vqa_forward(sd: Dict[str, Tensor], cfg: dict, v: Tensor, q: Tensor, q_len: Tensor,
                stages: Optional[dict] = None, masks: Optional[dict] = None, bf16: bool = False) -> Tensor:
    """VqaNet.forward (model.py:53-67). Returns logits [B, max_answers].
    masks=None: eval mode; masks = {site: keep-scale tensor}: train mode with those dropout masks (MASK_SITES);
    bf16: the bf16 path's rounding points (see rb above)."""
    img = image_encoder(sd, v, cfg["image"]["stride"], stages, masks, bf16)
    vn = l2_normalise(img)
    qf = question_encoder(sd, q, q_len, cfg["text"]["bidirectional"], masks, bf16)
    att = attention_scores(sd, vn, qf, cfg["attention"]["do_option"], masks, bf16)
    wv, probs = image_question_attention(vn, att)              # the weighted sum sees v WITHOUT attention.drop
    logits = classifier(sd, torch.cat([wv, qf], dim=1), masks, bf16)
    if stages is not None:
        stages.update(image=img, vnorm=vn, question=qf, attention=att, probs=probs,
                      weighted=wv, logits=logits)
    return logits


# --------------------------------------------------------------------------
# loss / metric — reference train.py:189-207, utils/train_utils.py:12-25
# --------------------------------------------------------------------------
def soft_ce_loss(logits: Tensor, a_indices: Tensor, a_values: Tensor) -> Tensor:
    """loss = sum_{b,k: a_idx[b,k]!=0} -log_softmax(logits)[b, a_idx[b,k]-1] * a_val[b,k]/10 / B.

    train.py:190-206: answers are 1-based, 0 is padding; counts are divided by 10.
    """
    nll = -torch.log_softmax(logits, dim=1)
    B = logits.shape[0]
    mask = a_indices != 0
    idx = (a_indices - 1).clamp(min=0)
    picked = torch.gather(nll, 1, idx)
    w = (a_values.to(logits.dtype) / 10.0) * mask.to(logits.dtype)
    return (picked * w).sum() / B


def batch_accuracy(logits: Tensor, a_indices: Tensor, a_values: Tensor) -> Tensor:
    """sum_b min(1, 0.3 * count of the arg-max answer)   (train_utils.py:12-25)."""
    pred = logits.argmax(dim=1, keepdim=True) + 1          # back to 1-based
    hit = (a_indices == pred) & (a_indices != 0)
    agreeing = (a_values * hit).sum(dim=1).to(torch.float32)
    return (agreeing * 0.3).clamp(max=1).sum()


def learning_rate(initial_lr: float, iteration: int, halflife: int = 50000) -> float:
    """train.py:31-35: lr = lr0 * 0.5 ** (iteration / 50000)."""
    return initial_lr * 0.5 ** (float(iteration) / halflife)


def adam_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, step: int, lr: float,
              beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8) -> None:
    """torch.optim.Adam defaults, no weight decay, no amsgrad (train.py:55,80). In place."""
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)


# --------------------------------------------------------------------------
# whole train step on the CPU (used as bench.py's cpu_baseline "port")
# --------------------------------------------------------------------------
def loss_and_grads(sd: Dict[str, Tensor], cfg: dict, v: Tensor, q: Tensor, q_len: Tensor,
                   a_indices: Tensor, a_values: Tensor,
                   loss_scale_batch: Optional[int] = None, masks: Optional[dict] = None, bf16: bool = False
                   ) -> Tuple[Tensor, Tensor, Dict[str, Tensor]]:
    """Forward + soft-CE + backward by autograd over the restated ops.

    ``loss_scale_batch`` overrides the divisor B (used by the data-parallel
    tests, where each rank divides by the GLOBAL batch); ``masks``: train mode, see vqa_forward."""
    params = {k: t.detach().clone().requires_grad_(True) for k, t in sd.items()}
    logits = vqa_forward(params, cfg, v, q, q_len, masks=masks, bf16=bf16)
    loss = soft_ce_loss(logits, a_indices, a_values)
    if loss_scale_batch is not None:
        loss = loss * (logits.shape[0] / float(loss_scale_batch))
    grads = torch.autograd.grad(loss, list(params.values()), allow_unused=True)
    out = {}
    for (k, p), g in zip(params.items(), grads):
        out[k] = torch.zeros_like(p) if g is None else g
    return logits.detach(), loss.detach(), out


def synthetic_batch(B: int, S: int, T: int, V: int, A: int, seed: int = 1, full_len: bool = False,
                    kmax: int = 3):
    """Synthetic 7-tuple in the dataset's layout (data_preprocessing.py:74-87), SURVEY §8d.

    v ~ N(0,1) f32 [B,3,S,S]; q uniform in [1,V) zero-padded past q_len; 1..kmax unique 1-based
    answers per sample with counts summing to <= 10."""
    g = torch.Generator().manual_seed(seed)
    v = torch.randn(B, 3, S, S, generator=g)
    q_len = torch.full((B,), T, dtype=torch.int64) if full_len else \
        torch.randint(1, T + 1, (B,), generator=g)
    q = torch.randint(1, V, (B, T), generator=g)
    q = q * (torch.arange(T)[None, :] < q_len[:, None])
    a_len = torch.randint(1, kmax + 1, (B,), generator=g)
    a_idx = torch.zeros(B, kmax, dtype=torch.int64)
    a_val = torch.zeros(B, kmax, dtype=torch.int64)
    for b in range(B):
        k = int(a_len[b])
        idx = torch.randperm(A, generator=g)[:k] + 1
        cnt = torch.randint(1, 4, (k,), generator=g)
        a_idx[b, :k] = idx
        a_val[b, :k] = cnt
    return v, q, a_idx, a_val, a_len, torch.arange(B), q_len
