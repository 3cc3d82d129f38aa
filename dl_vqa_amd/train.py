"""The per-batch half of the reference training procedure on the device (train.py:172-208,
utils/train_utils.py:12-25, train.py:31-35,55,80).

``run_batch(model, log_softmax, batch_data, max_answers)``, ``batch_accuracy`` and
``update_learning_rate`` keep the reference's names and argument meaning, so the reference's own
epoch loop (train.py:38-169: ``train`` / ``evaluate``, out of scope here and not restated) calls
them unchanged.  What differs is where the work runs: the soft-target cross entropy, its gradient
and the VQA score are one HIP kernel on the device (the reference builds numpy index arrays on
the host and syncs B+1 times per step, train.py:195-199, train_utils.py:19-23), and ``FusedAdam``
is torch.optim.Adam's update as one kernel over the flat parameter buffer (train.py:55,80).
"""
from __future__ import annotations

from typing import Optional

import torch

from . import ops
from .model import validate_question_lengths


# ------------------------------------------------------------------ loss head
class _SoftCEFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, a_indices, a_values, inv_batch):
        B, A = logits.shape
        dl = torch.empty(B, (A + 3) // 4 * 4, dtype=torch.float32, device=logits.device)
        need_grad = ctx.needs_input_grad[0]
        loss_rows, score_rows = ops.softce(logits, logits.stride(0), a_indices, a_values, A, inv_batch,
                                           dl if need_grad else None, dl.stride(0))
        out = torch.empty(2, dtype=torch.float32, device=logits.device)
        ops.colsum(loss_rows, B, 1, out[0:1])
        ops.colsum(score_rows, B, 1, out[1:2])
        ctx.dl = dl if need_grad else None
        ctx.A = A
        loss, score = out[0], out[1]
        ctx.mark_non_differentiable(score)
        return loss, score

    @staticmethod
    def backward(ctx, g_loss, g_score):
        dl = ctx.dl
        ops.scale_by(dl, g_loss.contiguous())
        return dl[:, :ctx.A], None, None, None


def soft_ce_loss_and_score(logits: torch.Tensor, a_indices: torch.Tensor, a_values: torch.Tensor,
                           batch_divisor: Optional[int] = None):
    """(loss, score): train.py:190-206 and train_utils.py:12-25 on the device.

    loss = sum_{b,k} -log_softmax(logits)[b, a_idx[b,k]-1] * a_val[b,k]/10 / batch_divisor
    score = sum_b min(1, 0.3 * count of the arg-max answer)."""
    dev = logits.device
    a_indices = a_indices.to(device=dev, dtype=torch.int64).contiguous()
    a_values = a_values.to(device=dev, dtype=torch.int64).contiguous()
    div = float(batch_divisor if batch_divisor is not None else logits.shape[0])
    return _SoftCEFunction.apply(logits, a_indices, a_values, 1.0 / div)


def batch_accuracy(predicted, true):
    """utils/train_utils.py:12-25 signature: true = (indices, values, size)."""
    indices, values, _size = true
    dev = predicted.device
    _, score = soft_ce_loss_and_score(predicted.detach(), indices.to(dev), values.to(dev))
    return score


def run_batch(model, log_softmax, batch_data, max_answers, batch_divisor: Optional[int] = None):
    """Reference signature (train.py:172-208). `log_softmax` is accepted and unused: the fused loss
    kernel computes log-softmax, the sparse soft-target cross entropy and the score in one pass.

    The loss is divided by the batch (train.py:206).  Under data parallelism (the model carries a
    dl_vqa_amd.distributed.DataParallel synchroniser) the divisor defaults to the GLOBAL batch, local batch x
    world size, because the gradients are SUM-all-reduced: the reference's own loop, which calls
    run_batch(model, log_softmax, batch_data, max_answers) with no divisor (train.py:70-73), then trains on the
    mean over the global batch exactly as it does on one GPU."""
    v, q, a_indices, a_values, a_length, idx, q_len = batch_data
    dev = next(model.parameters()).device
    if batch_divisor is None:
        sync = getattr(model, "_grad_sync", None)
        if sync is not None:
            batch_divisor = int(v.shape[0]) * int(sync.world_size)
    v = v.to(dev, non_blocking=True)
    q = q.to(dev, non_blocking=True)
    a_indices = a_indices.to(dev, non_blocking=True)
    a_values = a_values.to(dev, non_blocking=True)
    validate_question_lengths(q_len, q.shape[1])       # host-resident lengths: the error pack_padded_sequence would raise
    q_len = q_len.to(dev, non_blocking=True)
    # the dataset stores fp16 features (data_preprocessing.py:174 casts them on the host, per sample): the fp16
    # batch goes over PCIe as is and the first-block kernels read it as it is (widened where their LDS patch is staged)
    if v.dtype not in (torch.float32, torch.float16):
        v = v.float()
    y_hat = model(v, q, q_len)
    batch_loss, batch_score = soft_ce_loss_and_score(y_hat, a_indices, a_values, batch_divisor)
    return batch_loss, batch_score


# ------------------------------------------------------------------ optimiser
def update_learning_rate(optimizer, iteration, initial_lr):
    """train.py:31-35."""
    lr_halflife = 50000
    lr = initial_lr * 0.5 ** (float(iteration) / lr_halflife)
    for param_group in optimizer.param_groups:
        param_group["lr"] = lr


class FusedAdam:
    """torch.optim.Adam(model.parameters(), lr) defaults (betas 0.9/0.999, eps 1e-8, no weight decay)
    as ONE kernel over the model's flat parameter / gradient buffers."""

    def __init__(self, model, lr: float, betas=(0.9, 0.999), eps: float = 1e-8):
        self.model = model
        self.param_groups = [{"lr": lr, "betas": betas, "eps": eps}]
        self.step_count = 0
        self.exp_avg = None
        self.exp_avg_sq = None

    def _state(self):
        flat_p, flat_g, _ = self.model.flat_buffers()
        if self.exp_avg is None or self.exp_avg.numel() != flat_p.numel() or self.exp_avg.device != flat_p.device:
            self.exp_avg = torch.zeros_like(flat_p)
            self.exp_avg_sq = torch.zeros_like(flat_p)
        return flat_p, flat_g

    def zero_grad(self, set_to_none: bool = True):
        """torch.optim.Optimizer.zero_grad semantics.  set_to_none=True (torch's default) drops the gradients:
        the next backward then writes straight into the flat buffer (the fast path, and the one whose bucket
        all-reduces see the model's own buffer).  set_to_none=False zeroes existing gradients in place, as torch
        does; the next backward then goes through a fresh buffer that autograd ADDS to them (correct, one extra
        buffer pass per step)."""
        for p in self.model.parameters():
            if set_to_none or p.grad is None:
                p.grad = None
            else:
                p.grad.detach_()
                p.grad.requires_grad_(False)
                p.grad.zero_()

    def _gather_grads(self, flat_g):
        """The kernel reads the model's flat gradient buffer.  After a plain backward every p.grad IS a view
        of it; after gradient accumulation (or anything else that made autograd allocate its own p.grad) the
        gradients are copied into their slots first.  Returns the names of parameters WITHOUT a gradient (frozen
        with requires_grad=False, or unused): torch.optim.Adam skips those (train.py:55,80), so step() leaves their
        values and moments untouched."""
        _, _, offsets = self.model.flat_buffers()
        named = list(self.model.named_parameters())
        missing = [n for n, p in named if p.grad is None]
        if len(missing) == len(named):
            raise RuntimeError("FusedAdam.step: no parameter has a gradient: run backward first")
        base = flat_g.data_ptr()
        for n, p in named:
            o, k = offsets[n]
            if p.grad is None:
                flat_g[o:o + k].zero_()
            elif p.grad.data_ptr() != base + 4 * o or not p.grad.is_contiguous():
                flat_g[o:o + k].view(p.shape).copy_(p.grad)
        return missing

    def step(self, grad_scale: float = 1.0):
        flat_p, flat_g = self._state()
        missing = self._gather_grads(flat_g)
        g = self.param_groups[0]
        self.step_count += 1
        keep = []
        if missing:      # skipped parameters keep value AND moments (the kernel updates the whole buffer)
            _, _, offsets = self.model.flat_buffers()
            for n in missing:
                o, k = offsets[n]
                keep.append((o, k, flat_p[o:o + k].clone(), self.exp_avg[o:o + k].clone(), self.exp_avg_sq[o:o + k].clone()))
        ops.adam(flat_p, flat_g, self.exp_avg, self.exp_avg_sq, g["lr"], self.step_count, g["betas"][0],
                 g["betas"][1], g["eps"], grad_scale)
        for o, k, pv, m1, m2 in keep:
            flat_p[o:o + k].copy_(pv)
            self.exp_avg[o:o + k].copy_(m1)
            self.exp_avg_sq[o:o + k].copy_(m2)

    # checkpoint format of torch.optim.Adam, so `optimizer_state` in model.pth interchanges
    # (utils/train_logger.py:95-112, train.py:56-57)
    def state_dict(self):
        flat_p, _ = self._state()
        _, _, offsets = self.model.flat_buffers()
        state = {}
        names = [n for n, _ in self.model.named_parameters()]
        for i, n in enumerate(names):
            o, k = offsets[n]
            shape = dict(self.model.named_parameters())[n].shape
            state[i] = {"step": torch.tensor(float(self.step_count)),
                        "exp_avg": self.exp_avg[o:o + k].view(shape).clone(),
                        "exp_avg_sq": self.exp_avg_sq[o:o + k].view(shape).clone()}
        g = self.param_groups[0]
        group = {"lr": g["lr"], "betas": g["betas"], "eps": g["eps"], "weight_decay": 0, "amsgrad": False,
                 "params": list(range(len(names)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        self._state()
        _, _, offsets = self.model.flat_buffers()
        names = [n for n, _ in self.model.named_parameters()]
        for i, n in enumerate(names):
            if i not in sd["state"]:
                continue
            o, k = offsets[n]
            st = sd["state"][i]
            self.exp_avg[o:o + k].copy_(st["exp_avg"].reshape(-1))
            self.exp_avg_sq[o:o + k].copy_(st["exp_avg_sq"].reshape(-1))
            self.step_count = int(float(st["step"]))
        g = sd["param_groups"][0]
        self.param_groups[0].update(lr=g["lr"], betas=tuple(g["betas"]), eps=g["eps"])
