"""Host-side mirror of the reference training procedure (train.py:18-208, utils/train_utils.py).

Same function names and argument meaning as the reference so its loop is drop-in:
``run_batch(model, log_softmax, batch_data, max_answers)``, ``evaluate``, ``train``,
``update_learning_rate``, ``TrainParams``, ``batch_accuracy``.  What differs is where the work
runs: the soft-target cross entropy, its gradient and the VQA score are one HIP kernel on the
device (the reference builds numpy index arrays on the host and syncs B+1 times per step,
train.py:195-199, train_utils.py:19-23), and Adam is one fused kernel over the flat parameter
buffer (train.py:55,80).
"""
from __future__ import annotations

import time
from typing import Dict, Optional

import torch
import torch.nn as nn

from . import ops


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
    kernel computes log-softmax, the sparse soft-target cross entropy and the score in one pass."""
    v, q, a_indices, a_values, a_length, idx, q_len = batch_data
    dev = next(model.parameters()).device
    v = v.to(dev, non_blocking=True)
    q = q.to(dev, non_blocking=True)
    a_indices = a_indices.to(dev, non_blocking=True)
    a_values = a_values.to(dev, non_blocking=True)
    q_len = q_len.to(dev, non_blocking=True)
    # the dataset stores fp16 features (data_preprocessing.py:174 casts them on the host, per sample): the fp16
    # batch goes over PCIe as is and VqaNet.forward widens it on the device (vqa_half_to_float)
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
        for p in self.model.parameters():
            p.grad = None

    def step(self, grad_scale: float = 1.0):
        flat_p, flat_g = self._state()
        g = self.param_groups[0]
        self.step_count += 1
        ops.adam(flat_p, flat_g, self.exp_avg, self.exp_avg_sq, g["lr"], self.step_count, g["betas"][0],
                 g["betas"][1], g["eps"], grad_scale)

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


# ------------------------------------------------------------------ train / evaluate
class TrainParams:
    """utils/train_utils.py:58-80."""

    def __init__(self, **kwargs):
        self.n_epochs_stop = kwargs["n_epochs_stop"]
        self.num_epochs = kwargs["num_epochs"]
        self.lr = kwargs["lr"]["lr_value"]
        self.lr_decay = kwargs["lr"]["lr_decay"]
        self.lr_gamma = kwargs["lr"]["lr_gamma"]
        self.lr_step_size = kwargs["lr"]["lr_step_size"]
        self.save_model = kwargs["save_model"]
        self.max_answers = kwargs["max_answers"]


def get_train_params(cfg) -> TrainParams:
    return TrainParams(**cfg["train"])


def get_zeroed_metrics_dict() -> Dict:
    return {"train_loss": 0, "train_score": 0, "total_norm": 0, "count_norm": 0}


def get_metrics(best_eval_score, eval_score, train_loss):
    return {"Metrics/BestAccuracy": best_eval_score, "Metrics/LastAccuracy": eval_score,
            "Metrics/LastLoss": train_loss}


class _NullLogger:
    def write(self, *a, **k): pass
    def write_epoch_statistics(self, **k): print(k)
    def report_scalars(self, *a, **k): pass
    def report_scalars_same_plot(self, *a, **k): pass
    def save_model(self, *a, **k): pass


def train(model: nn.Module, train_loader, eval_loader, train_params: TrainParams, logger=None,
          optimizer_stuff: Optional[dict] = None, world_size: int = 1):
    """Training procedure with the control flow of the reference (train.py:38-141)."""
    logger = logger if logger is not None else _NullLogger()
    total_iterations = 0
    best_eval_score = torch.tensor(0.0)
    epochs_no_improve = 0
    optimizer = FusedAdam(model, lr=train_params.lr)
    if optimizer_stuff:
        optimizer.load_state_dict(optimizer_stuff)
    metrics = get_zeroed_metrics_dict()
    for epoch in range(train_params.num_epochs):
        t = time.time()
        metrics = get_zeroed_metrics_dict()
        for batch_data in train_loader:
            divisor = batch_data[0].shape[0] * world_size
            batch_loss, batch_score = run_batch(model, None, batch_data, train_params.max_answers, divisor)
            optimizer.zero_grad()
            update_learning_rate(optimizer=optimizer, iteration=total_iterations, initial_lr=train_params.lr)
            batch_loss.backward()
            optimizer.step()
            total_iterations += 1
            metrics["train_score"] += batch_score.detach()
            metrics["train_loss"] += batch_loss.detach()
        metrics["train_loss"] /= len(train_loader)
        metrics["train_score"] /= len(train_loader.dataset)
        metrics["train_score"] *= 100
        model.train(False)
        metrics["eval_score"], metrics["eval_loss"] = evaluate(model, eval_loader, train_params.max_answers)
        model.train(True)
        epoch_time = time.time() - t
        logger.write_epoch_statistics(epoch=epoch, epoch_time=epoch_time, train_loss=metrics["train_loss"], norm=0,
                                      train_score=metrics["train_score"], eval_score=metrics["eval_score"])
        logger.report_scalars({"Accuracy/Train": metrics["train_score"], "Accuracy/Validation": metrics["eval_score"],
                               "Loss/Train": metrics["train_loss"], "Loss/Validation": metrics["eval_loss"]}, epoch)
        if metrics["eval_score"] > best_eval_score:
            epochs_no_improve = 0
            best_eval_score = metrics["eval_score"]
            if train_params.save_model:
                logger.save_model(model, epoch, optimizer)
        else:
            epochs_no_improve += 1
        if epoch > 3 and epochs_no_improve == train_params.n_epochs_stop:
            logger.write("Early stopping!")
            break
    return get_metrics(best_eval_score, metrics["eval_score"], metrics["train_loss"])


@torch.no_grad()
def evaluate(model: nn.Module, dataloader, max_answers):
    """train.py:144-169: (accuracy in percent, mean loss) over a loader, no gradients."""
    score = torch.tensor(0.0)
    loss = 0
    for batch_data in dataloader:
        batch_loss, batch_score = run_batch(model, None, batch_data, max_answers)
        loss += batch_loss
        score = score.to(batch_score.device) + batch_score
    loss /= len(dataloader)
    score /= len(dataloader.dataset)
    score *= 100
    return score, loss
