"""Generate the golden fixtures under tests/golden/ from the REFERENCE model.

Run in the build container only (needs /root/reference; it does not exist on the
GPU box and nothing at test time imports this script's dependencies):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

What is recorded (data only: inputs and expected outputs, no reference source):
  tiny_{plus,mul,cat,stride2,uni,k5,k2_stride2,k1}.npz
      inputs, the full state_dict, per-stage activations, logits, loss, VQA score
      and the gradient of every parameter, all produced by the imported
      ``models.model.VqaNet`` in eval mode (dropout = identity) with autograd.
  tiny_{plus,mul,cat}_train.npz
      the same model in TRAIN mode with the dropout masks as recorded data: each nn.Dropout instance of the
      reference model is swapped for a module that multiplies by the next recorded keep-scale mask
      (Bernoulli(1-p)/(1-p), p = 0.3), so WHERE each mask is applied is decided by the reference's own
      forward (models/model.py:84,156,185,186,194,201,204).  Fixture: masks per site, logits, loss, gradients.
  full224_seed1.npz
      the north-star architecture (channels [3,64,128,256], E=300, H=1024, mid=1024,
      G=2, A=1000, V=5000) at S=224, B=2, T=14.  The 18.7 M parameters are not
      stored: they are re-created on the test side by seeding torch and constructing
      the same torch.nn layers in the same order (torch 2.10.0 CPU generator), so
      only the seeds, logits, loss and per-parameter gradient checksums are kept.

The loss is not importable from the reference (train.py needs tqdm/omegaconf and
a GPU), so it is replayed here from the statements of train.py:190-206 using the
reference model's logits; utils/train_utils.py:12-25 likewise for the score.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, "/root/reference")
from models.model import VqaNet  # noqa: E402  (reference, build container only)

HERE = os.path.dirname(os.path.abspath(__file__))


def tiny_cfg(do_option="+", stride=1, bidirectional=True, kernel_size=3):
    return {
        "text": {"question_features": 16, "embedding_features": 12, "dropout": 0.3,
                 "num_lstm_layers": 1, "bidirectional": bidirectional},
        "image": {"kernel_size": kernel_size, "dropout": 0.3, "num_channels": [3, 8, 16, 32],
                  "stride": stride, "do_skip_connection": False},
        "attention": {"hidden_dim": 24, "glimpses": 2, "do_option": do_option, "dropout": 0.3},
        "classifier": {"hidden_dim": 20, "dropout": 0.3},
        "max_answers": 12,
    }


def full_cfg():
    return {
        "text": {"question_features": 1024, "embedding_features": 300, "dropout": 0.3,
                 "num_lstm_layers": 1, "bidirectional": True},
        "image": {"kernel_size": 3, "dropout": 0.3, "num_channels": [3, 64, 128, 256],
                  "stride": 1, "do_skip_connection": False},
        "attention": {"hidden_dim": 1024, "glimpses": 2, "do_option": "+", "dropout": 0.3},
        "classifier": {"hidden_dim": 1024, "dropout": 0.3},
        "max_answers": 1000,
    }


def replay_loss(y_hat, a_indices, a_values, a_length):
    """train.py:190-206 replayed on the reference model's logits."""
    nll = -torch.log_softmax(y_hat, dim=1)
    B = y_hat.shape[0]
    rows = np.repeat(np.arange(B), a_length.numpy())
    flat_i = a_indices.flatten()
    cols = flat_i[flat_i != 0].numpy() - 1
    flat_v = a_values.flatten()
    w = flat_v[flat_v != 0].to(y_hat.dtype) / 10.0
    return (nll[rows, cols] * w).sum() / B


def replay_score(y_hat, a_indices, a_values):
    """utils/train_utils.py:12-25 replayed: min(1, 0.3*count) of the arg-max answer, summed."""
    pred = y_hat.argmax(dim=1)
    tot = 0.0
    for b in range(y_hat.shape[0]):
        agree = 0
        for k in range(a_indices.shape[1]):
            if a_indices[b, k] != 0 and a_indices[b, k] - 1 == pred[b]:
                agree = int(a_values[b, k])
        tot += min(1.0, 0.3 * agree)
    return torch.tensor(tot)


def make_inputs(B, S, T, V, A, q_len, seed):
    g = torch.Generator().manual_seed(seed)
    v = torch.randn(B, 3, S, S, generator=g)
    q = torch.randint(1, V, (B, T), generator=g)
    q_len = torch.tensor(q_len, dtype=torch.int64)
    q = q * (torch.arange(T)[None, :] < q_len[:, None])
    kmax = 3
    a_len = torch.randint(1, kmax + 1, (B,), generator=g)
    a_idx = torch.zeros(B, kmax, dtype=torch.int64)
    a_val = torch.zeros(B, kmax, dtype=torch.int64)
    for b in range(B):
        k = int(a_len[b])
        a_idx[b, :k] = torch.randperm(A, generator=g)[:k] + 1
        a_val[b, :k] = torch.randint(1, 5, (k,), generator=g)
    return v, q, q_len, a_idx, a_val, a_len


def run_reference(model, v, q, q_len, a_idx, a_val, a_len, capture=True, train=False):
    model.train(train)
    stages = {}
    hooks = []
    if capture:
        def grab(name):
            def fn(_m, _i, o):
                stages[name] = o.detach().clone()
            return fn
        for i in range(3):
            hooks.append(getattr(model.image, f"maxpool{i}").register_forward_hook(grab(f"pool{i}")))
        hooks.append(model.text.register_forward_hook(grab("question")))
        hooks.append(model.attention.register_forward_hook(grab("attention")))
    y = model(v, q, q_len)
    for h in hooks:
        h.remove()
    loss = replay_loss(y, a_idx, a_val, a_len)
    model.zero_grad()
    loss.backward()
    score = replay_score(y.detach(), a_idx, a_val)
    grads = {k: (p.grad.detach().clone() if p.grad is not None else torch.zeros_like(p))
             for k, p in model.named_parameters()}
    return y.detach(), loss.detach(), score, grads, stages


def tiny_case(name, do_option="+", stride=1, bidirectional=True, S=32, seed=1, kernel_size=3):
    cfg = tiny_cfg(do_option, stride, bidirectional, kernel_size)
    V, A, B, T = 50, cfg["max_answers"], 3, 5
    torch.manual_seed(seed)
    model = VqaNet(cfg, V)
    # make padding / unknown token appear inside a question too (row 0 must get zero grad)
    v, q, q_len, a_idx, a_val, a_len = make_inputs(B, S, T, V, A, [5, 3, 1], seed + 100)
    q[0, 2] = 0
    y, loss, score, grads, stages = run_reference(model, v, q, q_len, a_idx, a_val, a_len)
    out = {"v": v, "q": q, "q_len": q_len, "a_idx": a_idx, "a_val": a_val, "a_len": a_len,
           "logits": y, "loss": loss, "score": score}
    for k, t in model.state_dict().items():
        out["sd/" + k] = t
    for k, t in grads.items():
        out["grad/" + k] = t
    for k, t in stages.items():
        out["stage/" + k] = t
    meta = dict(do_option=do_option, stride=stride, bidirectional=bidirectional, S=S, V=V, seed=seed)
    if kernel_size != 3:
        meta["kernel_size"] = kernel_size
    np.savez_compressed(os.path.join(HERE, name + ".npz"),
                        **{k: t.numpy() for k, t in out.items()},
                        meta=np.array(repr(meta)))
    print(name, "loss", float(loss), "score", float(score), "logits[0,:3]", y[0, :3].tolist())


def kernel_size_cases():
    """image.kernel_size != 3 (utils/config_schema.py:59 admits any int; models/model.py:75-80): 5x5, 2x2 with stride 2, 1x1."""
    tiny_case("tiny_k5", "+", S=64, seed=5, kernel_size=5)
    tiny_case("tiny_k2_stride2", "*", stride=2, S=128, seed=6, kernel_size=2)
    tiny_case("tiny_k1", "|", S=32, seed=7, kernel_size=1)


class MaskDrop(torch.nn.Module):
    """Stand-in for one nn.Dropout INSTANCE of the reference model: call k multiplies by the k-th recorded mask."""

    def __init__(self, masks):
        super().__init__()
        self.masks, self.calls = masks, 0

    def forward(self, x):
        m = self.masks[self.calls]
        self.calls += 1
        assert m.shape == x.shape, (m.shape, x.shape)
        return x * m


def train_case(name, do_option="+", S=32, seed=1, p=0.3):
    cfg = tiny_cfg(do_option, 1, True)
    V, A, B, T = 50, cfg["max_answers"], 3, 5
    torch.manual_seed(seed)
    model = VqaNet(cfg, V)
    v, q, q_len, a_idx, a_val, a_len = make_inputs(B, S, T, V, A, [5, 3, 1], seed + 100)
    q[0, 2] = 0
    ch, E, H = cfg["image"]["num_channels"], cfg["text"]["embedding_features"], cfg["text"]["question_features"]
    mid, G, hid = cfg["attention"]["hidden_dim"], cfg["attention"]["glimpses"], cfg["classifier"]["hidden_dim"]
    g = S
    for _ in range(3):
        g = (g - 2) // 2
    C, Q = ch[-1], 2 * H
    gen = torch.Generator().manual_seed(seed + 500)
    mk = lambda *shape: (torch.rand(*shape, generator=gen) >= p).float() / (1.0 - p)
    masks = {"image": mk(B, C, g, g), "text": mk(B, T, E), "att_v": mk(B, C, g, g), "att_q": mk(B, Q),
             "att_x": mk(B, 2 * mid if do_option == "|" else mid, g, g), "cls1": mk(B, G * C + Q), "cls2": mk(B, hid)}
    # one stand-in per nn.Dropout instance; attention.drop is called three times: on v, on q, on x (model.py:185,186,194)
    model.image.drop = MaskDrop([masks["image"]])
    model.text.drop = MaskDrop([masks["text"]])
    model.attention.drop = MaskDrop([masks["att_v"], masks["att_q"], masks["att_x"]])
    model.classifier.drop1 = MaskDrop([masks["cls1"]])
    model.classifier.drop2 = MaskDrop([masks["cls2"]])
    y, loss, score, grads, _ = run_reference(model, v, q, q_len, a_idx, a_val, a_len, capture=False, train=True)
    assert model.attention.drop.calls == 3 and model.image.drop.calls == 1 and model.classifier.drop2.calls == 1
    out = {"v": v, "q": q, "q_len": q_len, "a_idx": a_idx, "a_val": a_val, "a_len": a_len,
           "logits": y, "loss": loss, "score": score}
    for k, t in model.state_dict().items():
        out["sd/" + k] = t
    for k, t in grads.items():
        out["grad/" + k] = t
    for k, t in masks.items():
        out["mask/" + k] = t
    meta = dict(do_option=do_option, stride=1, bidirectional=True, S=S, V=V, seed=seed, p=p)
    np.savez_compressed(os.path.join(HERE, name + ".npz"),
                        **{k: t.numpy() for k, t in out.items()},
                        meta=np.array(repr(meta)))
    print(name, "loss", float(loss), "score", float(score), "logits[0,:3]", y[0, :3].tolist())


def full_case(seed=1):
    cfg = full_cfg()
    V, A, B, T, S = 5000, 1000, 2, 14, 224
    torch.manual_seed(seed)
    model = VqaNet(cfg, V)
    v, q, q_len, a_idx, a_val, a_len = make_inputs(B, S, T, V, A, [14, 6], seed + 100)
    y, loss, score, grads, stages = run_reference(model, v, q, q_len, a_idx, a_val, a_len)
    out = {"logits": y.numpy(), "loss": loss.numpy(), "score": score.numpy(),
           "question": stages["question"].numpy(), "attention": stages["attention"].numpy(),
           "pool2_sample": stages["pool2"][:, ::16, ::5, ::5].numpy()}
    names = list(grads.keys())
    out["grad_names"] = np.array(names)
    out["grad_sum"] = np.array([float(grads[k].double().sum()) for k in names])
    out["grad_abs"] = np.array([float(grads[k].double().abs().sum()) for k in names])
    out["grad_l2"] = np.array([float(grads[k].double().pow(2).sum().sqrt()) for k in names])
    # a strided sample of each gradient for element-wise comparison
    for k in names:
        flat = grads[k].flatten()
        step = max(1, flat.numel() // 257)
        out["gsample/" + k] = flat[::step][:257].numpy()
    # parameter checksums so the test can prove it re-created the same parameters
    sd = model.state_dict()
    out["param_names"] = np.array(list(sd.keys()))
    out["param_sum"] = np.array([float(t.double().sum()) for t in sd.values()])
    out["param_abs"] = np.array([float(t.double().abs().sum()) for t in sd.values()])
    meta = dict(seed=seed, input_seed=seed + 100, B=B, S=S, T=T, V=V, A=A, q_len=[14, 6])
    np.savez_compressed(os.path.join(HERE, "full224_seed1.npz"), meta=np.array(repr(meta)), **out)
    print("full224 loss", float(loss), "logits[0,:3]", y[0, :3].tolist())


if __name__ == "__main__":
    if "--train-only" in sys.argv:      # add the train-mode fixtures without rewriting the others
        for nm, op in (("tiny_plus_train", "+"), ("tiny_mul_train", "*"), ("tiny_cat_train", "|")):
            train_case(nm, op)
        sys.exit(0)
    if "--kernel-only" in sys.argv:     # add the kernel_size != 3 fixtures without rewriting the others
        kernel_size_cases()
        sys.exit(0)
    tiny_case("tiny_plus", "+")
    tiny_case("tiny_mul", "*")
    tiny_case("tiny_cat", "|")
    tiny_case("tiny_stride2", "+", stride=2, S=96)
    tiny_case("tiny_uni", "+", bidirectional=False)
    # odd intermediate sizes like the north-star shapes (62->31, 29->14, 12->6)
    tiny_case("small64_plus", "+", S=64, seed=3)
    for nm, op in (("tiny_plus_train", "+"), ("tiny_mul_train", "*"), ("tiny_cat_train", "|")):
        train_case(nm, op)
    kernel_size_cases()
    full_case()
