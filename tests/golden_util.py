"""Helpers to load the committed golden fixtures (tests/golden/*.npz)."""
import ast
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

TINY_CASES = ["tiny_plus", "tiny_mul", "tiny_cat", "tiny_stride2", "tiny_uni", "small64_plus",
              # image.kernel_size != 3 (make_golden.kernel_size_cases)
              "tiny_k5", "tiny_k2_stride2", "tiny_k1"]
# reference model in train mode with the dropout masks recorded as data (make_golden.train_case)
TRAIN_CASES = ["tiny_plus_train", "tiny_mul_train", "tiny_cat_train"]


def tiny_cfg(meta):
    return {
        "text": {"question_features": 16, "embedding_features": 12, "dropout": 0.3,
                 "num_lstm_layers": 1, "bidirectional": meta["bidirectional"]},
        "image": {"kernel_size": meta.get("kernel_size", 3), "dropout": 0.3, "num_channels": [3, 8, 16, 32],
                  "stride": meta["stride"], "do_skip_connection": False},
        "attention": {"hidden_dim": 24, "glimpses": 2, "do_option": meta["do_option"], "dropout": 0.3},
        "classifier": {"hidden_dim": 20, "dropout": 0.3},
        "max_answers": 12,
    }


def full_cfg(max_answers=1000):
    return {
        "text": {"question_features": 1024, "embedding_features": 300, "dropout": 0.3,
                 "num_lstm_layers": 1, "bidirectional": True},
        "image": {"kernel_size": 3, "dropout": 0.3, "num_channels": [3, 64, 128, 256],
                  "stride": 1, "do_skip_connection": False},
        "attention": {"hidden_dim": 1024, "glimpses": 2, "do_option": "+", "dropout": 0.3},
        "classifier": {"hidden_dim": 1024, "dropout": 0.3},
        "max_answers": max_answers,
    }


class Golden:
    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
        self.meta = ast.literal_eval(str(z["meta"]))
        self.raw = z
        self.sd, self.grad, self.stage, self.t, self.mask = {}, {}, {}, {}, {}
        for k in z.files:
            if k == "meta" or z[k].dtype.kind in "US":
                continue
            t = torch.from_numpy(z[k])
            if k.startswith("sd/"):
                self.sd[k[3:]] = t
            elif k.startswith("grad/"):
                self.grad[k[5:]] = t
            elif k.startswith("stage/"):
                self.stage[k[6:]] = t
            elif k.startswith("mask/"):
                self.mask[k[5:]] = t
            else:
                self.t[k] = t


def full_inputs(meta):
    """Re-create full224's inputs with the generator calls of make_golden.make_inputs."""
    B, S, T, V, A = meta["B"], meta["S"], meta["T"], meta["V"], meta["A"]
    g = torch.Generator().manual_seed(meta["input_seed"])
    v = torch.randn(B, 3, S, S, generator=g)
    q = torch.randint(1, V, (B, T), generator=g)
    q_len = torch.tensor(meta["q_len"], dtype=torch.int64)
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
