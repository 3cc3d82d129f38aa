"""Drop-in ``VqaNet`` whose forward/backward run on the HIP kernel library.

Mirrors the reference's module interface (models/model.py:7-67): same constructor
``VqaNet(cfg, embedding_tokens)``, same ``forward(v, q, q_len) -> [B, max_answers]``, same
sub-module attribute names (``text``, ``image``, ``attention``, ``classifier`` — read by
utils/main_utils.py:21-41) and the same ``state_dict`` keys/shapes, so ``model.pth`` checkpoints
interchange (utils/train_logger.py:95-112, evaluate_vqa.py:72-75).

The sub-modules are parameter containers built from the same torch constructors in the same order
as the reference, so ``torch.manual_seed(s); VqaNet(cfg, V)`` yields bit-identical initial
weights.  Their parameters live in ONE flat fp32 buffer (ordered as backward produces the
gradients) so that the optimiser and the data-parallel all-reduce work on contiguous ranges.
"""
from __future__ import annotations

import warnings
import weakref
from collections import OrderedDict
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from . import ops
from .engine import Engine


class questionNet(nn.Module):
    """Parameter container for models/model.py:134-166 (Embedding pad 0 -> Dropout -> Tanh -> LSTM -> c_n)."""

    def __init__(self, embedding_tokens, embedding_features, lstm_features, num_lstm_layers, drop, bidirectional):
        super().__init__()
        self.embedding = nn.Embedding(num_embeddings=embedding_tokens, embedding_dim=embedding_features, padding_idx=0)
        self.drop = nn.Dropout(drop)
        self.tanh = nn.Tanh()
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")   # torch warns that dropout is a no-op with one layer (as in the reference)
            self.lstm = nn.LSTM(input_size=embedding_features, hidden_size=lstm_features, num_layers=num_lstm_layers,
                                dropout=drop, bidirectional=bidirectional)


class ImageNet2(nn.Sequential):
    """Parameter container for models/model.py:72-84 ([Conv2d k3, ReLU, MaxPool2d(2,2)] x n + Dropout)."""

    def __init__(self, image_cfg):
        super().__init__()
        ch = image_cfg["num_channels"]
        for i in range(len(ch) - 1):
            self.add_module(f"conv{i}", nn.Conv2d(ch[i], ch[i + 1], kernel_size=image_cfg["kernel_size"],
                                                  stride=image_cfg["stride"]))
            self.add_module(f"relu{i}", nn.ReLU())
            self.add_module(f"maxpool{i}", nn.MaxPool2d(2, 2))
        self.add_module("drop", nn.Dropout(image_cfg["dropout"]))


class Attention(nn.Module):
    """Parameter container for models/model.py:169-195."""

    def __init__(self, v_features, q_features, mid_features, glimpses, do_option, drop=0.0):
        super().__init__()
        self.do_option = do_option
        self.v_conv = nn.Conv2d(v_features, mid_features, kernel_size=1, bias=False)
        self.q_lin = nn.Linear(q_features, mid_features)
        self.x_conv = nn.Conv2d(2 * mid_features if do_option == "|" else mid_features, glimpses, kernel_size=1)
        self.drop = nn.Dropout(drop)
        self.relu = nn.ReLU(inplace=True)


class Classifier(nn.Sequential):
    """Parameter container for models/model.py:198-205."""

    def __init__(self, in_features, mid_features, out_features, drop=0.0):
        super().__init__()
        self.add_module("drop1", nn.Dropout(drop))
        self.add_module("lin1", nn.Linear(in_features, mid_features))
        self.add_module("relu", nn.ReLU())
        self.add_module("drop2", nn.Dropout(drop))
        self.add_module("lin2", nn.Linear(mid_features, out_features))


def _flat_order(names: List[str]) -> List[str]:
    """Parameter order inside the flat buffer = the order backward produces gradients."""
    def key(n):
        if n.startswith("classifier."):
            return (0, 0 if "lin2" in n else 1, n)
        if n.startswith("attention."):
            return (1, 0, n)
        if n.startswith("text.lstm"):
            return (2, 0, n)
        if n.startswith("text."):
            return (2, 1, n)
        if n.startswith("image.conv"):
            return (3, -int(n.split(".")[1][4:]), n)
        return (4, 0, n)
    return sorted(names, key=key)


def validate_question_lengths(q_len, T: int) -> None:
    """pack_padded_sequence (models/model.py:159-162) raises for a length below 1 or above the padded width.  Lengths that
    arrive on the host (the data loader's, before run_batch uploads them) are checked there; device-resident lengths are
    taken as they are -- the recurrence kernels then treat a sample as finished from step q_len[b] on (0: c_n = 0)."""
    if q_len.is_cuda or q_len.numel() == 0:
        return
    lo, hi = int(q_len.min()), int(q_len.max())
    if lo < 1:
        raise RuntimeError(f"question length {lo}: every sample needs a length of at least 1 "
                           "(pack_padded_sequence would raise: models/model.py:159-162)")
    if hi > T:
        raise RuntimeError(f"question length {hi} exceeds the padded question width {T} "
                           "(pack_padded_sequence would raise: models/model.py:159-162)")


class _VqaFunction(torch.autograd.Function):
    """Autograd node of one VqaNet.forward call.

    Gradient storage: the model owns ONE flat gradient buffer (`_flat_grad`) that the fused optimiser
    and the data-parallel buckets work on.  backward writes into it directly -- and starts each
    bucket's all-reduce from inside backward -- only when that is safe: no parameter holds a gradient
    yet and no other forward of this module is still waiting for its backward (two forwards inside
    one graph would otherwise overwrite each other's gradients in place and autograd would then add
    two aliases of the same memory).  In every other case (gradient accumulation,
    ``zero_grad(set_to_none=False)``, several forwards per backward) the gradients go to a fresh
    buffer that autograd accumulates as usual; under data parallelism that buffer is all-reduced
    (through the same four bucket collectives, issued from inside backward in the same order) before it is handed
    to autograd, so replicas never step on unreduced gradients and the collective sequence is rank-invariant.
    """

    @staticmethod
    def forward(ctx, model, v, q, q_len, seed, *params):
        P = model._param_dict()
        logits, saved = model._engine.forward(P, v, q, q_len, model.training, seed, keep=True,
                                              bad_tokens=model._bad_tokens)
        ctx.model = model
        ctx.saved = saved
        model._last_ctx = saved
        model._pending.add(ctx)          # weak: a graph that is dropped without backward leaves the set
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        model = ctx.model
        if ctx.saved is None:
            raise RuntimeError("dl_vqa_amd.VqaNet: backward through the same forward twice (the saved "
                               "activations are released after the first backward; retain_graph is not supported)")
        names = model._names
        others = [c for c in model._pending if c is not ctx]
        direct = not others and all(p.grad is None for p in model._params)
        if not direct and others and not model._warned_stale:
            model._warned_stale = True
            warnings.warn("dl_vqa_amd.VqaNet: backward runs while another grad-enabled forward of this module is still "
                          "waiting for its backward (an output kept for metrics / logging?): gradients go to a fresh "
                          "buffer instead of the model's flat gradient buffer (one extra buffer and, for FusedAdam, one "
                          "copy per step).  Use .detach() or torch.no_grad() for forwards that are never differentiated.")
        flat, Gr = model._grad_buffer(fresh=not direct)
        sync = model._grad_sync
        # data parallel: the SAME four bucket all-reduces, in the same order, from inside backward on either path
        # (a rank on the fresh-buffer path used to issue one whole-buffer collective against the other
        # ranks' four)
        on_ready = (lambda group: sync.bucket_ready(model, group, flat)) if sync is not None else None
        model._last_backward_direct = direct
        model._engine.backward(model._param_dict(), ctx.saved, dlogits, Gr, on_ready)
        if sync is not None:
            sync.finish(model)
        ctx.saved = None
        model._pending.discard(ctx)
        return (None, None, None, None, None) + tuple(Gr[n] for n in names)


class VqaNet(nn.Module):
    """MI355X-native VqaNet (reference: models/model.py:7-67)."""

    def __init__(self, cfg, embedding_tokens, compute_dtype: str = "fp32"):
        """compute_dtype (not in the reference): "fp32" (default; the parity path, exact fp32 MFMA) or "bf16" (BASELINE
        configs[3]: conv blocks 1.. and the v_conv products on bf16 MFMA with fp32 accumulation; parameters, LSTM,
        reductions and Adam stay fp32)."""
        super().__init__()
        text_cfg, image_cfg = cfg["text"], cfg["image"]
        attention_cfg, classifier_cfg = cfg["attention"], cfg["classifier"]
        lstm_out_features = text_cfg["question_features"] * (2 if text_cfg["bidirectional"] else 1)
        glimpses = attention_cfg["glimpses"]
        image_features = image_cfg["num_channels"][-1]
        # same construction order as the reference => same RNG stream => identical initial weights
        self.text = questionNet(embedding_tokens=embedding_tokens,
                                embedding_features=text_cfg["embedding_features"],
                                lstm_features=text_cfg["question_features"], drop=text_cfg["dropout"],
                                num_lstm_layers=text_cfg["num_lstm_layers"],
                                bidirectional=text_cfg["bidirectional"])
        self.image = ImageNet2(image_cfg)
        self.attention = Attention(v_features=image_features, q_features=lstm_out_features,
                                   mid_features=attention_cfg["hidden_dim"], glimpses=glimpses,
                                   do_option=attention_cfg["do_option"], drop=attention_cfg["dropout"])
        self.classifier = Classifier(in_features=glimpses * image_features + lstm_out_features,
                                     mid_features=classifier_cfg["hidden_dim"], out_features=cfg["max_answers"],
                                     drop=classifier_cfg["dropout"])
        self.compute_dtype = compute_dtype
        self._engine = Engine(cfg, embedding_tokens, compute_dtype)
        named = OrderedDict(self.named_parameters())
        self._names: List[str] = list(named.keys())                    # state_dict order (autograd inputs)
        self._params: List[nn.Parameter] = list(named.values())
        self._flat_names = _flat_order(self._names)
        self._flat_param: Optional[torch.Tensor] = None
        self._flat_grad: Optional[torch.Tensor] = None
        self._offsets: Dict[str, tuple] = {}
        self._grad_sync = None          # set by dl_vqa_amd.distributed.DataParallel
        self._last_ctx = None
        self._seed_rank = 0
        self._pending = weakref.WeakSet()   # autograd nodes of forwards whose backward has not run yet
        self._warned_stale = False
        self._last_backward_direct = None   # did the last backward write the model's own flat buffer (bench, tests)
        self._bad_tokens = None             # device counter of out-of-vocabulary token ids (+ pinned host copy, event)
        self._bad_host = None
        self._bad_event = None

    # ------------------------------------------------------------------ flat parameter storage
    def _flatten(self, device):
        named = dict(zip(self._names, self._params))
        off = 0
        self._offsets = {}
        for n in self._flat_names:
            numel = named[n].numel()
            self._offsets[n] = (off, numel)
            off += (numel + 63) // 64 * 64            # 256-byte aligned starts (GEMM operands need 16 B)
        flat = torch.zeros(off, dtype=torch.float32, device=device)
        for n in self._flat_names:
            o, numel = self._offsets[n]
            p = named[n]
            view = flat[o:o + numel].view(p.shape)
            view.copy_(p.data)
            p.data = view
        self._flat_param = flat
        self._flat_grad = torch.zeros_like(flat)

    def _ensure_flat(self):
        dev = self._params[0].device
        if dev.type != "cuda":
            raise RuntimeError("dl_vqa_amd.VqaNet runs only on an MI355X: move the model with .cuda() "
                               "(there is no CPU fallback)")
        ok = self._flat_param is not None and self._flat_param.device == dev
        if ok:
            base = self._flat_param.data_ptr()
            for n, p in zip(self._names, self._params):
                if p.dtype != torch.float32 or p.data_ptr() != base + 4 * self._offsets[n][0]:
                    ok = False
                    break
        if not ok:
            self._flatten(dev)

    def _param_dict(self) -> Dict[str, torch.Tensor]:
        return {n: p.data for n, p in zip(self._names, self._params)}

    def _grad_buffer(self, fresh: bool = False):
        """(flat buffer, {name: view}) -- the model's own flat gradient buffer, or a fresh one of the same layout."""
        flat = torch.zeros_like(self._flat_grad) if fresh else self._flat_grad
        out = {}
        for n, p in zip(self._names, self._params):
            o, numel = self._offsets[n]
            out[n] = flat[o:o + numel].view(p.shape)
        return flat, out

    def _grad_views(self, fresh: bool = False) -> Dict[str, torch.Tensor]:
        return self._grad_buffer(fresh)[1]

    def flat_buffers(self):
        """(flat parameters, flat gradients, {name: (offset, numel)}) — used by the fused optimiser and DP."""
        self._ensure_flat()
        return self._flat_param, self._flat_grad, dict(self._offsets)

    def group_range(self, group: str):
        """Contiguous [lo, hi) range of the flat buffers that holds a backward group's parameters."""
        offs = [self._offsets[n] for n in self._flat_names if n.split(".")[0] == group]
        lo = min(o for o, _ in offs)
        hi = max(o + (k + 63) // 64 * 64 for o, k in offs)
        return lo, hi

    # ------------------------------------------------------------------ forward
    def _next_seed(self) -> int:
        # one draw from torch's CPU generator per training forward: reproducible under torch.manual_seed
        s = int(torch.empty((), dtype=torch.int64).random_().item())
        return (s ^ (self._seed_rank * 0x5851F42D4C957F2D)) & ((1 << 63) - 1)

    # ------------------------------------------------------------------ token-id validation
    # nn.Embedding raises for ids outside [0, V) (models/model.py:155).  Ids that arrive on the host are checked
    # there; ids already on the device are counted by the embedding kernel and the count is read back without
    # blocking: the error surfaces at the next forward (or at check_token_ids(), which synchronises) -- the
    # same deferred reporting a device-side assert has.
    def _validate_tokens(self, q):
        V = self._engine.V
        if not q.is_cuda:
            if q.numel() and (int(q.min()) < 0 or int(q.max()) >= V):
                raise IndexError(f"question token id out of range [0, {V}) (nn.Embedding would raise: models/model.py:155)")
            return
        self._raise_if_bad(block=False)
        dev = q.device
        if self._bad_tokens is None or self._bad_tokens.device != dev:
            self._bad_tokens = torch.zeros(1, dtype=torch.int32, device=dev)
            self._bad_host = torch.zeros(1, dtype=torch.int32).pin_memory()
            self._bad_event = None

    def _after_forward_tokens(self, q):
        if q.is_cuda and self._bad_tokens is not None:
            self._bad_host.copy_(self._bad_tokens, non_blocking=True)
            self._bad_event = torch.cuda.Event()
            self._bad_event.record(torch.cuda.current_stream(q.device))

    def _raise_if_bad(self, block: bool):
        ev = self._bad_event
        if ev is None:
            return
        if block:
            ev.synchronize()
        elif not ev.query():
            return
        self._bad_event = None
        n = int(self._bad_host[0])
        if n:
            self._bad_tokens.zero_()
            raise IndexError(f"{n} question token id(s) out of range [0, {self._engine.V}) in an earlier forward "
                             "(nn.Embedding would raise: models/model.py:155); they were embedded as zeros")

    def check_token_ids(self):
        """Synchronise and raise IndexError if a forward so far saw a token id outside the vocabulary."""
        self._raise_if_bad(block=True)

    def forward(self, v, q, q_len):
        self._ensure_flat()
        if not v.is_cuda:
            raise RuntimeError("dl_vqa_amd.VqaNet.forward needs CUDA (HIP) tensors; there is no CPU fallback")
        self._validate_tokens(q)
        validate_question_lengths(q_len, q.shape[1])
        # v may be the dataset's fp16 storage format (preprocessing/preprocess_images.py:39-53): the first-block kernels read
        # it as it is (engine.py); no widened copy is made
        if v.dtype not in (torch.float32, torch.float16):
            v = v.float()
        seed = self._next_seed() if self.training else 0
        need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self._params)
        if need_grad:
            logits = _VqaFunction.apply(self, v, q, q_len, seed, *self._params)
        else:
            logits, _ = self._engine.forward(self._param_dict(), v, q, q_len, self.training, seed, keep=False,
                                             bad_tokens=self._bad_tokens)
        self._after_forward_tokens(q)
        return logits
