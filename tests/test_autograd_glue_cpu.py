"""CPU: the autograd / gradient-buffer / data-parallel glue of dl_vqa_amd.model._VqaFunction.

No kernel runs here: the module's Engine is replaced by a stand-in that computes the same forward and
backward with the CPU oracle (test infrastructure), so what is under test is the host logic only --
which gradient buffer a backward writes into, what autograd then accumulates, and when the
data-parallel all-reduce happens:
  * two forwards inside one autograd graph (the second backward used to overwrite the first's
    gradients in the shared flat buffer and autograd added two aliases of the same memory);
  * gradient accumulation / zero_grad(set_to_none=False) under data parallelism (the
    all-reduce used to be skipped silently and the replicas diverged).
"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.golden_util import tiny_cfg

GROUPS = ("classifier", "attention", "text", "image")


class OracleEngine:
    """Engine stand-in: forward / backward of the hot path by the CPU oracle (eval mode)."""

    def __init__(self, cfg):
        self.cfg = cfg

    def forward(self, P, v, q, q_len, training, seed, keep, bad_tokens=None):
        from oracle import vqa_oracle as O
        params = {k: t.detach().clone().requires_grad_(True) for k, t in P.items()}
        with torch.enable_grad():
            logits = O.vqa_forward(params, self.cfg, v, q, q_len)
        return logits.detach(), (params, logits)

    def backward(self, P, saved, dlogits, Gr, on_ready=None):
        params, logits = saved
        grads = torch.autograd.grad(logits, list(params.values()), dlogits, allow_unused=True)
        for (k, p), g in zip(params.items(), grads):
            Gr[k].copy_(torch.zeros_like(p) if g is None else g)
        for group in GROUPS:
            if on_ready is not None:
                on_ready(group)


def make_model(seed=3):
    from dl_vqa_amd import VqaNet
    from dl_vqa_amd.model import _VqaFunction

    class CpuVqaNet(VqaNet):          # test-only: flat buffers on the CPU, oracle engine
        def _ensure_flat(self):
            if self._flat_param is None:
                self._flatten(torch.device("cpu"))

        def forward(self, v, q, q_len):
            self._ensure_flat()
            return _VqaFunction.apply(self, v, q, q_len, 0, *self._params)

    torch.manual_seed(seed)
    cfg = tiny_cfg(dict(bidirectional=True, stride=1, do_option="+"))
    m = CpuVqaNet(cfg, 40).eval()
    m._engine = OracleEngine(cfg)
    return m, cfg


def oracle_grads(model, cfg, batch, divisor):
    from oracle import vqa_oracle as O
    v, q, a_idx, a_val, _, _, q_len = batch
    sd = {k: p.data.clone() for k, p in model.named_parameters()}
    return O.loss_and_grads(sd, cfg, v, q, q_len, a_idx, a_val, loss_scale_batch=divisor)[2]


def loss_of(model, batch, divisor):
    from oracle import vqa_oracle as O
    v, q, a_idx, a_val, _, _, q_len = batch
    y = model(v, q, q_len)
    return O.soft_ce_loss(y, a_idx, a_val) * (v.shape[0] / float(divisor))


def close(got, ref, name):
    scale = max(float(ref.abs().max()), 1e-12)
    assert float((got - ref).abs().max()) / scale < 1e-5 or name.endswith("attention.x_conv.bias"), name


def test_single_backward_writes_the_models_flat_buffer():
    from oracle import vqa_oracle as O
    m, cfg = make_model()
    b = O.synthetic_batch(4, 32, 5, 40, 12, seed=1)
    loss_of(m, b, 4).backward()
    base = m._flat_grad.data_ptr()
    ref = oracle_grads(m, cfg, b, 4)
    for n, p in m.named_parameters():
        assert p.grad.data_ptr() == base + 4 * m._offsets[n][0], n     # p.grad aliases the flat buffer
        close(p.grad, ref[n], n)
    assert len(m._pending) == 0


def test_two_forwards_in_one_graph_accumulate_correctly():
    from oracle import vqa_oracle as O
    m, cfg = make_model()
    b1 = O.synthetic_batch(3, 32, 5, 40, 12, seed=1)
    b2 = O.synthetic_batch(3, 32, 5, 40, 12, seed=2)
    (loss_of(m, b1, 3) + loss_of(m, b2, 3)).backward()
    r1, r2 = oracle_grads(m, cfg, b1, 3), oracle_grads(m, cfg, b2, 3)
    for n, p in m.named_parameters():
        close(p.grad, r1[n] + r2[n], n)
    assert len(m._pending) == 0


def test_accumulation_over_two_backwards_and_a_dropped_graph():
    from oracle import vqa_oracle as O
    m, cfg = make_model()
    b1 = O.synthetic_batch(3, 32, 5, 40, 12, seed=1)
    b2 = O.synthetic_batch(3, 32, 5, 40, 12, seed=2)
    dropped = loss_of(m, b1, 3)          # a forward whose graph is thrown away must not block the direct path for ever
    del dropped
    loss_of(m, b1, 3).backward()
    loss_of(m, b2, 3).backward()         # p.grad is set: accumulates
    r1, r2 = oracle_grads(m, cfg, b1, 3), oracle_grads(m, cfg, b2, 3)
    for n, p in m.named_parameters():
        close(p.grad, r1[n] + r2[n], n)
    # FusedAdam's view of the gradients: after accumulation p.grad may live outside the flat buffer
    from dl_vqa_amd.train import FusedAdam
    opt = FusedAdam(m, lr=1e-3)
    opt._gather_grads(m._flat_grad)
    for n, p in m.named_parameters():
        o, k = m._offsets[n]
        assert torch.equal(m._flat_grad[o:o + k].view(p.shape), p.grad), n
    for p in m.parameters():
        p.grad = None
    with pytest.raises(RuntimeError, match="without a gradient"):
        opt._gather_grads(m._flat_grad)


def test_second_backward_through_one_forward_is_refused():
    from oracle import vqa_oracle as O
    m, _ = make_model()
    b = O.synthetic_batch(2, 32, 5, 40, 12, seed=1)
    loss = loss_of(m, b, 2)
    loss.backward(retain_graph=True)
    with pytest.raises(RuntimeError, match="twice"):
        loss.backward()


# ------------------------------------------------------------------ data parallel, world_size 2 (gloo)
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _dp_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dl_vqa_amd.distributed import DataParallel, shard_batch
        from oracle import vqa_oracle as O
        m, cfg = make_model(seed=20 + rank)            # different weights per rank: the wrapper broadcasts rank 0's
        DataParallel(m)
        out = {}
        # (a) plain step: buckets reduced from inside backward, p.grad aliases the flat buffer
        g1 = O.synthetic_batch(4, 32, 5, 40, 12, seed=5)
        loss_of(m, shard_batch(g1, rank, world), 4).backward()
        out["plain"] = {n: p.grad.clone() for n, p in m.named_parameters()}
        # (b) second micro-step WITHOUT dropping the gradients (accumulation): must still be reduced
        g2 = O.synthetic_batch(4, 32, 5, 40, 12, seed=6)
        loss_of(m, shard_batch(g2, rank, world), 4).backward()
        out["accum"] = {n: p.grad.clone() for n, p in m.named_parameters()}
        # (c) zero_grad(set_to_none=False) semantics: gradients zeroed in place, not dropped
        for p in m.parameters():
            p.grad.zero_()
        loss_of(m, shard_batch(g2, rank, world), 4).backward()
        out["zeroed"] = {n: p.grad.clone() for n, p in m.named_parameters()}
        torch.save(out, os.path.join(out_dir, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_dp_gradients_are_reduced_with_and_without_accumulation(tmp_path):
    from oracle import vqa_oracle as O
    world, port = 2, _free_port()
    mp.spawn(_dp_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = torch.load(tmp_path / "rank0.pt"), torch.load(tmp_path / "rank1.pt")
    m, cfg = make_model(seed=20)
    g1 = O.synthetic_batch(4, 32, 5, 40, 12, seed=5)
    g2 = O.synthetic_batch(4, 32, 5, 40, 12, seed=6)
    ref1, ref2 = oracle_grads(m, cfg, g1, 4), oracle_grads(m, cfg, g2, 4)
    for n in ref1:
        for key, ref in (("plain", ref1[n]), ("accum", ref1[n] + ref2[n]), ("zeroed", ref2[n])):
            assert torch.equal(r0[key][n], r1[key][n]), (key, n)          # replicas agree bit for bit
            close(r0[key][n], ref, f"{key}:{n}")
