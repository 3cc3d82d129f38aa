"""CPU: the oracle (oracle/vqa_oracle.py) against the golden vectors the reference produced."""
import pytest
import torch

from oracle import vqa_oracle as O
from tests.golden_util import TINY_CASES, TRAIN_CASES, Golden, tiny_cfg


@pytest.mark.parametrize("name", TINY_CASES)
def test_forward_stages_and_logits(name):
    g = Golden(name)
    cfg = tiny_cfg(g.meta)
    st = {}
    y = O.vqa_forward(g.sd, cfg, g.t["v"], g.t["q"], g.t["q_len"], st)
    for i in range(3):
        torch.testing.assert_close(st[f"pool{i}"], g.stage[f"pool{i}"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(st["question"], g.stage["question"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(st["attention"], g.stage["attention"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(y, g.t["logits"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("name", TINY_CASES)
def test_loss_score_grads(name):
    g = Golden(name)
    cfg = tiny_cfg(g.meta)
    y, loss, grads = O.loss_and_grads(g.sd, cfg, g.t["v"], g.t["q"], g.t["q_len"],
                                      g.t["a_idx"], g.t["a_val"])
    torch.testing.assert_close(loss, g.t["loss"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(O.batch_accuracy(y, g.t["a_idx"], g.t["a_val"]), g.t["score"])
    assert set(grads) == set(g.grad)
    for k in grads:
        torch.testing.assert_close(grads[k], g.grad[k], rtol=2e-4, atol=2e-6, msg=lambda m: f"{k}: {m}")
    # padding / unknown-token row of the embedding gets no gradient
    assert float(grads["text.embedding.weight"][0].abs().max()) == 0.0


@pytest.mark.parametrize("name", TRAIN_CASES)
def test_train_mode_with_recorded_masks(name):
    """Dropout placement: the reference model ran in train mode with every nn.Dropout fed a recorded mask
    (tests/golden/make_golden.py: train_case); the oracle given the same masks must reproduce its logits, loss
    and every gradient -- and must NOT when a mask is withheld (the fixture really exercises every site)."""
    g = Golden(name)
    cfg = tiny_cfg(g.meta)
    assert set(g.mask) == set(O.MASK_SITES)
    y, loss, grads = O.loss_and_grads(g.sd, cfg, g.t["v"], g.t["q"], g.t["q_len"], g.t["a_idx"], g.t["a_val"],
                                      masks=g.mask)
    torch.testing.assert_close(y, g.t["logits"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(loss, g.t["loss"], rtol=1e-5, atol=1e-6)
    for k in grads:
        torch.testing.assert_close(grads[k], g.grad[k], rtol=2e-4, atol=2e-6, msg=lambda m: f"{k}: {m}")
    for site in O.MASK_SITES:
        partial = {k: v for k, v in g.mask.items() if k != site}
        y2 = O.vqa_forward(g.sd, cfg, g.t["v"], g.t["q"], g.t["q_len"], masks=partial)
        assert float((y2 - g.t["logits"]).abs().max()) > 1e-4, f"site {site} has no effect on the logits"


def test_float64_oracle_close_to_fp32_reference():
    g = Golden("tiny_plus")
    sd64 = {k: v.double() for k, v in g.sd.items()}
    y = O.vqa_forward(sd64, tiny_cfg(g.meta), g.t["v"].double(), g.t["q"], g.t["q_len"])
    assert float((y.float() - g.t["logits"]).abs().max()) < 1e-5


def test_adam_and_lr_against_torch():
    torch.manual_seed(0)
    p = torch.randn(257)
    ref = p.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=5e-4)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    for step in range(1, 4):
        gr = torch.randn(257)
        lr = O.learning_rate(5e-4, step - 1)
        for gp in opt.param_groups:
            gp["lr"] = lr
        ref.grad = gr.clone()
        opt.step()
        O.adam_step(p, gr, m, v, step, lr)
    torch.testing.assert_close(p, ref.detach(), rtol=1e-6, atol=1e-7)
    assert abs(O.learning_rate(1.0, 50000) - 0.5) < 1e-12
