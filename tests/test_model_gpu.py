"""GPU parity of the whole hot path: dl_vqa_amd.VqaNet (HIP) vs the reference's golden vectors and
vs the CPU oracle, forward and backward.  Tolerances: logits 1e-3 absolute (BASELINE.json
north_star), everything else relative to the tensor's magnitude as stated per check."""
import math

import pytest
import torch

from tests.golden_util import TINY_CASES, Golden, full_cfg, full_inputs, tiny_cfg

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

HIP_CASES = TINY_CASES   # every reference fixture: '+', '*', '|', stride 2, unidirectional, 64x64


# softmax over positions is shift invariant, so d loss / d x_conv.bias is identically zero in exact
# arithmetic (the reference's value is rounding noise ~1e-9): compared absolutely, not relatively.
ZERO_GRAD = "attention.x_conv.bias"


def rel(got, ref):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    return float((got - ref).abs().max()) / max(float(ref.abs().max()), 1e-30)


# For do_option '|' the question half of x adds the same constant to every position of a glimpse, so (in
# eval mode) nothing flows back through q_lin either: its gradients are identically zero as well.
ZERO_GRAD_CAT = ("attention.q_lin.weight", "attention.q_lin.bias")


def grad_err(name, got, ref, do_option="+"):
    if name == ZERO_GRAD or (do_option == "|" and name in ZERO_GRAD_CAT):
        return float((got.detach().double().cpu() - ref.double().cpu()).abs().max()) * 1e2   # |err| < 1e-6 passes 1e-4
    return rel(got, ref)


def build(cfg, V, sd=None, compute_dtype="fp32"):
    from dl_vqa_amd import VqaNet
    m = VqaNet(cfg, V, compute_dtype=compute_dtype)
    if sd is not None:
        m.load_state_dict(sd)
    return m.to(DEV)


def nchw(t):
    return t.permute(0, 3, 1, 2)


@pytest.mark.parametrize("name", HIP_CASES)
def test_golden_forward_loss_grads(name):
    from dl_vqa_amd.train import soft_ce_loss_and_score
    g = Golden(name)
    cfg = tiny_cfg(g.meta)
    m = build(cfg, g.meta["V"], g.sd).eval()
    assert list(m.state_dict().keys()) == list(g.sd.keys())
    v, q, ql = g.t["v"].to(DEV), g.t["q"].to(DEV), g.t["q_len"].to(DEV)
    y = m(v, q, ql)
    loss, score = soft_ce_loss_and_score(y, g.t["a_idx"].to(DEV), g.t["a_val"].to(DEV))
    loss.backward()
    torch.cuda.synchronize()
    ctx = m._last_ctx
    for i in range(3):
        e = rel(nchw(ctx.acts[i + 1]), g.stage[f"pool{i}"])
        print(f"[parity] {name} pool{i}: {e:.3e}")
        assert e < 2e-5
    e = rel(ctx.stages["combined"][:, m._engine.GC:], g.stage["question"])
    print(f"[parity] {name} question: {e:.3e}")
    assert e < 2e-5
    B, G = y.shape[0], m._engine.G
    e = rel(ctx.stages["score"].view(B, G, -1), g.stage["attention"].reshape(B, G, -1))
    print(f"[parity] {name} attention: {e:.3e}")
    assert e < 2e-5
    err = float((y.cpu() - g.t["logits"]).abs().max())
    print(f"[parity] {name} logits max abs err: {err:.3e}")
    assert err < 1e-3 and err < 1e-5          # north star 1e-3; fp32 MFMA is far inside it
    assert abs(float(loss) - float(g.t["loss"])) < 1e-5
    assert abs(float(score) - float(g.t["score"])) < 1e-6
    for k, p in m.named_parameters():
        e = grad_err(k, p.grad, g.grad[k], g.meta["do_option"])
        print(f"[parity] {name} grad {k}: {e:.3e}")
        assert e < 2e-4, (k, e)
    assert float(dict(m.named_parameters())["text.embedding.weight"].grad[0].abs().max()) == 0.0


def test_fp16_image_features_are_widened_on_the_device():
    """the dataset stores fp16 image features; VqaNet.forward takes them as they are.  At this tiny shape
    (8 output channels: not a dedicated-first-block shape) vqa_half_to_float widens them on the device: same logits, bit
    for bit, as a host-side .float()."""
    g = Golden("tiny_plus")
    m = build(tiny_cfg(g.meta), g.meta["V"], g.sd).eval()
    v16 = g.t["v"].half()
    q, ql = g.t["q"].to(DEV), g.t["q_len"].to(DEV)
    with torch.no_grad():
        y16 = m(v16.to(DEV), q, ql)
        y32 = m(v16.float().to(DEV), q, ql)
    torch.cuda.synchronize()
    assert torch.equal(y16, y32)
    assert float((y16.cpu() - g.t["logits"]).abs().max()) < 5e-2      # fp16 rounding of the input only


@pytest.mark.parametrize("compute_dtype", ["fp32", "bf16"])
def test_fp16_image_features_read_by_the_first_block_kernels(compute_dtype, monkeypatch):
    """at shapes the dedicated first-block kernels cover (the north-star architecture), the fp16 image
    batch is read by vqa_conv0_relu_pool_fwd / vqa_conv0_wgrad(_bf16) as it is -- no vqa_half_to_float pass -- and, the
    widening being exact, logits and every gradient equal those of a widened fp32 copy bit for bit."""
    from dl_vqa_amd import ops
    from dl_vqa_amd.train import soft_ce_loss_and_score
    from oracle import vqa_oracle as O
    cfg = full_cfg(100)
    torch.manual_seed(5)
    m = build(cfg, 60, compute_dtype=compute_dtype).eval()
    v, q, a_idx, a_val, _, _, ql = O.synthetic_batch(2, 96, 6, 60, 100, seed=4)
    v16 = v.half()
    out = {}
    for name, vin in (("fp32", v16.float()), ("fp16", v16)):
        if name == "fp16":
            def boom(*a, **k):
                raise AssertionError("vqa_half_to_float must not run on the forward path at conv0-supported shapes")
            monkeypatch.setattr(ops, "half_to_float", boom)
        for p in m.parameters():
            p.grad = None
        y = m(vin.to(DEV), q.to(DEV), ql.to(DEV))
        loss, _ = soft_ce_loss_and_score(y, a_idx.to(DEV), a_val.to(DEV))
        loss.backward()
        torch.cuda.synchronize()
        assert m._last_ctx is None or True
        out[name] = (y.detach().clone(), {k: p.grad.clone() for k, p in m.named_parameters()})
    assert torch.equal(out["fp16"][0], out["fp32"][0])
    for k in out["fp32"][1]:
        assert torch.equal(out["fp16"][1][k], out["fp32"][1][k]), k


@pytest.mark.parametrize("compute_dtype", ["fp32", "fp32x3"])
def test_full224_reference_logits_and_gradients(compute_dtype):
    """North-star architecture at S=224, B=2: parameters re-created from the seed, logits and
    gradient checksums compared with what the reference produced (tests/golden/make_golden.py).
    fp32x3: the same tolerances with conv blocks 1-2 on the bf16 matrix cores (exact 3 x bf16 operand split)."""
    from dl_vqa_amd.train import soft_ce_loss_and_score
    import numpy as np
    g = Golden("full224_seed1")
    meta = g.meta
    torch.manual_seed(meta["seed"])
    m = build(full_cfg(meta["A"]), meta["V"], compute_dtype=compute_dtype)
    if compute_dtype == "fp32x3":
        assert m._engine._x3_layer((2, 111, 111, 64), 128) and m._engine._x3_layer((2, 54, 54, 128), 256)
    sd = m.state_dict()
    names = [str(n) for n in g.raw["param_names"]]
    assert list(sd.keys()) == names
    for n, s_ref, a_ref in zip(names, g.raw["param_sum"], g.raw["param_abs"]):
        t = sd[n].double().cpu()
        assert abs(float(t.sum()) - s_ref) <= 1e-9 * max(1.0, a_ref), f"parameter {n} differs from the reference init"
        assert abs(float(t.abs().sum()) - a_ref) <= 1e-9 * max(1.0, a_ref)
    v, q, ql, a_idx, a_val, _ = full_inputs(meta)
    m.eval()
    y = m(v.to(DEV), q.to(DEV), ql.to(DEV))
    loss, score = soft_ce_loss_and_score(y, a_idx.to(DEV), a_val.to(DEV))
    loss.backward()
    torch.cuda.synchronize()
    err = float((y.cpu() - g.t["logits"]).abs().max())
    print(f"[parity] full224 ({compute_dtype}) logits max abs err {err:.3e}; loss {float(loss):.6f} vs {float(g.t['loss']):.6f}")
    assert err < 1e-3
    assert abs(float(loss) - float(g.t["loss"])) < 1e-4
    ctx = m._last_ctx
    assert rel(ctx.stages["combined"][:, m._engine.GC:], g.t["question"]) < 1e-4
    assert rel(ctx.stages["score"].view(2, 2, -1), g.t["attention"].reshape(2, 2, -1)) < 1e-4
    assert rel(nchw(ctx.acts[3])[:, ::16, ::5, ::5], g.t["pool2_sample"]) < 1e-4
    gnames = [str(n) for n in g.raw["grad_names"]]
    grads = dict((k, p.grad) for k, p in m.named_parameters())
    for n, l2 in zip(gnames, g.raw["grad_l2"]):
        if n == ZERO_GRAD:
            assert float(grads[n].abs().max()) < 1e-6
            continue
        flat = grads[n].flatten().double().cpu()
        step = max(1, flat.numel() // 257)
        sample = flat[::step][:257]
        ref = torch.from_numpy(g.raw["gsample/" + n]).double()
        e = float((sample - ref).abs().max()) / max(float(ref.abs().max()), 1e-30)
        e2 = abs(float(flat.pow(2).sum().sqrt()) - l2) / max(l2, 1e-30)
        print(f"[parity] full224 ({compute_dtype}) grad {n}: sample {e:.3e} l2 {e2:.3e}")
        assert e < 2e-3 and e2 < 1e-3, (n, e, e2)


def test_long_questions_3000_answers_match_oracle():
    """BASELINE configs[4] shapes on the text / classifier side (config/config.yaml:74 max_answers = 3000, 30-token
    questions, ragged lengths incl. 1 and 30) with the north-star widths (E=300, H=1024, mid=1024) on 64x64 images."""
    from oracle import vqa_oracle as O
    from dl_vqa_amd.train import soft_ce_loss_and_score
    cfg = full_cfg(3000)
    V, B, S, T = 5000, 5, 64, 30
    torch.manual_seed(4)
    m = build(cfg, V).eval()
    v, q, a_idx, a_val, _, _, ql = O.synthetic_batch(B, S, T, V, 3000, seed=6)
    ql = torch.tensor([30, 1, 17, 30, 8])
    q = q * (torch.arange(T)[None, :] < ql[:, None])
    q[0, 4] = 0                                        # padding / unknown token inside a question
    sd = {k: t.detach().cpu().clone() for k, t in m.state_dict().items()}
    y_ref, loss_ref, grads_ref = O.loss_and_grads(sd, cfg, v, q, ql, a_idx, a_val)
    y = m(v.to(DEV), q.to(DEV), ql.to(DEV))
    loss, _ = soft_ce_loss_and_score(y, a_idx.to(DEV), a_val.to(DEV))
    loss.backward()
    torch.cuda.synchronize()
    err = float((y.detach().cpu() - y_ref).abs().max())
    print(f"[parity] T=30 A=3000 logits max abs err {err:.3e}; loss {float(loss):.6f} vs {float(loss_ref):.6f}")
    assert y.shape == (B, 3000) and err < 1e-3 and err < 1e-5
    assert abs(float(loss) - float(loss_ref)) < 1e-5
    for k, p in m.named_parameters():
        e = grad_err(k, p.grad, grads_ref[k])
        print(f"[parity] T=30 A=3000 grad {k}: {e:.3e}")
        assert e < 2e-4, (k, e)


def test_headline_batch_256_is_batch_invariant():
    """BASELINE configs[1] at its real size: B=256, 224x224, T=14, A=1000, eval mode.  The batch is the two
    samples of the reference fixture full224_seed1 repeated 128 times, so every logits row must equal the
    reference's row for that sample and every parameter gradient (a mean over the batch) must equal the B=2
    gradient the reference produced -- which pins the tile plans, split counts and grids that only run at full
    size (plan_wgrad's splits, xcd_swizzle over 10^4+ workgroups, the 192-/384-row wgrad tiles, the LSTM step
    kernel at M=256) without any CPU computation at B=256."""
    from dl_vqa_amd.train import soft_ce_loss_and_score
    g = Golden("full224_seed1")
    meta = g.meta
    torch.manual_seed(meta["seed"])
    m = build(full_cfg(meta["A"]), meta["V"]).eval()
    v, q, ql, a_idx, a_val, _ = full_inputs(meta)
    R = 128
    rep = lambda t: t.repeat(R, *([1] * (t.dim() - 1))).to(DEV)
    y = m(rep(v), rep(q), rep(ql))
    loss, score = soft_ce_loss_and_score(y, rep(a_idx), rep(a_val))
    loss.backward()
    torch.cuda.synchronize()
    assert y.shape == (2 * R, meta["A"])
    ref = g.t["logits"].repeat(R, 1)
    err = float((y.detach().cpu() - ref).abs().max())
    print(f"[parity] B=256 logits max abs err {err:.3e}; loss {float(loss):.6f} vs {float(g.t['loss']):.6f}")
    assert err < 1e-3
    assert float((y[0::2] - y[0]).abs().max()) < 1e-5 and float((y[1::2] - y[1]).abs().max()) < 1e-5
    assert abs(float(loss) - float(g.t["loss"])) < 1e-4
    gnames = [str(n) for n in g.raw["grad_names"]]
    grads = dict((k, p.grad) for k, p in m.named_parameters())
    for n, l2 in zip(gnames, g.raw["grad_l2"]):
        if n == ZERO_GRAD:
            assert float(grads[n].abs().max()) < 1e-6
            continue
        flat = grads[n].flatten().double().cpu()
        step = max(1, flat.numel() // 257)
        sample = flat[::step][:257]
        refg = torch.from_numpy(g.raw["gsample/" + n]).double()
        e = float((sample - refg).abs().max()) / max(float(refg.abs().max()), 1e-30)
        e2 = abs(float(flat.pow(2).sum().sqrt()) - l2) / max(l2, 1e-30)
        print(f"[parity] B=256 grad {n}: sample {e:.3e} l2 {e2:.3e}")
        assert e < 2e-3 and e2 < 1e-3, (n, e, e2)


def test_headline_batch_256_distinct_samples_match_oracle():
    """BASELINE configs[1] at its real size with 256 DISTINCT samples (VERDICT r2 'weak' 2: the repeated-pair test
    above cannot see an indexing slip that maps sample b to b +- 2k).
    (1) forward: B=256, 224x224, ragged question lengths, eval mode -- every logits row and the loss against the CPU
        oracle (run in chunks of 32 samples on the host cores; forward only, ~1 TFLOP);
    (2) backward: 16 distinct samples tiled 16 times (B=256, period 16, coprime-free of the pair test's period 2):
        every parameter gradient against the oracle's gradient for the 16 samples (a mean over the batch is invariant
        under tiling), which walks plan_wgrad's real split counts, the batch chunking and the XCD swizzle with
        sample-dependent data in every slot."""
    from oracle import vqa_oracle as O
    from dl_vqa_amd.train import soft_ce_loss_and_score
    cfg = full_cfg(1000)
    V, B, S, T, A = 5000, 256, 224, 14, 1000
    torch.manual_seed(21)
    m = build(cfg, V).eval()
    sd = {k: t.detach().cpu().clone() for k, t in m.state_dict().items()}
    v, q, a_idx, a_val, _, _, ql = O.synthetic_batch(B, S, T, V, A, seed=31)         # full_len=False: ragged lengths
    assert len(set(ql.tolist())) > 4 and float((v[0] - v[1]).abs().max()) > 0
    with torch.no_grad():
        y = m(v.to(DEV), q.to(DEV), ql.to(DEV))
        loss, _ = soft_ce_loss_and_score(y, a_idx.to(DEV), a_val.to(DEV))
    torch.cuda.synchronize()
    y_ref = torch.empty(B, A)
    with torch.no_grad():
        for b0 in range(0, B, 32):
            sl = slice(b0, b0 + 32)
            y_ref[sl] = O.vqa_forward(sd, cfg, v[sl], q[sl], ql[sl])
    loss_ref = O.soft_ce_loss(y_ref, a_idx, a_val)
    err = (y.cpu() - y_ref).abs().max(dim=1).values
    print(f"[parity] B=256 distinct: logits max abs err {float(err.max()):.3e} (worst sample {int(err.argmax())}); "
          f"loss {float(loss):.6f} vs {float(loss_ref):.6f}")
    assert float(err.max()) < 1e-3 and float(err.max()) < 1e-5
    assert abs(float(loss) - float(loss_ref)) < 1e-5
    # rows really differ (a permutation of samples would show): the closest OTHER reference row is far away
    d01 = float((y_ref[0] - y_ref[1]).abs().max())
    assert d01 > 100 * float(err.max())

    n = 16
    vt, qt, qlt, ait, avt = (t[:n].repeat(B // n, *([1] * (t.dim() - 1))) for t in (v, q, ql, a_idx, a_val))
    y2 = m(vt.to(DEV), qt.to(DEV), qlt.to(DEV))
    loss2, _ = soft_ce_loss_and_score(y2, ait.to(DEV), avt.to(DEV))
    loss2.backward()
    torch.cuda.synchronize()
    # reference gradients in float64: the first block's weight gradient is a sum of 256 x 222 x 222 = 1.3e7 terms per
    # element, where an fp32 CPU reference carries as much rounding noise as the kernel under test (first GPU run: 2.1e-4
    # between the two fp32 results)
    sd64 = {k: t.double() for k, t in sd.items()}
    _, loss16, g16 = O.loss_and_grads(sd64, cfg, v[:n].double(), q[:n], ql[:n], a_idx[:n], a_val[:n])
    assert abs(float(loss2) - float(loss16)) < 1e-5
    assert float((y2.detach().cpu() - y_ref[:n].repeat(B // n, 1)).abs().max()) < 1e-5
    worst = {}
    for k, p in m.named_parameters():
        e = grad_err(k, p.grad, g16[k])
        print(f"[parity] B=256 (16 distinct x 16) grad {k}: {e:.3e}")
        worst[k] = e
    # fp32 accumulation noise against float64 at this size (measured: convolutions 2e-4, v_conv 9e-4, q_lin 1.5e-3 of the
    # largest entry -- sums over 676 positions x 256 samples with cancellation; the B=2 reference fixture shows the same
    # levels): the tolerance of test_headline_batch_256_is_batch_invariant.  A mis-indexed sample is an O(1) error.
    for k, e in worst.items():
        assert e < 2e-3, (k, e)


def test_matches_cpu_oracle_on_random_batch():
    """A shape no fixture covers (B=5, S=48, T=7 ragged lengths): HIP vs the oracle run in float64."""
    from oracle import vqa_oracle as O
    from dl_vqa_amd.train import soft_ce_loss_and_score
    cfg = tiny_cfg(dict(bidirectional=True, stride=1, do_option="+"))
    torch.manual_seed(11)
    m = build(cfg, 40).eval()
    v, q, a_idx, a_val, a_len, _, ql = O.synthetic_batch(5, 48, 7, 40, 12, seed=3)
    sd64 = {k: t.double().cpu() for k, t in m.state_dict().items()}
    y_ref, loss_ref, grads_ref = O.loss_and_grads(sd64, cfg, v.double(), q, ql, a_idx, a_val)
    y = m(v.to(DEV), q.to(DEV), ql.to(DEV))
    loss, _ = soft_ce_loss_and_score(y, a_idx.to(DEV), a_val.to(DEV))
    loss.backward()
    torch.cuda.synchronize()
    assert float((y.cpu().double() - y_ref).abs().max()) < 1e-5
    assert abs(float(loss) - float(loss_ref)) < 1e-5
    for k, p in m.named_parameters():
        assert grad_err(k, p.grad, grads_ref[k]) < 1e-4, k


def test_eval_is_deterministic_and_no_grad_path_matches():
    cfg = tiny_cfg(dict(bidirectional=True, stride=1, do_option="+"))
    torch.manual_seed(2)
    m = build(cfg, 30).eval()
    v = torch.randn(4, 3, 32, 32, device=DEV)
    q = torch.randint(1, 30, (4, 6), device=DEV)
    ql = torch.tensor([6, 2, 4, 1], device=DEV)
    y1 = m(v, q, ql)
    with torch.no_grad():
        y2 = m(v, q, ql)
    assert torch.equal(y1.detach(), y2)


def test_train_mode_dropout_fused_adam_reduces_loss():
    """Train mode (all 7 dropout sites active) + FusedAdam: gradients finite, loss goes down on a
    fixed batch, eval output changes accordingly; torch.manual_seed makes it repeatable."""
    from oracle import vqa_oracle as O
    from dl_vqa_amd.train import FusedAdam, run_batch, update_learning_rate
    cfg = tiny_cfg(dict(bidirectional=True, stride=1, do_option="+"))

    def run(seed):
        torch.manual_seed(seed)
        m = build(cfg, 40).train()
        batch = O.synthetic_batch(8, 32, 6, 40, 12, seed=5)
        opt = FusedAdam(m, lr=5e-3)
        losses = []
        for it in range(12):
            loss, score = run_batch(m, None, batch, 12)
            opt.zero_grad()
            update_learning_rate(opt, it, 5e-3)
            loss.backward()
            for p in m.parameters():
                assert p.grad is not None and bool(torch.isfinite(p.grad).all())
            opt.step()
            losses.append(float(loss))
        return losses

    a, b = run(7), run(7)
    assert a == b                                     # same seed -> same masks -> same trajectory
    assert a[-1] < a[0], a
    assert run(8) != a                                # another seed -> other dropout masks


def test_fused_adam_step_matches_oracle_and_checkpoint_format():
    from oracle import vqa_oracle as O
    from dl_vqa_amd.train import FusedAdam, run_batch
    g = Golden("tiny_plus")
    cfg = tiny_cfg(g.meta)
    m = build(cfg, g.meta["V"], g.sd).eval()
    batch = (g.t["v"], g.t["q"], g.t["a_idx"], g.t["a_val"], g.t["a_len"], torch.arange(3), g.t["q_len"])
    opt = FusedAdam(m, lr=5e-4)
    loss, _ = run_batch(m, None, batch, 12)
    opt.zero_grad()
    loss.backward()
    opt.step()
    torch.cuda.synchronize()
    for k, p in m.named_parameters():
        if k == ZERO_GRAD:
            continue
        ref = g.sd[k].clone()
        O.adam_step(ref, g.grad[k], torch.zeros_like(ref), torch.zeros_like(ref), 1, 5e-4)
        # Adam's first step moves every weight by ~lr*sign(g): compare the UPDATE, not the weight
        upd, upd_ref = p.detach().cpu() - g.sd[k], ref - g.sd[k]
        # (where |g| ~ eps = 1e-8 the update is ill-conditioned in g, so only well-scaled entries)
        big = g.grad[k].abs() > 1e-3 * g.grad[k].abs().max()
        if bool(big.any()):
            assert float((upd - upd_ref)[big].abs().max()) < 1e-2 * 5e-4, k
    sd = opt.state_dict()
    ref_opt = torch.optim.Adam([torch.nn.Parameter(t.clone()) for t in g.sd.values()], lr=5e-4)
    assert set(sd["param_groups"][0]) >= {"lr", "betas", "eps", "params"}
    ref_opt.load_state_dict(sd)                       # the reference resumes with torch.optim.Adam (train.py:56-57)
    opt2 = FusedAdam(m, lr=1.0)
    opt2.load_state_dict(sd)
    assert opt2.step_count == 1 and torch.equal(opt2.exp_avg, opt.exp_avg)


def test_two_forwards_before_one_backward_match_oracle():
    """(loss(model(a)) + loss(model(b))).backward() -- the second backward must not overwrite the
    first one's gradients in the shared flat buffer; and a later FusedAdam.step must see the accumulated sum."""
    from oracle import vqa_oracle as O
    from dl_vqa_amd.train import FusedAdam, soft_ce_loss_and_score
    g = Golden("tiny_plus")
    cfg = tiny_cfg(g.meta)
    m = build(cfg, g.meta["V"], g.sd).eval()
    b1 = O.synthetic_batch(3, 32, 5, g.meta["V"], 12, seed=21)
    b2 = O.synthetic_batch(4, 32, 6, g.meta["V"], 12, seed=22)

    def hip_loss(b):
        v, q, a_idx, a_val, _, _, ql = b
        return soft_ce_loss_and_score(m(v.to(DEV), q.to(DEV), ql.to(DEV)), a_idx.to(DEV), a_val.to(DEV))[0]

    (hip_loss(b1) + hip_loss(b2)).backward()
    torch.cuda.synchronize()
    refs = [O.loss_and_grads(g.sd, cfg, b[0], b[1], b[6], b[2], b[3])[2] for b in (b1, b2)]
    for k, p in m.named_parameters():
        assert grad_err(k, p.grad, refs[0][k] + refs[1][k]) < 2e-4, k
    opt = FusedAdam(m, lr=1e-3)
    opt.step()                                          # gathers p.grad into the flat buffer first
    torch.cuda.synchronize()
    for k, p in m.named_parameters():
        o, n = m._offsets[k]
        assert torch.equal(m._flat_grad[o:o + n].view(p.shape), p.grad), k


def test_out_of_vocabulary_token_ids_raise():
    """nn.Embedding raises for ids outside [0, V) (models/model.py:155): host-resident ids are checked before the
    upload, device-resident ids are counted by the kernel and reported at check_token_ids() / the next forward."""
    g = Golden("tiny_plus")
    m = build(tiny_cfg(g.meta), g.meta["V"], g.sd).eval()
    v, q, ql = g.t["v"].to(DEV), g.t["q"].clone(), g.t["q_len"].to(DEV)
    q[1, 0] = g.meta["V"]
    with torch.no_grad():
        with pytest.raises(IndexError, match="out of range"):
            m(v, q, ql)                                  # q on the host
        y = m(v, q.to(DEV), ql)                          # q on the device: counted, embedded as zeros
        assert bool(torch.isfinite(y).all())
        with pytest.raises(IndexError, match="1 question token"):
            m.check_token_ids()
        m(v, g.t["q"].to(DEV), ql)                       # valid ids: nothing pending afterwards
        m.check_token_ids()


def test_bad_question_lengths_raise_like_pack_padded_sequence():
    """models/model.py:159-162: pack_padded_sequence raises for a length of 0 or one above the padded width; the mirror raises
    for host-resident lengths (what the data loader hands to run_batch)."""
    from dl_vqa_amd.train import run_batch
    g = Golden("tiny_plus")
    m = build(tiny_cfg(g.meta), g.meta["V"], g.sd).eval()
    v, q = g.t["v"].to(DEV), g.t["q"].to(DEV)
    for bad in ([5, 0, 1], [5, 3, 6]):
        with torch.no_grad(), pytest.raises(RuntimeError, match="question length"):
            m(v, q, torch.tensor(bad))
        batch = (g.t["v"], g.t["q"], g.t["a_idx"], g.t["a_val"], g.t["a_len"], None, torch.tensor(bad))
        with torch.no_grad(), pytest.raises(RuntimeError, match="question length"):
            run_batch(m, None, batch, 12)
    with torch.no_grad():
        y = m(v, q, g.t["q_len"])                        # valid host-resident lengths: same logits as device-resident ones
        assert torch.equal(y, m(v, q, g.t["q_len"].to(DEV)))


def test_cpu_tensors_are_rejected():
    cfg = tiny_cfg(dict(bidirectional=True, stride=1, do_option="+"))
    from dl_vqa_amd import VqaNet
    m = VqaNet(cfg, 30)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.randn(1, 3, 32, 32), torch.ones(1, 4, dtype=torch.int64), torch.tensor([4]))


@pytest.mark.parametrize("do_option", ["+", "*", "|"])
def test_train_mode_matches_oracle_with_shared_masks(do_option):
    """Train mode (all 7 dropout sites, p = 0.3) against the oracle: the masks the HIP kernels generate are
    extracted as data (tests/hip_masks.py) and given to the oracle, whose mask PLACEMENT is pinned by
    reference-generated train-mode fixtures (tests/test_oracle_golden.py::test_train_mode_with_recorded_masks).
    Logits within 1e-3 (north star), every gradient within 2e-4 relative."""
    from oracle import vqa_oracle as O
    from dl_vqa_amd.train import soft_ce_loss_and_score
    from tests.hip_masks import hip_masks
    g = Golden({"+": "tiny_plus", "*": "tiny_mul", "|": "tiny_cat"}[do_option])
    cfg = tiny_cfg(g.meta)
    m = build(cfg, g.meta["V"], g.sd).train()
    v, q, ql = g.t["v"], g.t["q"], g.t["q_len"]
    torch.manual_seed(123)
    y = m(v.to(DEV), q.to(DEV), ql.to(DEV))
    loss, _ = soft_ce_loss_and_score(y, g.t["a_idx"].to(DEV), g.t["a_val"].to(DEV))
    loss.backward()
    torch.cuda.synchronize()
    ctx = m._last_ctx
    assert ctx.p_att == 0.3 and ctx.seed != 0
    masks = hip_masks(m._engine, ctx.seed, v.shape[0], q.shape[1], ctx.acts[-1].shape[1], DEV)
    for k, mk in masks.items():      # real masks: values in {0, 1/(1-p)}, both present
        vals = set(round(float(x), 4) for x in mk.unique())
        assert vals == {0.0, round(1 / 0.7, 4)}, (k, vals)
    y_ref, loss_ref, grads_ref = O.loss_and_grads(g.sd, cfg, v, q, ql, g.t["a_idx"], g.t["a_val"], masks=masks)
    err = float((y.detach().cpu() - y_ref).abs().max())
    print(f"[parity] train-mode ({do_option}) logits max abs err {err:.3e}; loss {float(loss):.6f} vs {float(loss_ref):.6f}")
    assert err < 1e-3 and err < 2e-5
    assert abs(float(loss) - float(loss_ref)) < 1e-5
    y_eval = O.vqa_forward(g.sd, cfg, v, q, ql)
    assert float((y_ref - y_eval).abs().max()) > 1e-2          # the masks matter: eval-mode logits are far away
    for k, p in m.named_parameters():
        e = grad_err(k, p.grad, grads_ref[k], "+")             # in train mode q_lin's gradient is not zero for '|'
        print(f"[parity] train-mode ({do_option}) grad {k}: {e:.3e}")
        assert e < 2e-4, (k, e)


@pytest.mark.parametrize("compute_dtype", ["fp32", "fp32x3"])
def test_train_mode_full224_matches_oracle_with_shared_masks(compute_dtype):
    """The north-star architecture (224x224, B=2, T=14, A=1000) in train mode, HIP vs oracle with shared masks; the same
    tolerances for the fp32x3 mode (conv blocks 1-2 and v_conv on the bf16 matrix cores through exact operand splits)."""
    from oracle import vqa_oracle as O
    from dl_vqa_amd.train import soft_ce_loss_and_score
    from tests.hip_masks import hip_masks
    g = Golden("full224_seed1")
    meta = g.meta
    torch.manual_seed(meta["seed"])
    cfg = full_cfg(meta["A"])
    m = build(cfg, meta["V"], compute_dtype=compute_dtype).train()
    sd = {k: t.detach().cpu().clone() for k, t in m.state_dict().items()}
    v, q, ql, a_idx, a_val, _ = full_inputs(meta)
    y = m(v.to(DEV), q.to(DEV), ql.to(DEV))
    loss, _ = soft_ce_loss_and_score(y, a_idx.to(DEV), a_val.to(DEV))
    loss.backward()
    torch.cuda.synchronize()
    ctx = m._last_ctx
    masks = hip_masks(m._engine, ctx.seed, 2, q.shape[1], ctx.acts[-1].shape[1], DEV)
    y_ref, loss_ref, grads_ref = O.loss_and_grads(sd, cfg, v, q, ql, a_idx, a_val, masks=masks)
    err = float((y.detach().cpu() - y_ref).abs().max())
    print(f"[parity] train-mode full224 ({compute_dtype}) logits max abs err {err:.3e}; loss {float(loss):.6f} vs {float(loss_ref):.6f}")
    assert err < 1e-3
    assert abs(float(loss) - float(loss_ref)) < 1e-4
    for k, p in m.named_parameters():
        e = grad_err(k, p.grad, grads_ref[k], "+")
        print(f"[parity] train-mode full224 ({compute_dtype}) grad {k}: {e:.3e}")
        assert e < 1e-3, (k, e)        # fp32 both sides, K up to 6.4e5 products per element in the conv wgrads


@pytest.mark.parametrize("do_option", ["+", "*", "|"])
def test_train_mode_gradient_matches_finite_difference(do_option):
    """Train mode (dropout masks fixed by the seed) is a deterministic function of the weights: the HIP
    backward, masks regenerated from the same counter hash, must agree with a central finite difference of
    the HIP forward along a random direction over ALL parameters.  Covers the dropout sites of every
    do_option, which the eval-mode reference fixtures cannot."""
    from dl_vqa_amd.train import soft_ce_loss_and_score
    from oracle import vqa_oracle as O
    cfg = tiny_cfg(dict(bidirectional=True, stride=1, do_option=do_option))
    for sec in ("text", "image", "attention", "classifier"):
        cfg[sec]["dropout"] = 0.25
    torch.manual_seed(5)
    m = build(cfg, 40).train()
    v, q, a_idx, a_val, _, _, ql = O.synthetic_batch(6, 32, 5, 40, 12, seed=4)
    v, q, ql, a_idx, a_val = v.to(DEV), q.to(DEV), ql.to(DEV), a_idx.to(DEV), a_val.to(DEV)
    seed = 987654321

    def loss_at():
        with torch.no_grad():
            logits, _ = m._engine.forward(m._param_dict(), v, q, ql, True, seed, keep=False)
            return float(soft_ce_loss_and_score(logits, a_idx, a_val)[0])

    m._ensure_flat()
    logits, ctx = m._engine.forward(m._param_dict(), v, q, ql, True, seed, keep=True)
    B, A = logits.shape
    dl = torch.zeros(B, A, device=DEV)
    from dl_vqa_amd import ops
    ops.softce(logits, A, a_idx, a_val, A, 1.0 / B, dl, A)
    grads = m._grad_views()
    m._engine.backward(m._param_dict(), ctx, dl, grads)
    g = torch.Generator().manual_seed(1)
    flat_p = m._flat_param
    direction = torch.zeros_like(flat_p)
    gdot = 0.0
    for name, p in m.named_parameters():
        o, n = m._offsets[name]
        d = torch.randn(n, generator=g).to(DEV) * float(p.detach().abs().mean() + 1e-3)
        direction[o:o + n] = d
        gdot += float((grads[name].reshape(-1).double() * d.double()).sum())
    eps = 2e-3
    base = flat_p.clone()
    flat_p.copy_(base + eps * direction)
    lp = loss_at()
    flat_p.copy_(base - eps * direction)
    lm = loss_at()
    flat_p.copy_(base)
    fd = (lp - lm) / (2 * eps)
    print(f"[parity] train-mode directional derivative ({do_option}): backward {gdot:.6f} vs finite difference {fd:.6f}")
    assert abs(fd - gdot) <= 0.03 * max(abs(fd), abs(gdot)) + 1e-4
