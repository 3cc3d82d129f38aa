"""GPU parity of the bf16 path (BASELINE configs[3]) through the C ABI.

Reference for every bf16 kernel: the same operator in float64 on operands ROUNDED TO bf16 (products of two bf16
values are exact in fp32, so what remains is the fp32 accumulation order): tolerance 3e-6 * sqrt(K) relative to the
largest result for fp32 outputs, one bf16 ulp (2^-8 relative) for bf16 outputs -- stated per test."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ops():
    from dl_vqa_amd import ops
    return ops


def rb(x):
    """round to bf16 and back (what the bf16 storage does)"""
    return x.to(torch.bfloat16).to(torch.float32)


def rel_err(got, ref):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    return float((got - ref).abs().max()) / max(float(ref.abs().max()), 1e-30)


def check(name, got, ref, tol):
    e = rel_err(got, ref)
    print(f"[parity-bf16] {name}: max|err|/max|ref| = {e:.3e} (tol {tol:.1e})")
    assert e <= tol, f"{name}: {e} > {tol}"


def test_converters_round_to_nearest_even():
    ops = _ops()
    g = torch.Generator().manual_seed(1)
    x = torch.randn(1000 * 37 + 5, generator=g) * 3
    x[:4] = torch.tensor([1.0 + 2 ** -8, 1.0 + 3 * 2 ** -8, -0.0, 65280.0])    # ties: to even mantissa
    y = ops.to_bf16(x.to(DEV))
    torch.cuda.synchronize()
    assert torch.equal(y.cpu(), x.to(torch.bfloat16))
    assert torch.equal(ops.to_f32(y).cpu(), x.to(torch.bfloat16).float())
    w = torch.randn(70, 45, generator=g)
    assert torch.equal(ops.to_bf16_transposed(w.to(DEV)).cpu(), w.t().contiguous().to(torch.bfloat16))


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (304, 200, 96), (64, 64, 64), (8, 8, 8), (256, 1024, 2560),
                                   (1032, 264, 3584), (136, 4096, 304), (72, 56, 40), (200, 136, 1000), (384, 256, 66008), (65544, 256, 328)])
@pytest.mark.parametrize("transA,transB", [(False, True), (False, False), (True, True), (True, False)])
def test_gemm_bf16_layouts(M, N, K, transA, transB):
    """Every operand layout: k-contiguous rows (the fp32 engine's LDS image reused) and reduction-major rows (the
    ds_read_b64_tr_b16 transpose-read image); asymmetric random data, tiles with edges, K tails, split-K plans; the last
    but one shape with transA / not transB is the long-K weight gradient with a 256-column output (128 x 256 tiles, K tail of 8), the
    last one is a tall product with a ragged last row tile and a K tail (the v_conv backward-data shape class)."""
    ops = _ops()
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    A = rb(torch.randn(M, K, generator=g))
    Bm = rb(torch.randn(K, N, generator=g))
    ref = A.double() @ Bm.double()
    As = (A.t().contiguous() if transA else A).to(torch.bfloat16).to(DEV)      # [K][M] or [M][K]
    Bs = (Bm.t().contiguous() if transB else Bm).to(torch.bfloat16).to(DEV)    # [N][K] or [K][N]
    Cd = torch.full((M, N + 3), 7.0, device=DEV)
    ops.gemm_bf16(As, Bs, Cd, M, N, K, transA=transA, transB=transB, ldc=N + 3)
    torch.cuda.synchronize()
    check(f"gemm_bf16 {M}x{N}x{K} tA={transA} tB={transB}", Cd[:, :N], ref, 3e-6 * math.sqrt(K))
    assert float((Cd[:, N:] - 7.0).abs().max()) == 0.0


def test_gemm_bf16_epilogue_and_bf16_output():
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    Bn, P, K, N = 3, 7, 40, 40
    M = Bn * P
    A, W = rb(torch.randn(M, K, generator=g)), rb(torch.randn(N, K, generator=g))
    b1, rg = torch.randn(N, generator=g), torch.randn(Bn, N, generator=g)
    acc = A.double() @ W.double().t()
    ref = torch.relu(acc + rg.double().repeat_interleave(P, dim=0) + b1.double())
    Cf = torch.empty(M, N, device=DEV)
    Cb = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    for C in (Cf, Cb):
        ops.gemm_bf16(A.to(torch.bfloat16).to(DEV), W.to(torch.bfloat16).to(DEV), C, M, N, K, bias1=b1.to(DEV),
                      rowgroup=rg.to(DEV), rg_div=P, relu=True)
    torch.cuda.synchronize()
    check("gemm_bf16 epilogue fp32 out", Cf, ref, 1e-5)
    check("gemm_bf16 epilogue bf16 out", Cb.float(), ref, 2 ** -8)
    assert torch.equal(Cb.cpu(), Cf.cpu().to(torch.bfloat16))        # the bf16 result is the rounded fp32 result


@pytest.mark.parametrize("Bn,P,K,N,op", [(8, 676, 256, 1024, 0), (3, 2916, 256, 256, 1), (40, 300, 192, 512, 0), (17, 1000, 64, 128, None),
                                         (33, 2916, 512, 384, 1), (9, 4000, 128, 1024, None)])
def test_gemm_tall_bf16(Bn, P, K, N, op):
    """csrc/gemm_tall_bf16.hip (persistent 256 x 128 tiles, short K) against the float64 product of the bf16-rounded
    operands: row-group add / multiply / none, ReLU, tiles that span two groups, a last row tile that is partly past M,
    several tiles per workgroup (the three-stage ring crosses tile boundaries), one to eight k-stages per tile.  The result must equal the rounded fp32 result of vqa_gemm_bf16 up to summation order."""
    ops = _ops()
    M = Bn * P
    assert ops.gemm_tall_bf16_supported(M, N, K, P, op is not None)
    g = torch.Generator().manual_seed(Bn * 7 + K)
    A, W = rb(torch.randn(M, K, generator=g)), rb(torch.randn(N, K, generator=g))
    rg = torch.randn(Bn, N, generator=g)
    acc = A.double() @ W.double().t()
    rgx = rg.double().repeat_interleave(P, dim=0)
    ref = torch.relu(acc if op is None else (acc * rgx if op == 1 else acc + rgx))
    C = torch.full((M, N), 3.0, dtype=torch.bfloat16, device=DEV)
    ops.gemm_tall_bf16(A.to(torch.bfloat16).to(DEV), W.to(torch.bfloat16).to(DEV), C, M, N, K,
                       rowgroup=(rg.to(DEV) if op is not None else None), rg_div=P, rg_op=(op or 0), relu=True)
    torch.cuda.synchronize()
    check(f"gemm_tall_bf16 {M}x{N}x{K} op={op}", C.float(), ref, 2 ** -7)
    near = ref.float().to(torch.bfloat16)
    assert float((C.cpu() != near).float().mean()) < 2e-3          # a bf16 rounding flips only where fp32 sums differ in the last bit


@pytest.mark.parametrize("Bn,P,K,N", [(3, 200, 256, 256), (2, 676, 64, 1024), (5, 129, 96, 136)])
def test_gemm_bf16_staged_bf16_output(Bn, P, K, N):
    """bf16 result of 128 x 128 tiles: interior tiles store through the wave-private LDS scratch (16 bytes per lane), edge
    tiles keep the element-wise stores; row-group term (one / two groups per tile), bias, ReLU.  The bf16 result must be
    the rounded fp32 result of the same GEMM, element for element."""
    ops = _ops()
    g = torch.Generator().manual_seed(Bn * 100 + N)
    M = Bn * P
    A, W = rb(torch.randn(M, K, generator=g)), rb(torch.randn(N, K, generator=g))
    b1, rg = torch.randn(N, generator=g), torch.randn(Bn, N, generator=g)
    ref = torch.relu(A.double() @ W.double().t() + rg.double().repeat_interleave(P, dim=0) + b1.double())
    Cf = torch.empty(M, N, device=DEV)
    Cb = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    for C in (Cf, Cb):
        ops.gemm_bf16(A.to(torch.bfloat16).to(DEV), W.to(torch.bfloat16).to(DEV), C, M, N, K, bias1=b1.to(DEV),
                      rowgroup=rg.to(DEV), rg_div=P, relu=True)
    torch.cuda.synchronize()
    check(f"gemm_bf16 staged epilogue fp32 out {M}x{N}x{K}", Cf, ref, 1e-5)
    assert torch.equal(Cb.cpu(), Cf.cpu().to(torch.bfloat16))


@pytest.mark.parametrize("B,H,W,Ci,Co,stride", [(2, 30, 30, 64, 128, 1), (1, 30, 34, 128, 256, 1), (3, 22, 20, 64, 64, 1),
                                                (2, 41, 37, 64, 128, 2), (2, 58, 58, 64, 128, 1)])
def test_conv_bf16_fwd_dgrad_wgrad(B, H, W, Ci, Co, stride):
    """The three bf16 convolution kernels against float64 autograd on bf16-rounded x, w, dy: forward (bf16 and fp32
    outputs), dgrad (bf16 and fp32 outputs), wgrad + bias gradient (fp32)."""
    ops = _ops()
    g = torch.Generator().manual_seed(B * 1000 + H * 10 + Ci)
    x = rb(torch.randn(B, Ci, H, W, generator=g))
    w = rb(torch.randn(Co, Ci, 3, 3, generator=g) / math.sqrt(9 * Ci))
    b = torch.randn(Co, generator=g) * 0.1
    xr, wr, br = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    yr = F.max_pool2d(torch.relu(F.conv2d(xr, wr, br, stride=stride)), 2, 2)
    dy = rb(torch.randn(yr.shape, generator=g))
    yr.backward(dy.double())

    xd = x.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(DEV)
    wfT, wdT = ops.conv_pack_weights_bf16(w.to(DEV), Ci)
    p32, amax = ops.conv_fwd_bf16(xd, wfT, b.to(DEV), stride, out_dtype=torch.float32)
    p16, amax2 = ops.conv_fwd_bf16(xd, wfT, b.to(DEV), stride)
    torch.cuda.synchronize()
    tag = f"{B,H,W,Ci,Co,stride}"
    check(f"conv_bf16 fwd fp32-out {tag}", p32.permute(0, 3, 1, 2), yr, 3e-6 * math.sqrt(9 * Ci))
    assert torch.equal(amax, amax2) and torch.equal(p16, p32.to(torch.bfloat16))
    assert bool(((p32 == 0) == (amax == 4)).all())

    dyd = dy.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(DEV)
    dx32 = ops.conv_dgrad_bf16(dyd, amax, wdT, xd.shape, stride, out_dtype=torch.float32)
    dx16 = ops.conv_dgrad_bf16(dyd, amax, wdT, xd.shape, stride)
    dw, db = torch.empty(Co, Ci, 3, 3, device=DEV), torch.empty(Co, device=DEV)
    ops.conv_wgrad_bf16(xd, dyd, amax, dw, db, stride)
    torch.cuda.synchronize()
    check(f"conv_bf16 dgrad fp32-out {tag}", dx32.permute(0, 3, 1, 2), xr.grad, 5e-6 * math.sqrt(9 * Co))
    assert torch.equal(dx16, dx32.to(torch.bfloat16))
    check(f"conv_bf16 wgrad {tag}", dw, wr.grad, 2e-5)
    check(f"conv_bf16 bias grad {tag}", db, br.grad, 2e-5)


# ----------------------------------------------------------------------------- the whole module in bf16 mode
def bf16_cfg(do_option="+", p=0.0):
    return {
        "text": {"question_features": 32, "embedding_features": 20, "dropout": p, "num_lstm_layers": 1, "bidirectional": True},
        "image": {"kernel_size": 3, "dropout": p, "num_channels": [3, 64, 64, 128], "stride": 1, "do_skip_connection": False},
        "attention": {"hidden_dim": 64, "glimpses": 2, "do_option": do_option, "dropout": p},
        "classifier": {"hidden_dim": 40, "dropout": p},
        "max_answers": 24,
    }


@pytest.mark.parametrize("do_option,train", [("+", False), ("+", True), ("*", True), ("|", False)])
def test_bf16_module_matches_bf16_oracle(do_option, train):
    """VqaNet(compute_dtype="bf16") against the oracle with the SAME rounding points (oracle.vqa_forward(bf16=True):
    activations between conv blocks, conv / v_conv weights, the v_conv input, and in backward the stored bf16
    gradients), eval mode and train mode with shared dropout masks.  Tolerance: the two sides differ by fp32
    accumulation order, which can flip a bf16 rounding (2^-9 relative) of a few stored elements: logits 5e-5
    absolute, gradients 2e-2 of the largest entry -- an order of magnitude below the distance to the fp32 result,
    which is asserted as well."""
    from oracle import vqa_oracle as O
    from dl_vqa_amd import VqaNet
    from dl_vqa_amd.train import soft_ce_loss_and_score
    from tests.hip_masks import hip_masks
    cfg = bf16_cfg(do_option, 0.3 if train else 0.0)
    V, B, S, T = 50, 3, 48, 6
    torch.manual_seed(3)
    m = VqaNet(cfg, V, compute_dtype="bf16").to(DEV)
    m.train(train)
    sd = {k: t.detach().cpu().clone() for k, t in m.state_dict().items()}
    v, q, a_idx, a_val, _, _, ql = O.synthetic_batch(B, S, T, V, 24, seed=5)
    torch.manual_seed(9)
    y = m(v.to(DEV), q.to(DEV), ql.to(DEV))
    loss, _ = soft_ce_loss_and_score(y, a_idx.to(DEV), a_val.to(DEV))
    loss.backward()
    torch.cuda.synchronize()
    ctx = m._last_ctx
    assert ctx.acts[1].dtype == torch.bfloat16 and ctx.acts[2].dtype == torch.bfloat16 and ctx.acts[3].dtype == torch.float32
    masks = hip_masks(m._engine, ctx.seed, B, T, ctx.acts[-1].shape[1], DEV) if train else None
    y_ref, loss_ref, g_ref = O.loss_and_grads(sd, cfg, v, q, ql, a_idx, a_val, masks=masks, bf16=True)
    y_f32, _, g_f32 = O.loss_and_grads(sd, cfg, v, q, ql, a_idx, a_val, masks=masks)
    err = float((y.detach().cpu() - y_ref).abs().max())
    dist = float((y_f32 - y_ref).abs().max())
    print(f"[parity-bf16] module ({do_option}, train={train}) logits |err| vs bf16 oracle {err:.3e}; bf16 vs fp32 oracle {dist:.3e}")
    assert err < 5e-5 and err < 0.5 * dist          # measured 1.2e-6 .. 1.8e-6 against 9e-5 .. 1.1e-4
    assert abs(float(loss) - float(loss_ref)) < 5e-5
    worst = 0.0
    for k, p in m.named_parameters():
        ref = g_ref[k]
        scale = max(float(ref.abs().max()), 1e-12)
        # identically zero in exact arithmetic (softmax shift invariance; for '|' in eval mode also q_lin): absolute
        if k == "attention.x_conv.bias" or (do_option == "|" and not train and k.startswith("attention.q_lin")):
            assert float(p.grad.abs().max()) < 1e-6
            continue
        e = float((p.grad.cpu() - ref).abs().max()) / scale
        d = float((g_f32[k] - ref).abs().max()) / scale
        print(f"[parity-bf16] module ({do_option}, train={train}) grad {k}: vs bf16 oracle {e:.3e}; bf16 vs fp32 oracle {d:.3e}")
        worst = max(worst, e)
        assert e < 2e-2, (k, e)
    print(f"[parity-bf16] worst gradient error {worst:.3e}")


@pytest.mark.parametrize("lstm16", ["1", "0"])
def test_bf16_module_lstm_side_products_match_bf16_oracle(lstm16, monkeypatch):
    """T * B a multiple of 8 (here 4 x 6): the LSTM's non-recurrent products (xg = x . W_ih^T, dW_hh, dW_ih, dx) run on bf16
    MFMA with the embedding width padded from 20 to 24 (engine.py, VQA_LSTM16); the oracle rounds the same operands
    (vqa_oracle._HhProduct, lstm16_ok).  The other module tests have T * B = 12 / 18 / 28 and keep these products in fp32.
    Train mode with shared masks, both settings of the switch, the tolerances of test_bf16_module_matches_bf16_oracle."""
    from oracle import vqa_oracle as O
    from dl_vqa_amd import VqaNet
    from dl_vqa_amd.train import soft_ce_loss_and_score
    from tests.hip_masks import hip_masks
    monkeypatch.setenv("VQA_LSTM16", lstm16)
    cfg = bf16_cfg("+", 0.3)
    V, B, S, T = 50, 4, 48, 6
    torch.manual_seed(5)
    m = VqaNet(cfg, V, compute_dtype="bf16").to(DEV).train()
    sd = {k: t.detach().cpu().clone() for k, t in m.state_dict().items()}
    v, q, a_idx, a_val, _, _, ql = O.synthetic_batch(B, S, T, V, 24, seed=12)
    torch.manual_seed(9)
    y = m(v.to(DEV), q.to(DEV), ql.to(DEV))
    loss, _ = soft_ce_loss_and_score(y, a_idx.to(DEV), a_val.to(DEV))
    loss.backward()
    torch.cuda.synchronize()
    ctx = m._last_ctx
    assert (ctx.x16 is not None) == (lstm16 == "1")
    if lstm16 == "1":
        assert ctx.x16.dtype == torch.bfloat16 and tuple(ctx.x16.shape) == (T * B, 24)
        assert float(ctx.x16[:, 20:].float().abs().max()) == 0.0
    masks = hip_masks(m._engine, ctx.seed, B, T, ctx.acts[-1].shape[1], DEV)
    assert O.lstm16_ok(32, T * B) == (lstm16 == "1")
    y_ref, loss_ref, g_ref = O.loss_and_grads(sd, cfg, v, q, ql, a_idx, a_val, masks=masks, bf16=True)
    y_f32, _, g_f32 = O.loss_and_grads(sd, cfg, v, q, ql, a_idx, a_val, masks=masks)
    err = float((y.detach().cpu() - y_ref).abs().max())
    dist = float((y_f32 - y_ref).abs().max())
    print(f"[parity-bf16] module (lstm16={lstm16}) logits |err| vs bf16 oracle {err:.3e}; bf16 vs fp32 oracle {dist:.3e}")
    assert err < 5e-5 and err < 0.5 * dist
    assert abs(float(loss) - float(loss_ref)) < 5e-5
    for k, p in m.named_parameters():
        ref = g_ref[k]
        scale = max(float(ref.abs().max()), 1e-12)
        if k == "attention.x_conv.bias":
            assert float(p.grad.abs().max()) < 1e-6
            continue
        e = float((p.grad.cpu() - ref).abs().max()) / scale
        d = float((g_f32[k] - ref).abs().max()) / scale
        print(f"[parity-bf16] module (lstm16={lstm16}) grad {k}: vs bf16 oracle {e:.3e}; bf16 vs fp32 oracle {d:.3e}")
        assert e < 2e-2, (k, e)
        if k.startswith("text.lstm.weight") or k == "text.embedding.weight":
            assert e < 2e-3, (k, e)      # the products this test is about: far inside the common bound


@pytest.mark.parametrize("pconv", ["1", "0"])
def test_bf16_module_patch_conv_path_matches_bf16_oracle(pconv, monkeypatch):
    """The bf16 module on the reference's channel counts 3/64/128/256 (small image), where blocks 1.. run on the patch
    convolutions (C16 activations, materialised pre-pool gradient), against the bf16 oracle -- and the same with
    VQA_PCONV=0 (implicit-GEMM kernels): both paths must meet the same tolerances."""
    from oracle import vqa_oracle as O
    from dl_vqa_amd import VqaNet
    from dl_vqa_amd.train import soft_ce_loss_and_score
    from tests.hip_masks import hip_masks
    monkeypatch.setenv("VQA_PCONV", pconv)
    cfg = bf16_cfg("+", 0.3)
    cfg["image"]["num_channels"] = [3, 64, 128, 256]
    V, B, S, T = 50, 2, 68, 6              # grid 6 x 6: B * P = 72 rows (the bf16 GEMMs want multiples of 8)
    torch.manual_seed(4)
    m = VqaNet(cfg, V, compute_dtype="bf16").to(DEV).train()
    sd = {k: t.detach().cpu().clone() for k, t in m.state_dict().items()}
    v, q, a_idx, a_val, _, _, ql = O.synthetic_batch(B, S, T, V, 24, seed=8)
    torch.manual_seed(9)
    y = m(v.to(DEV), q.to(DEV), ql.to(DEV))
    loss, _ = soft_ce_loss_and_score(y, a_idx.to(DEV), a_val.to(DEV))
    loss.backward()
    torch.cuda.synchronize()
    ctx = m._last_ctx
    assert ctx.use_pc == (pconv == "1") and (ctx.acts[1].dim() == 5) == ctx.use_pc
    masks = hip_masks(m._engine, ctx.seed, B, T, ctx.acts[-1].shape[1], DEV)
    y_ref, loss_ref, g_ref = O.loss_and_grads(sd, cfg, v, q, ql, a_idx, a_val, masks=masks, bf16=True)
    y_f32, _, g_f32 = O.loss_and_grads(sd, cfg, v, q, ql, a_idx, a_val, masks=masks)
    err = float((y.detach().cpu() - y_ref).abs().max())
    dist = float((y_f32 - y_ref).abs().max())
    print(f"[parity-bf16] module (pconv={pconv}) logits |err| vs bf16 oracle {err:.3e}; bf16 vs fp32 oracle {dist:.3e}")
    assert err < 5e-5 and err < 0.5 * dist
    assert abs(float(loss) - float(loss_ref)) < 5e-5
    for k, p in m.named_parameters():
        ref = g_ref[k]
        scale = max(float(ref.abs().max()), 1e-12)
        if k == "attention.x_conv.bias":
            assert float(p.grad.abs().max()) < 1e-6
            continue
        e = float((p.grad.cpu() - ref).abs().max()) / scale
        print(f"[parity-bf16] module (pconv={pconv}) grad {k}: vs bf16 oracle {e:.3e}")
        assert e < 2e-2, (k, e)


def test_bf16_four_block_448_matches_bf16_oracle():
    """A DEEPER network than the one benchmarked (448x448 images, FOUR conv blocks 64/128/256/512 -> a 26x26 grid,
    Hq=Ha=Hc=1024, T=14, A=1000) at B=2 in train mode with shared dropout masks: the bf16 kernels at a 512-channel block
    and K up to 4.6e3.  (configs[3]'s own architecture -- the reference's three blocks, 54x54 grid -- is
    test_bf16_configs3_bench_architecture below.)  Tolerances as in test_bf16_module_matches_bf16_oracle, wider for the depth (four rounding
    points per path instead of three, K up to 4.6e3 per conv output): logits 5e-4 absolute and at most half the
    bf16-vs-fp32 distance; gradients 5e-2 of the largest entry."""
    from oracle import vqa_oracle as O
    from dl_vqa_amd import VqaNet
    from dl_vqa_amd.train import soft_ce_loss_and_score
    from tests.hip_masks import hip_masks
    from tests.golden_util import full_cfg
    cfg = full_cfg(1000)
    cfg["image"]["num_channels"] = [3, 64, 128, 256, 512]
    V, B, S, T = 3000, 2, 448, 14
    torch.manual_seed(11)
    m = VqaNet(cfg, V, compute_dtype="bf16").to(DEV).train()
    sd = {k: t.detach().cpu().clone() for k, t in m.state_dict().items()}
    v, q, a_idx, a_val, _, _, ql = O.synthetic_batch(B, S, T, V, 1000, seed=6)
    y = m(v.to(DEV), q.to(DEV), ql.to(DEV))
    loss, _ = soft_ce_loss_and_score(y, a_idx.to(DEV), a_val.to(DEV))
    loss.backward()
    torch.cuda.synchronize()
    ctx = m._last_ctx
    masks = hip_masks(m._engine, ctx.seed, B, T, ctx.acts[-1].shape[1], DEV)
    y_ref, loss_ref, g_ref = O.loss_and_grads(sd, cfg, v, q, ql, a_idx, a_val, masks=masks, bf16=True)
    y_f32, _, g_f32 = O.loss_and_grads(sd, cfg, v, q, ql, a_idx, a_val, masks=masks)
    err = float((y.detach().cpu() - y_ref).abs().max())
    dist = float((y_f32 - y_ref).abs().max())
    print(f"[parity-bf16] configs[3] architecture logits |err| vs bf16 oracle {err:.3e}; bf16 vs fp32 oracle {dist:.3e}")
    assert err < 5e-4 and err < 0.5 * dist
    assert abs(float(loss) - float(loss_ref)) < 5e-4
    for k, p in m.named_parameters():
        ref = g_ref[k]
        scale = max(float(ref.abs().max()), 1e-12)
        if k == "attention.x_conv.bias":
            assert float(p.grad.abs().max()) < 1e-6
            continue
        e = float((p.grad.cpu() - ref).abs().max()) / scale
        d = float((g_f32[k] - ref).abs().max()) / scale
        print(f"[parity-bf16] configs[3] architecture grad {k}: vs bf16 oracle {e:.3e}; bf16 vs fp32 oracle {d:.3e}")
        assert e < 5e-2, (k, e)


def test_bf16_configs3_bench_architecture():
    """BASELINE.json configs[3] as `bench.py --dtype bf16 --size 448` runs it: the reference's THREE conv blocks
    3/64/128/256 (config.yaml:60) on 448x448 images -> 54x54 grid, P = 2 916 positions, Hq=Ha=Hc=1024, T=14, A=1000;
    B=2, train mode with shared dropout masks.  These are the shapes of profiles/r0x_bf16_448_kernel_stats.txt: the
    bf16 attention stage (gemm_bf16 row-group epilogue, att_score_*<2,true>, l2norm_fwd<true>) at P = 2 916 and the conv
    kernels at 223 / 110-pixel maps (VERDICT r2 'weak' 1).
    Parity for this row is UNPINNED by the reference (it has no bf16 path): the bf16 oracle rounds where the HIP path
    rounds, so agreement shows 'same algorithm up to accumulation order'; the external anchor is the second
    assertion -- the HIP result sits closer to the bf16 oracle than half the bf16-to-fp32 distance, and that distance
    itself stays small against the logits' scale."""
    from oracle import vqa_oracle as O
    from dl_vqa_amd import VqaNet
    from dl_vqa_amd.train import soft_ce_loss_and_score
    from tests.hip_masks import hip_masks
    from tests.golden_util import full_cfg
    cfg = full_cfg(1000)
    assert cfg["image"]["num_channels"] == [3, 64, 128, 256]
    V, B, S, T = 3000, 2, 448, 14
    torch.manual_seed(12)
    m = VqaNet(cfg, V, compute_dtype="bf16").to(DEV).train()
    sd = {k: t.detach().cpu().clone() for k, t in m.state_dict().items()}
    v, q, a_idx, a_val, _, _, ql = O.synthetic_batch(B, S, T, V, 1000, seed=7)
    y = m(v.to(DEV), q.to(DEV), ql.to(DEV))
    loss, _ = soft_ce_loss_and_score(y, a_idx.to(DEV), a_val.to(DEV))
    loss.backward()
    torch.cuda.synchronize()
    ctx = m._last_ctx
    hw = lambda t: tuple(t.shape[2:4]) if t.dim() == 5 else tuple(t.shape[1:3])      # C16 [B,C/16,H,W,16] or NHWC
    assert ctx.Pn == 2916 and hw(ctx.acts[1]) == (223, 223) and hw(ctx.acts[2]) == (110, 110)
    masks = hip_masks(m._engine, ctx.seed, B, T, ctx.acts[-1].shape[1], DEV)
    y_ref, loss_ref, g_ref = O.loss_and_grads(sd, cfg, v, q, ql, a_idx, a_val, masks=masks, bf16=True)
    y_f32, _, g_f32 = O.loss_and_grads(sd, cfg, v, q, ql, a_idx, a_val, masks=masks)
    err = float((y.detach().cpu() - y_ref).abs().max())
    dist = float((y_f32 - y_ref).abs().max())
    print(f"[parity-bf16] configs[3] bench architecture (P=2916) logits |err| vs bf16 oracle {err:.3e}; bf16 vs fp32 oracle {dist:.3e}; "
          f"logits scale {float(y_f32.abs().max()):.3e}")
    assert err < 5e-4 and err < 0.5 * dist
    assert dist < 2e-2 * max(float(y_f32.abs().max()), 1e-3)
    assert abs(float(loss) - float(loss_ref)) < 5e-4
    for k, p in m.named_parameters():
        ref = g_ref[k]
        scale = max(float(ref.abs().max()), 1e-12)
        if k == "attention.x_conv.bias":
            assert float(p.grad.abs().max()) < 1e-6
            continue
        e = float((p.grad.cpu() - ref).abs().max()) / scale
        d = float((g_f32[k] - ref).abs().max()) / scale
        print(f"[parity-bf16] configs[3] bench architecture grad {k}: vs bf16 oracle {e:.3e}; bf16 vs fp32 oracle {d:.3e}")
        assert e < 5e-2, (k, e)


def test_conv0_bf16_mfma_forward():
    """First block on bf16 MFMA (image and weights rounded to bf16, fp32 accumulate) against float64 on rounded inputs,
    incl. a 448-wide image (> 64 KB of LDS)."""
    ops = _ops()
    for (B, H, W, Co) in ((2, 30, 32, 64), (1, 22, 448, 64), (3, 17, 20, 32), (7, 50, 132, 64), (40, 26, 24, 64)):
        g = torch.Generator().manual_seed(H + W)
        x = torch.randn(B, 3, H, W, generator=g)
        w = torch.randn(Co, 3, 3, 3, generator=g) * 0.2
        b = torch.randn(Co, generator=g) * 0.1
        ref = F.max_pool2d(torch.relu(F.conv2d(rb(x).double(), rb(w).double(), b.double())), 2, 2)
        p16, am = ops.conv0_fwd(x.to(DEV), w.to(DEV), b.to(DEV), out_dtype=torch.bfloat16, bf16_mfma=True)
        torch.cuda.synchronize()
        check(f"conv0 bf16-MFMA fwd {B,H,W,Co}", p16.float().permute(0, 3, 1, 2), ref, 2 ** -8)
        assert bool(((p16 == 0) == (am == 4)).all())
        # the persistent channel-blocked form (what the patch convolutions read), float and __half images (the dataset's
        # features, widened exactly): the same bits
        xh = x.half()
        ph, amh = ops.conv0_fwd(xh.float().to(DEV), w.to(DEV), b.to(DEV), out_dtype=torch.bfloat16, bf16_mfma=True)
        for xin in (xh.float(), xh):
            pc, amc = ops.conv0_fwd(xin.to(DEV), w.to(DEV), b.to(DEV), out_dtype=torch.bfloat16, bf16_mfma=True, out_c16=True)
            torch.cuda.synchronize()
            assert torch.equal(ops.from_c16(pc), ph) and torch.equal(amc, amh), (B, H, W, Co, xin.dtype)


def test_conv0_bf16_mfma_wgrad():
    """First block's weight / bias gradient on bf16 MFMA (image rounded to bf16, bf16 pooled gradient) against float64
    autograd on the rounded operands; odd sizes (pixel groups with padding), a 448-wide image, several images."""
    ops = _ops()
    for (B, H, W, Co) in ((2, 30, 32, 64), (1, 14, 448, 64), (3, 17, 20, 32), (5, 40, 44, 64), (300, 12, 16, 64)):
        g = torch.Generator().manual_seed(H * 3 + W)
        x = torch.randn(B, 3, H, W, generator=g)
        w = torch.randn(Co, 3, 3, 3, generator=g) * 0.2
        b = torch.randn(Co, generator=g) * 0.1
        wr, br = rb(w).double().requires_grad_(True), b.double().requires_grad_(True)
        yr = F.max_pool2d(torch.relu(F.conv2d(rb(x).double(), wr, br)), 2, 2)
        dy = rb(torch.randn(yr.shape, generator=g))
        yr.backward(dy.double())
        p16, am = ops.conv0_fwd(x.to(DEV), w.to(DEV), b.to(DEV), out_dtype=torch.bfloat16, bf16_mfma=True)
        dw, db = torch.empty(Co, 3, 3, 3, device=DEV), torch.empty(Co, device=DEV)
        ops.conv0_wgrad_bf16(x.to(DEV), dy.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(DEV), am, dw, db)
        torch.cuda.synchronize()
        check(f"conv0 bf16-MFMA wgrad {B,H,W,Co}", dw, wr.grad, 3e-5)
        check(f"conv0 bf16-MFMA bias grad {B,H,W,Co}", db, br.grad, 3e-5)
        # __half image: the same bits as its widened copy
        xh = x.half()
        dyd = dy.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(DEV)
        dw1, db1, dw2, db2 = (torch.empty_like(dw), torch.empty_like(db), torch.empty_like(dw), torch.empty_like(db))
        ops.conv0_wgrad_bf16(xh.float().to(DEV), dyd, am, dw1, db1)
        ops.conv0_wgrad_bf16(xh.to(DEV), dyd, am, dw2, db2)
        torch.cuda.synchronize()
        assert torch.equal(dw1, dw2) and torch.equal(db1, db2)


def test_bf16_path_rejects_unsupported_configs():
    from dl_vqa_amd import VqaNet
    cfg = bf16_cfg()
    cfg["image"]["num_channels"] = [3, 8, 16, 32]
    with pytest.raises(ValueError, match="multiples of 64"):
        VqaNet(cfg, 10, compute_dtype="bf16")
    with pytest.raises(ValueError, match="compute_dtype"):
        VqaNet(bf16_cfg(), 10, compute_dtype="fp8")


# ----------------------------------------------------------------------------- patch convolutions (LDS-resident input patch)
@pytest.mark.parametrize("B,H,W,Ci,Co", [(2, 30, 30, 64, 128), (1, 30, 34, 128, 256), (3, 22, 20, 64, 64), (2, 58, 58, 64, 128),
                                         (1, 70, 45, 16, 64), (2, 37, 75, 128, 128), (5, 18, 100, 32, 256)])
def test_pconv_fwd_dgrad_wgrad(B, H, W, Ci, Co):
    """csrc/conv_patch_bf16.hip against float64 autograd on bf16-rounded x, w, dy: forward (fp32 NHWC and bf16 C16 outputs,
    arg-max C16), backward-data with the pre-pool gradient routed in the kernel (fp32 / bf16 NHWC and bf16 C16 outputs: zero
    border, dropped pool rows / columns, dead windows), weight + bias gradient.  Shapes with tile overhang on both axes, odd sizes (dropped pool row / column),
    one and several 64-/128-channel slabs and roles, several images (persistent tile streams that cross image borders)."""
    ops = _ops()
    assert ops.pconv_supported(H, W, Ci, Co)
    g = torch.Generator().manual_seed(B * 1000 + H * 10 + Ci)
    x = rb(torch.randn(B, Ci, H, W, generator=g))
    w = rb(torch.randn(Co, Ci, 3, 3, generator=g) / math.sqrt(9 * Ci))
    b = torch.randn(Co, generator=g) * 0.1
    xr, wr, br = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    yr = F.max_pool2d(torch.relu(F.conv2d(xr, wr, br)), 2, 2)
    dy = rb(torch.randn(yr.shape, generator=g))
    yr.backward(dy.double())

    xd = x.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(DEV)        # NHWC
    xc = ops.to_c16(xd)                                                        # what the patch kernels read
    assert torch.equal(ops.from_c16(xc), xd)
    wf, wd = ops.pconv_pack_weights(w.to(DEV), need_wd=Ci % 64 == 0)
    p32, amax16 = ops.pconv_fwd(xc, wf, b.to(DEV), Co, out_dtype=torch.float32)
    p16, amax2 = ops.pconv_fwd(xc, wf, b.to(DEV), Co)
    torch.cuda.synchronize()
    tag = f"{B,H,W,Ci,Co}"
    Hp, Wp = yr.shape[2], yr.shape[3]
    assert tuple(amax16.shape) == (B, Co // 16, Hp, Wp, 16)              # arg-max bytes are channel-blocked like the activations
    amax = ops.from_c16(amax16)
    check(f"pconv fwd fp32-out {tag}", p32.permute(0, 3, 1, 2), yr, 3e-6 * math.sqrt(9 * Ci))
    assert torch.equal(amax16, amax2) and torch.equal(ops.from_c16(p16), p32.to(torch.bfloat16))
    assert bool(((p32 == 0) == (amax == 4)).all())
    # the same result as the implicit-GEMM kernel up to fp32 summation order; identical arg-max wherever the winner is clear
    if Ci % 64 == 0:
        wfT, _ = ops.conv_pack_weights_bf16(w.to(DEV), Ci, need_wd=False)
        q32, amax_ig = ops.conv_fwd_bf16(xd, wfT, b.to(DEV), 1, out_dtype=torch.float32)
        torch.cuda.synchronize()
        assert float((q32 - p32).abs().max()) <= 1e-5 * max(1.0, float(q32.abs().max()))
        assert float((amax_ig != amax).float().mean()) < 1e-3

    # the pre-pool gradient the backward kernels route for themselves: dY[b, y, x, c] = dP[b, y//2, x//2, c] iff
    # argmax == (y%2)*2 + x%2 -- built here with the KERNEL's arg-max (a float64 near-tie may pick another pixel of a window)
    dyd = dy.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(DEV)
    dpc = ops.to_c16(dyd)
    ref = torch.zeros(B, H - 2, W - 2, Co, dtype=torch.bfloat16, device=DEV)
    for j in range(4):
        sel = torch.where(amax == j, dyd, torch.zeros_like(dyd))
        ref[:, (j >> 1):2 * Hp:2, (j & 1):2 * Wp:2, :] = sel
    dyfull = ref.float().permute(0, 3, 1, 2).double().cpu()
    if Ci % 64 == 0:
        dx32 = ops.pconv_dgrad(dpc, amax16, wd, xd.shape, out_dtype=torch.float32)
        dx16 = ops.pconv_dgrad(dpc, amax16, wd, xd.shape)
        dxc = ops.pconv_dgrad(dpc, amax16, wd, xd.shape, out_c16=True)
        torch.cuda.synchronize()
        dx_ref = torch.nn.grad.conv2d_input(xr.shape, wr.detach(), dyfull)
        check(f"pconv dgrad fp32-out {tag}", dx32.permute(0, 3, 1, 2), dx_ref, 5e-6 * math.sqrt(9 * Co))
        assert torch.equal(dx16, dx32.to(torch.bfloat16))
        assert tuple(dxc.shape) == (B, Ci // 16, H, W, 16) and torch.equal(ops.from_c16(dxc), dx16)
    if ops.pconv_wgrad_supported(H, W, Ci, Co):
        dw, db = torch.empty(Co, Ci, 3, 3, device=DEV), torch.empty(Co, device=DEV)
        ops.pconv_wgrad(xc, dpc, amax16, dw, db)
        torch.cuda.synchronize()
        dw_ref = torch.nn.grad.conv2d_weight(xr.detach(), wr.shape, dyfull)
        check(f"pconv wgrad {tag}", dw, dw_ref, 2e-5)
        check(f"pconv bias grad {tag}", db, dyfull.sum(dim=(0, 2, 3)), 2e-5)
