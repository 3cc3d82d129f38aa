"""fp32 on the bf16 matrix cores (csrc/x3_core.hpp: exact three-way bf16 operand split, six partial products, fp32
accumulate) against float64 references, beside the native fp32 MFMA kernels on the same inputs: the split path must be
as accurate as the fp32 MFMA path (its dropped partial products are below one fp32 rounding of a product)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ops():
    from dl_vqa_amd import ops
    return ops


def rel_err(got, ref):
    ref = ref.double().cpu()
    return float((got.double().cpu() - ref).abs().max() / ref.abs().max().clamp_min(1e-30))


def _nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def test_split_is_exact():
    """x == hi + mid + lo exactly (float64 sum of the three bf16 terms), hi = RNE(x), |mid| <= 2^-8 |hi|, |lo| <= 2^-16 |hi|:
    random values over 60 binades, powers of two, values that round up to the next binade, zeros, both signs."""
    ops = _ops()
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1 << 16, generator=g) * torch.exp2(torch.randint(-30, 30, (1 << 16,), generator=g).float())
    special = torch.tensor([0.0, -0.0, 1.0, -1.0, 1.9999999, 0.99999994, 3.0e38, -3.0e38, 1e-30, 255.99998, 65535.996, 1.00390625])
    x[:special.numel()] = special
    hi, mid, lo = ops.x3_split(x.to(DEV)).cpu()
    torch.cuda.synchronize()
    assert torch.equal(hi, x.to(torch.bfloat16))
    assert torch.equal(hi.double() + mid.double() + lo.double(), x.double())
    assert bool((mid.double().abs() <= hi.double().abs() * 2.0 ** -8).all())
    assert bool((lo.double().abs() <= hi.double().abs() * 2.0 ** -16).all())


@pytest.fixture
def knob(monkeypatch):
    """Set a VQA_* knob for one test (the library reads them once: vqa_reload_knobs after every change)."""
    from dl_vqa_amd import _lib

    def set_knob(name, value):
        monkeypatch.setenv(name, value)
        _lib.load().vqa_reload_knobs()
    yield set_knob
    monkeypatch.undo()
    _lib.load().vqa_reload_knobs()


X3_CASES = [  # B, H, W, Ci, Co, stride
    (2, 58, 58, 64, 128, 1),   # conv1 family: 192x128 tiles forward / wgrad, 256x64 dgrad
    (1, 38, 38, 128, 256, 1),  # conv2 family
    (3, 34, 70, 64, 64, 1),    # Co = 64: the 256x64 forward tile; wide rows (row cursor wraps rows and images)
    (2, 38, 42, 32, 96, 1),    # partial N tile (Co = 96), one K-step per tap
    (2, 69, 73, 32, 64, 2),    # stride 2
    (5, 36, 36, 96, 160, 1),   # three K-steps per tap, 1.25 N tiles, rows % 192 != 0
    (4, 122, 122, 96, 128, 1), # > 256 tiles: the persistent forward / dgrad kernels (tile seams, uneven tile counts)
]


@pytest.mark.parametrize("B,H,W,Ci,Co,stride", X3_CASES)
def test_conv_x3_matches_float64_like_native_fp32(B, H, W, Ci, Co, stride):
    ops = _ops()
    assert ops.conv_x3_supported(H, W, Ci, Co, stride)
    g = torch.Generator().manual_seed(B * 1000 + H * 10 + Ci)
    x = torch.randn(B, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, 3, 3, generator=g) / math.sqrt(9 * Ci)
    b = torch.randn(Co, generator=g) * 0.1
    xr, wr, br = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    yr = F.max_pool2d(torch.relu(F.conv2d(xr, wr, br, stride=stride)), 2, 2)
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy.double())

    xd = _nhwc(x).to(DEV)
    wf, wd = ops.conv_pack_weights(w.to(DEV), Ci)
    wfx, wdx = ops.x3_split(wf), ops.x3_split(wd)
    dyd = _nhwc(dy).to(DEV)
    tag = f"{B,H,W,Ci,Co,stride}"
    res = {}
    xp = ops.x3_pack(xd)
    # the x3-packed input form (activations split once per tensor) must give bit-identical results to the fp32 input form
    pooled_p, amax_p = ops.conv_fwd(xp, wfx, b.to(DEV), stride, x3=True)
    for x3 in (False, True):
        pooled, amax = ops.conv_fwd(xd, wfx if x3 else wf, b.to(DEV), stride, x3=x3)
        # the backward kernels of both paths get the SAME arg-max bytes (a pre-activation tie broken differently by
        # rounding would compare different functions)
        am = amax if not x3 else res[False][1]
        dx = ops.conv_dgrad(dyd, am, wdx if x3 else wd, xd.shape, stride, x3=x3)
        dw, db = torch.empty(Co, Ci, 3, 3, device=DEV), torch.empty(Co, device=DEV)
        ops.conv_wgrad(xd, dyd, am, dw, db, stride, x3=x3)
        torch.cuda.synchronize()
        res[x3] = (pooled, amax, dx, dw, db)
        if x3:
            dw_p, db_p = torch.empty_like(dw), torch.empty_like(db)
            ops.conv_wgrad(xp, dyd, am, dw_p, db_p, stride, x3=True)
            torch.cuda.synchronize()
            assert torch.equal(pooled_p, pooled) and torch.equal(amax_p, amax)
            assert torch.equal(dw_p, dw) and torch.equal(db_p, db)
            # x3-packed pooled gradient (routing as a mask on the packed halves): bit-identical again
            dyp = ops.x3_pack(dyd)
            db_s = torch.empty_like(db)
            assert torch.equal(ops.x3_pack_pooled_grad(dyd, am, db_s), dyp)        # pack + bias gradient in one pass
            dw_s = torch.empty_like(dw)
            ops.conv_wgrad(xp, dyd, am, dw_s, None, stride, x3=True, dpooled_packed=dyp)   # dbias None: not recomputed
            torch.cuda.synchronize()
            assert torch.equal(db_s, db) and torch.equal(dw_s, dw)
            dx_p = ops.conv_dgrad(dyp, am, wdx, xd.shape, stride, x3=True)
            dw_q, db_q = torch.empty_like(dw), torch.empty_like(db)
            ops.conv_wgrad(xp, dyd, am, dw_q, db_q, stride, x3=True, dpooled_packed=dyp)
            dw_r, db_r = torch.empty_like(dw), torch.empty_like(db)
            ops.conv_wgrad(xd, dyd, am, dw_r, db_r, stride, x3=True, dpooled_packed=dyp)
            torch.cuda.synchronize()
            assert torch.equal(dx_p, dx)
            assert torch.equal(dw_q, dw) and torch.equal(db_q, db) and torch.equal(dw_r, dw)
            # output written x3-packed by the epilogue == vqa_x3_pack of the fp32 output, bit for bit
            pooled_q, amax_q = ops.conv_fwd(xp, wfx, b.to(DEV), stride, x3=True, out_packed=True)
            torch.cuda.synchronize()
            assert torch.equal(pooled_q, ops.x3_pack(pooled)) and torch.equal(amax_q, amax)
    refs = (yr, None, xr.grad, wr.grad, br.grad)
    names = ("fwd", None, "dgrad", "wgrad", "bias grad")
    tols = (3e-6 * math.sqrt(9 * Ci), None, 5e-6 * math.sqrt(9 * Co), 2e-5, 2e-5)
    for k in (0, 2, 3, 4):
        got_n = res[False][k].permute(0, 3, 1, 2) if k in (0, 2) else res[False][k]
        got_x = res[True][k].permute(0, 3, 1, 2) if k in (0, 2) else res[True][k]
        en, ex = rel_err(got_n, refs[k]), rel_err(got_x, refs[k])
        print(f"[parity-x3] conv {names[k]} {tag}: fp32-MFMA err {en:.3e}, 3xbf16 err {ex:.3e} (tolerance {tols[k]:.1e})")
        assert ex < tols[k], (names[k], ex)
        assert ex < 2.0 * en + 2e-7, (names[k], ex, en)     # as accurate as the fp32 MFMA path
    # arg-max bytes: identical except where two pre-activations of a window tie to within rounding
    diff = (res[False][1] != res[True][1]).float().mean().item()
    print(f"[parity-x3] conv arg-max bytes differing {tag}: {diff:.2e}")
    assert diff < 1e-4
    assert bool(((res[True][0] == 0) == (res[True][1] == 4)).all())


def test_pack_layout_and_first_block_packed_output():
    """x3_pack: every four channels as hi[4] mid[4] lo[4] (== the planes of x3_split, regrouped); the dedicated first
    block writing that form directly == x3_pack of its fp32 output."""
    ops = _ops()
    g = torch.Generator().manual_seed(4)
    x = torch.randn(3, 5, 7, 64, generator=g).to(DEV)
    planes = ops.x3_split(x)                                   # [3, 3, 5, 7, 64]
    want = planes.view(3, 3, 5, 7, 16, 4).permute(1, 2, 3, 4, 0, 5).contiguous()
    assert torch.equal(ops.x3_pack(x), want)
    for (B, H, W, Co) in ((2, 30, 32, 64), (1, 22, 448, 64), (3, 18, 20, 32)):
        img = torch.randn(B, 3, H, W, generator=g).to(DEV)
        w = (torch.randn(Co, 3, 3, 3, generator=g) * 0.2).to(DEV)
        b = (torch.randn(Co, generator=g) * 0.1).to(DEV)
        p32, am = ops.conv0_fwd(img, w, b)
        pp, am2 = ops.conv0_fwd(img, w, b, out_packed=True)
        torch.cuda.synchronize()
        assert torch.equal(pp, ops.x3_pack(p32)) and torch.equal(am, am2)


def test_fp32x3_module_uses_packed_activations_and_matches_fp32_module():
    """VqaNet(compute_dtype="fp32x3") on the north-star architecture (B=3, train mode): conv blocks 1-2 run on the split
    kernels with x3-packed inputs; logits and every gradient agree with the fp32-MFMA module to fp32 rounding."""
    from dl_vqa_amd import VqaNet
    from dl_vqa_amd.train import soft_ce_loss_and_score
    from oracle import vqa_oracle as O
    from tests.golden_util import full_cfg
    cfg = full_cfg(1000)
    V, B, S, T = 500, 3, 224, 14
    v, q, a_idx, a_val, _, _, ql = O.synthetic_batch(B, S, T, V, 1000, seed=8)
    out = {}
    for dt in ("fp32", "fp32x3"):
        torch.manual_seed(21)
        m = VqaNet(cfg, V, compute_dtype=dt).to(DEV).train()
        torch.manual_seed(5)
        y = m(v.to(DEV), q.to(DEV), ql.to(DEV))
        loss, _ = soft_ce_loss_and_score(y, a_idx.to(DEV), a_val.to(DEV))
        loss.backward()
        torch.cuda.synchronize()
        out[dt] = (y.detach(), {k: p.grad.clone() for k, p in m.named_parameters()}, m._last_ctx)
    acts = out["fp32x3"][2].acts
    assert acts[1].dim() == 6 and acts[2].dim() == 6 and acts[3].dim() == 4 and acts[3].dtype == torch.float32
    assert out["fp32"][2].seed == out["fp32x3"][2].seed                       # the same dropout masks
    err = float((out["fp32"][0] - out["fp32x3"][0]).abs().max())
    print(f"[parity-x3] module logits fp32 vs fp32x3: {err:.3e}")
    assert err < 1e-5
    for k, gx in out["fp32x3"][1].items():
        gn = out["fp32"][1][k]
        if k == "attention.x_conv.bias":          # identically zero in exact arithmetic (softmax shift invariance)
            assert float(gx.abs().max()) < 1e-6
            continue
        e = float((gx - gn).abs().max()) / max(float(gn.abs().max()), 1e-12)
        print(f"[parity-x3] module grad {k}: {e:.3e}")
        # 1e-3 as test_train_mode_full224_matches_oracle_with_shared_masks: the first block's weight gradient sums 1e7
        # products per element behind two arg-max routings, where a tie broken the other way moves it by ~1e-4
        assert e < 1e-3, (k, e)


@pytest.mark.parametrize("M,N,K", [(1352, 1024, 256), (1024, 256, 5000), (700, 256, 1024), (200, 130, 70),
                                   (20000, 520, 96)])     # 105 x 5 tiles: the persistent kernel, tails in M and N
@pytest.mark.parametrize("transA,transB", [(False, True), (False, False), (True, True), (True, False)])
def test_gemm_x3_layouts(M, N, K, transA, transB):
    """vqa_gemm_x3 in the four operand layouts (tails in M, N, K; split-K slabs for the few-tile shapes) against
    float64, beside vqa_gemm on the same operands."""
    ops = _ops()
    g = torch.Generator().manual_seed(M + N + K)
    lda = (M if transA else K) + 3 & ~3
    ldb = (K if transB else N) + 3 & ~3
    A = torch.randn((K, lda) if transA else (M, lda), generator=g)
    Bm = torch.randn((N, ldb) if transB else (K, ldb), generator=g)
    A2 = (A[:, :M].t() if transA else A[:, :K]).double()
    B2 = (Bm[:, :K].t() if transB else Bm[:, :N]).double()
    ref = A2 @ B2
    out = {}
    for x3 in (False, True):
        C = torch.empty(M, N, device=DEV)
        ops.gemm(A.to(DEV), Bm.to(DEV), C, M, N, K, transA=transA, transB=transB, lda=lda, ldb=ldb, x3=x3)
        torch.cuda.synchronize()
        out[x3] = rel_err(C, ref)
    print(f"[parity-x3] gemm {M}x{N}x{K} tA={transA} tB={transB}: fp32-MFMA err {out[False]:.3e}, 3xbf16 err {out[True]:.3e}")
    assert out[True] < 2e-6 * math.sqrt(K / 256 + 1) and out[True] < 2.0 * out[False] + 2e-7


def test_gemm_x3_epilogue():
    """The v_conv forward form: row-group term tiled over positions (add / multiply), bias, ReLU, raw-product output,
    accumulate -- the fp32 engine's epilogue on the split kernel's accumulators."""
    ops = _ops()
    g = torch.Generator().manual_seed(12)
    Bn, Pn, C, mid = 3, 676, 256, 384
    M = Bn * Pn
    A, W = torch.randn(M, C, generator=g), torch.randn(mid, C, generator=g) * 0.1
    qp, bias = torch.randn(Bn, mid, generator=g), torch.randn(mid, generator=g)
    raw = A.double() @ W.double().t()
    for op in (0, 1):
        tiled = qp.double().repeat_interleave(Pn, dim=0)
        ref = torch.relu((raw * tiled if op else raw + tiled) + bias.double())
        C1, aux = torch.empty(M, mid, device=DEV), torch.empty(M, mid, device=DEV)
        ops.gemm(A.to(DEV), W.to(DEV), C1, M, mid, C, rowgroup=qp.to(DEV), rg_div=Pn, rg_op=op, bias1=bias.to(DEV), relu=True,
                 aux=aux, x3=True)
        torch.cuda.synchronize()
        assert rel_err(C1, ref) < 3e-6 and rel_err(aux, raw) < 3e-6
    acc0 = torch.randn(M, mid, generator=g)
    C2 = acc0.clone().to(DEV)
    ops.gemm(A.to(DEV), W.to(DEV), C2, M, mid, C, accumulate=True, x3=True)
    torch.cuda.synchronize()
    assert rel_err(C2, raw + acc0.double()) < 3e-6


def test_persistent_tiles_equal_one_workgroup_per_tile(knob):
    """Forward and dgrad of a layer with more than 256 tiles run as persistent workgroups (the loaders run ahead into the
    next tile during the epilogue); VQA_PERSISTENT=0 selects one workgroup per tile: same arithmetic, identical bits."""
    ops = _ops()
    g = torch.Generator().manual_seed(77)
    B, H, W, Ci, Co = 3, 150, 142, 64, 128
    x = torch.randn(B, H, W, Ci, generator=g).to(DEV)
    w = (torch.randn(Co, Ci, 3, 3, generator=g) / math.sqrt(9 * Ci)).to(DEV)
    b = (torch.randn(Co, generator=g) * 0.1).to(DEV)
    wf, wd = ops.conv_pack_weights(w, Ci)
    wfx, wdx, xp = ops.x3_split(wf), ops.x3_split(wd), ops.x3_pack(x)
    out = {}
    for mode in ("default", "0"):
        if mode == "0":
            knob("VQA_PERSISTENT", "0")
        pooled, am = ops.conv_fwd(xp, wfx, b, 1, x3=True)
        pooled_p, _ = ops.conv_fwd(x, wfx, b, 1, x3=True, out_packed=True)
        dp = torch.randn(pooled.shape, generator=torch.Generator().manual_seed(5)).to(DEV)
        dx = ops.conv_dgrad(ops.x3_pack(dp), am, wdx, x.shape, 1, x3=True)
        dx2 = ops.conv_dgrad(dp, am, wdx, x.shape, 1, x3=True)
        torch.cuda.synchronize()
        out[mode] = (pooled, am, pooled_p, dx, dx2)
    for a, c in zip(out["default"], out["0"]):
        assert torch.equal(a, c)
    assert torch.equal(out["default"][3], out["default"][4])


def test_fp32x3_falls_back_to_fp32_kernels_on_small_layers():
    """Layers the split kernels do not take (channel counts that are no multiples of 32, narrow rows) keep the fp32 MFMA
    kernels: on such a model the fp32x3 module is the fp32 module, bit for bit."""
    from dl_vqa_amd import VqaNet
    from oracle import vqa_oracle as O
    cfg = {
        "text": {"question_features": 32, "embedding_features": 20, "dropout": 0.0, "num_lstm_layers": 1, "bidirectional": True},
        "image": {"kernel_size": 3, "dropout": 0.0, "num_channels": [3, 8, 16, 24], "stride": 1, "do_skip_connection": False},
        "attention": {"hidden_dim": 64, "glimpses": 2, "do_option": "+", "dropout": 0.0},
        "classifier": {"hidden_dim": 40, "dropout": 0.0},
        "max_answers": 24,
    }
    v, q, a_idx, a_val, _, _, ql = O.synthetic_batch(3, 48, 6, 50, 24, seed=2)
    ys = []
    for dt in ("fp32", "fp32x3"):
        torch.manual_seed(4)
        m = VqaNet(cfg, 50, compute_dtype=dt).to(DEV).eval()
        y = m(v.to(DEV), q.to(DEV), ql.to(DEV))
        y.sum().backward()
        torch.cuda.synchronize()
        ys.append((y.detach(), [p.grad.clone() for p in m.parameters()]))
    assert torch.equal(ys[0][0], ys[1][0])
    assert all(torch.equal(a, b) for a, b in zip(ys[0][1], ys[1][1]))


def test_batch_chunking_with_packed_operands(knob):
    """Batches whose tensors would pass 4 GiB are walked in chunks inside the C ABI; VQA_CONV_CHUNK forces that path on
    small tensors.  With x3-packed inputs / outputs / pooled gradients the chunk offsets count 6 bytes per element:
    forward (both output forms) and dgrad are bit-identical to one launch, wgrad's chunks are extra split-K slabs."""
    ops = _ops()
    g = torch.Generator().manual_seed(78)
    B, H, W, Ci, Co = 5, 38, 42, 64, 128
    x = torch.randn(B, H, W, Ci, generator=g).to(DEV)
    w = (torch.randn(Co, Ci, 3, 3, generator=g) / math.sqrt(9 * Ci)).to(DEV)
    b = (torch.randn(Co, generator=g) * 0.1).to(DEV)
    wf, wd = ops.conv_pack_weights(w, Ci)
    wfx, wdx, xp = ops.x3_split(wf), ops.x3_split(wd), ops.x3_pack(x)

    def run():
        pooled, am = ops.conv_fwd(xp, wfx, b, 1, x3=True)
        pooled_p, _ = ops.conv_fwd(xp, wfx, b, 1, x3=True, out_packed=True)
        dp = torch.sin(pooled * 3.0) + 0.1
        dpp = ops.x3_pack(dp)
        dw, db = torch.empty_like(w), torch.empty_like(b)
        ops.conv_wgrad(xp, dp, am, dw, db, 1, x3=True, dpooled_packed=dpp)
        dx = ops.conv_dgrad(dpp, am, wdx, x.shape, 1, x3=True)
        torch.cuda.synchronize()
        return pooled, am, pooled_p, dx, dw, db

    ref = run()
    knob("VQA_CONV_CHUNK", "2")      # 2 + 2 + 1 images
    got = run()
    for k in range(4):
        assert torch.equal(ref[k], got[k]), k
    assert rel_err(got[4], ref[4]) < 1e-5 and rel_err(got[5], ref[5]) < 1e-5
