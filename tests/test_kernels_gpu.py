"""GPU parity tests, kernel by kernel, through the C ABI (dl_vqa_amd.ops -> libvqa_hip.so).

Each HIP entry point is compared with a float64 torch-CPU computation of the same operator
(floating-point kernels: tolerance stated per test, relative to the magnitude of the result).
"""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _ops():
    from dl_vqa_amd import ops
    return ops


def rel_err(got: torch.Tensor, ref: torch.Tensor) -> float:
    got = got.detach().double().cpu()
    ref = ref.detach().double().cpu()
    scale = max(float(ref.abs().max()), 1e-30)
    return float((got - ref).abs().max()) / scale


@pytest.fixture
def knob(monkeypatch):
    """Set VQA_* environment knobs for one test: the library reads them once, so it is told to re-read them
    (vqa_reload_knobs) after every change and again when the test's environment has been restored."""
    from dl_vqa_amd import _lib

    def set_knob(name, value):
        monkeypatch.setenv(name, value)
        _lib.load().vqa_reload_knobs()
    yield set_knob
    monkeypatch.undo()
    _lib.load().vqa_reload_knobs()


def check(name, got, ref, tol):
    e = rel_err(got, ref)
    print(f"[parity] {name}: max|err|/max|ref| = {e:.3e} (tol {tol:.1e})")
    assert e <= tol, f"{name}: {e} > {tol}"


# ----------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("M,N,K", [(300, 200, 100), (64, 64, 32), (5, 3, 8), (256, 1024, 2560),
                                   (1030, 260, 3584), (129, 4096, 300),
                                   (70, 50, 45), (40, 33, 7), (200, 130, 1001)])   # K tails: K % 4 != 0, K < BK
@pytest.mark.parametrize("transA,transB", [(False, True), (False, False), (True, True), (True, False)])
def test_gemm_layouts(M, N, K, transA, transB):
    ops = _ops()
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    A = torch.randn(M, K, generator=g)
    Bm = torch.randn(K, N, generator=g)
    ref = A.double() @ Bm.double()
    pad = lambda n: (n + 3) // 4 * 4
    if transA:   # stored [K][M]
        As = torch.zeros(K, pad(M)); As[:, :M] = A.t(); lda = pad(M)
    else:
        As = torch.zeros(M, pad(K)); As[:, :K] = A; lda = pad(K)
    if transB:   # stored [N][K]
        Bs = torch.zeros(N, pad(K)); Bs[:, :K] = Bm.t(); ldb = pad(K)
    else:
        Bs = torch.zeros(K, pad(N)); Bs[:, :N] = Bm; ldb = pad(N)
    Cd = torch.full((M, N + 3), 7.0, device=DEV)
    ops.gemm(As.to(DEV), Bs.to(DEV), Cd, M, N, K, transA=transA, transB=transB, lda=lda, ldb=ldb, ldc=N + 3)
    torch.cuda.synchronize()
    check(f"gemm {M}x{N}x{K} tA={transA} tB={transB}", Cd[:, :N], ref, 2e-6 * math.sqrt(K))
    assert float((Cd[:, N:] - 7.0).abs().max()) == 0.0   # nothing written outside N


def test_gemm_epilogue_bias_rowgroup_relu_accumulate():
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    Bn, P, K, N = 3, 7, 40, 36
    M = Bn * P
    A, W = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g)
    b1, b2 = torch.randn(N, generator=g), torch.randn(N, generator=g)
    rg = torch.randn(Bn, N, generator=g)
    C0 = torch.randn(M, N, generator=g)
    acc = A.double() @ W.double().t()
    for op in (0, 1):
        rgx = rg.double().repeat_interleave(P, dim=0)
        ref = acc + rgx if op == 0 else acc * rgx
        ref = torch.relu(ref + b1.double() + b2.double()) + C0.double()
        Cd = C0.clone().to(DEV)
        ops.gemm(A.to(DEV), W.to(DEV), Cd, M, N, K, bias1=b1.to(DEV), bias2=b2.to(DEV), rowgroup=rg.to(DEV),
                 rg_div=P, rg_op=op, relu=True, accumulate=True)
        torch.cuda.synchronize()
        check(f"gemm epilogue rg_op={op}", Cd, ref, 1e-5)
    # split-K path with epilogue (small output, long K)
    M, N, K = 64, 48, 4096
    A, W = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g)
    b1 = torch.randn(N, generator=g)
    C0 = torch.randn(M, N, generator=g)
    ref = torch.relu(A.double() @ W.double().t() + b1.double()) + C0.double()
    Cd = C0.clone().to(DEV)
    ops.gemm(A.to(DEV), W.to(DEV), Cd, M, N, K, bias1=b1.to(DEV), relu=True, accumulate=True)
    torch.cuda.synchronize()
    check("gemm split-K epilogue", Cd, ref, 2e-5)


@pytest.mark.parametrize("M_groups,P,N", [(3, 150, 36), (2, 676, 200), (5, 130, 64)])
def test_gemm_epilogue_tall_rowgroups_and_aux(M_groups, P, N):
    """Row-group terms whose groups are at least a tile tall (the attention shape: q' tiled over the P image
    positions of a sample): tiles inside one group, tiles straddling a boundary, the last partial tile; with the
    raw product also written to `aux`, with and without accumulate, add and multiply."""
    ops = _ops()
    g = torch.Generator().manual_seed(M_groups * 31 + P)
    K = 72
    M = M_groups * P
    A, W = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g)
    b1 = torch.randn(N, generator=g)
    rg = torch.randn(M_groups, N, generator=g)
    C0 = torch.randn(M, N, generator=g)
    acc = A.double() @ W.double().t()
    rgx = rg.double().repeat_interleave(P, dim=0)
    for op, accumulate, relu in ((0, False, True), (1, False, False), (0, True, False), (1, True, True)):
        ref = acc + rgx if op == 0 else acc * rgx
        ref = ref + b1.double()
        if relu:
            ref = torch.relu(ref)
        if accumulate:
            ref = ref + C0.double()
        Cd = C0.clone().to(DEV)
        aux = torch.full((M, N), 3.0, device=DEV)
        ops.gemm(A.to(DEV), W.to(DEV), Cd, M, N, K, bias1=b1.to(DEV), rowgroup=rg.to(DEV), rg_div=P, rg_op=op,
                 relu=relu, accumulate=accumulate, aux=aux)
        torch.cuda.synchronize()
        check(f"gemm rowgroup P={P} op={op} acc={accumulate}", Cd, ref, 1e-5)
        check(f"gemm aux P={P} op={op}", aux, acc, 1e-5)


@pytest.mark.parametrize("n", [8, 37, 3 * 224 * 224 * 2 + 5])
def test_half_to_float(n):
    """fp16 image features are widened on the device; exact for every half value."""
    ops = _ops()
    g = torch.Generator().manual_seed(n)
    x = (torch.randn(n, generator=g) * 3).half()
    x[: min(n, 6)] = torch.tensor([0.0, -0.0, 65504.0, -65504.0, 6e-8, float("inf")])[: min(n, 6)].half()
    y = ops.half_to_float(x.to(DEV))
    torch.cuda.synchronize()
    assert torch.equal(y.cpu(), x.float())


def test_gemm_rejects_misaligned():
    from dl_vqa_amd._lib import VqaHipError
    ops = _ops()
    A = torch.zeros(8, 10, device=DEV)
    B = torch.zeros(8, 10, device=DEV)
    Cd = torch.zeros(8, 8, device=DEV)
    with pytest.raises(VqaHipError):
        ops.gemm(A, B, Cd, 8, 8, 10)        # lda = 10 is not a multiple of 4


# ----------------------------------------------------------------------------- conv + relu + pool
def _nhwc(x, cpad=None):
    x = x.permute(0, 2, 3, 1).contiguous()
    if cpad and cpad != x.shape[-1]:
        x = F.pad(x, (0, cpad - x.shape[-1]))
    return x.contiguous()


CONV_CASES = [  # B, H, W, Ci, Co, stride
    (2, 16, 16, 8, 16, 1),
    (3, 31, 29, 3, 8, 1),      # Ci=3 padded to 4, odd sizes (floor in conv and pool)
    (2, 33, 35, 8, 12, 2),     # stride 2
    (2, 58, 58, 64, 128, 1),   # the conv1 shape family (128x128 tile)
    (1, 30, 30, 128, 256, 1),  # conv2 shape family
    (2, 40, 40, 4, 64, 1),     # conv0 shape family (NHWC4, 128x64 tile)
    (2, 38, 42, 32, 96, 1),    # uniform-tap loaders, Co = 96: partial N tile, wgrad column groups past Co
    (2, 20, 20, 32, 32, 1),    # 2*Wp = 18 < BK: uniform forward / dgrad, general (per-lane cursor) wgrad
    (1, 36, 36, 48, 80, 1),    # channel counts that are no multiples of 32: general loaders everywhere
    (2, 37, 41, 32, 64, 2),    # stride 2 through the uniform-tap loaders
    (3, 34, 70, 64, 64, 1),    # wide rows: the wgrad row cursor wraps rows and images (Mtot % 32 != 0)
]


@pytest.mark.parametrize("B,H,W,Ci,Co,stride", CONV_CASES)
def test_conv_relu_pool_fwd_bwd(B, H, W, Ci, Co, stride):
    ops = _ops()
    g = torch.Generator().manual_seed(B * 1000 + H * 10 + Ci)
    CiP = (Ci + 3) // 4 * 4
    x = torch.randn(B, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, 3, 3, generator=g) / math.sqrt(9 * Ci)
    b = torch.randn(Co, generator=g) * 0.1
    xr = x.double().requires_grad_(True)
    wr = w.double().requires_grad_(True)
    br = b.double().requires_grad_(True)
    yr = F.max_pool2d(torch.relu(F.conv2d(xr, wr, br, stride=stride)), 2, 2)
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy.double())

    xd = _nhwc(x, CiP).to(DEV)
    wf, wd = ops.conv_pack_weights(w.to(DEV), CiP)
    pooled, amax = ops.conv_fwd(xd, wf, b.to(DEV), stride)
    torch.cuda.synchronize()
    check(f"conv fwd {B,H,W,Ci,Co,stride}", pooled.permute(0, 3, 1, 2), yr, 3e-6 * math.sqrt(9 * Ci))
    dead = (amax == 4)
    assert bool(((pooled == 0) == dead).all()), "arg-max code 4 must mark exactly the zero outputs"

    dyd = _nhwc(dy).to(DEV)
    dx = ops.conv_dgrad(dyd, amax, wd, xd.shape, stride)
    dw = torch.empty(Co, Ci, 3, 3, device=DEV)
    db = torch.empty(Co, device=DEV)
    ops.conv_wgrad(xd, dyd, amax, dw, db, stride)
    torch.cuda.synchronize()
    check("conv dgrad", dx[..., :Ci].permute(0, 3, 1, 2), xr.grad, 5e-6 * math.sqrt(9 * Co))
    if CiP != Ci:
        assert float(dx[..., Ci:].abs().max()) == 0.0
    check("conv wgrad", dw, wr.grad, 2e-5)
    check("conv bias grad", db, br.grad, 2e-5)


@pytest.mark.parametrize("B,H,W,Ci,Co", [(2, 30, 30, 64, 128), (3, 23, 41, 128, 256), (1, 111, 111, 64, 128), (5, 17, 9, 64, 8),
                                         (2, 54, 54, 128, 256), (1, 12, 200, 64, 16)])
def test_pconvf_dgrad_matches_the_implicit_gemm_kernel_and_float64(B, H, W, Ci, Co):
    """csrc/conv_patch_f32.hip (flattened-segment patch kernel, exact fp32 MFMA): dX against float64 autograd and against
    vqa_conv3x3_dgrad on the same operands -- odd sizes (dropped pool rows / columns, padded-width garbage columns), segments that
    end inside an image, several images per workgroup stream, 64- and 128-channel outputs, wide and narrow maps."""
    ops = _ops()
    assert ops.pconvf_supported(H, W, Ci, Co)
    g = torch.Generator().manual_seed(H * 7 + W + Ci)
    x = torch.randn(B, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, 3, 3, generator=g) / math.sqrt(9 * Ci)
    b = torch.randn(Co, generator=g) * 0.1
    xr, wr, br = x.double().requires_grad_(True), w.double(), b.double()
    yr = F.max_pool2d(torch.relu(F.conv2d(xr, wr, br)), 2, 2)
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy.double())
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    wf, wd = ops.conv_pack_weights(w.to(DEV), Ci)
    pooled, am = ops.conv_fwd(xd, wf, b.to(DEV), 1)
    dp = dy.permute(0, 2, 3, 1).contiguous().to(DEV)
    want = ops.conv_dgrad(dp, am, wd, tuple(xd.shape), 1)
    got = ops.pconvf_dgrad(dp, am, ops.pconvf_pack_weights(w.to(DEV)), tuple(xd.shape))
    torch.cuda.synchronize()
    scale = float(want.abs().max())
    assert float((got - want).abs().max()) <= 2e-6 * math.sqrt(9 * Co) * scale
    # float64 reference with the KERNEL's arg-max (a near-tie may pick another pixel of a window in float64)
    Hp, Wp = yr.shape[2], yr.shape[3]
    dyfull = torch.zeros(B, H - 2, W - 2, Co, dtype=torch.float64)
    amc, dpc = am.cpu(), dp.cpu().double()
    for j in range(4):
        dyfull[:, (j >> 1):2 * Hp:2, (j & 1):2 * Wp:2, :] = torch.where(amc == j, dpc, torch.zeros_like(dpc))
    dx_ref = torch.nn.grad.conv2d_input(xr.shape, wr, dyfull.permute(0, 3, 1, 2))
    check(f"pconvf dgrad {B,H,W,Ci,Co}", got.permute(0, 3, 1, 2), dx_ref, 3e-6 * math.sqrt(9 * Co))


CONVK_CASES = [  # B, H, W, Ci, Co, ks, stride, images per im2col chunk (0 = one chunk)
    (2, 20, 20, 8, 16, 5, 1, 0),
    (3, 23, 27, 3, 8, 5, 1, 2),      # Ci = 3 padded to 4, odd sizes, a ragged last chunk
    (2, 33, 35, 8, 12, 2, 2, 1),     # even kernel, stride 2, one image per chunk (wgrad accumulates over chunks)
    (2, 17, 19, 16, 8, 1, 1, 0),     # 1 x 1
    (1, 30, 31, 12, 20, 7, 2, 0),    # 7 x 7, stride 2
    (2, 26, 26, 64, 128, 4, 1, 0),   # MFMA-sized channel counts
]


@pytest.mark.parametrize("B,H,W,Ci,Co,ks,stride,chunk", CONVK_CASES)
def test_convk_block_fwd_bwd(B, H, W, Ci, Co, ks, stride, chunk):
    """image.kernel_size != 3 (csrc/conv_generic.hip: im2col + vqa_gemm + pool / route / col2im) against float64 autograd of
    Conv2d(k, stride) -> ReLU -> MaxPool2d(2,2) (models/model.py:80-82)."""
    ops = _ops()
    g = torch.Generator().manual_seed(B * 1000 + H * 10 + Ci + ks)
    CiP = (Ci + 3) // 4 * 4
    x = torch.randn(B, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, ks, ks, generator=g) / math.sqrt(ks * ks * Ci)
    b = torch.randn(Co, generator=g) * 0.1
    xr, wr, br = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    yr = F.max_pool2d(torch.relu(F.conv2d(xr, wr, br, stride=stride)), 2, 2)
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy.double())

    xd = _nhwc(x, CiP).to(DEV)
    wk = ops.convk_pack_weights(w.to(DEV), CiP)
    pooled, amax = ops.convk_fwd(xd, wk, b.to(DEV), ks, stride, chunk=chunk)
    torch.cuda.synchronize()
    check(f"convk fwd {B,H,W,Ci,Co,ks,stride}", pooled.permute(0, 3, 1, 2), yr, 3e-6 * math.sqrt(ks * ks * Ci))
    assert bool(((pooled == 0) == (amax == 4)).all()), "arg-max code 4 must mark exactly the zero outputs"
    dw = torch.empty(Co, Ci, ks, ks, device=DEV)
    db = torch.empty(Co, device=DEV)
    dx = ops.convk_bwd(xd, _nhwc(dy).to(DEV), amax, wk, dw, db, ks, stride, need_dx=True, chunk=chunk)
    torch.cuda.synchronize()
    check("convk dgrad", dx[..., :Ci].permute(0, 3, 1, 2), xr.grad, 5e-6 * math.sqrt(ks * ks * Co))
    if CiP != Ci:
        assert float(dx[..., Ci:].abs().max()) == 0.0
    check("convk wgrad", dw, wr.grad, 2e-5)
    check("convk bias grad", db, br.grad, 2e-5)
    assert ops.convk_bwd(xd, _nhwc(dy).to(DEV), amax, wk, dw, db, ks, stride, need_dx=False, chunk=chunk) is None


def test_convk_block_chunks_at_a_north_star_sized_layer():
    """Block 1 of the reference architecture with kernel_size 5 at 224 x 224 (110 x 110 x 64 -> 128 channels): one image's im2col
    matrix is 72 MB, so a batch of 40 is walked in chunks of 29 + 11 images (2 GiB limit).  The per-image results must not
    depend on the chunking (bitwise), the weight gradient only by its summation order."""
    ops = _ops()
    B, H, Ci, Co, ks = 40, 110, 64, 128, 5
    g = torch.Generator().manual_seed(17)
    x = torch.randn(B, H, H, Ci, generator=g).to(DEV)
    w = (torch.randn(Co, Ci, ks, ks, generator=g) / math.sqrt(ks * ks * Ci)).to(DEV)
    b = (torch.randn(Co, generator=g) * 0.1).to(DEV)
    wk = ops.convk_pack_weights(w, Ci)
    auto = ops.convk_chunk(B, H, H, Ci, Co, ks, 1)
    assert 1 < auto < B
    p1, a1 = ops.convk_fwd(x, wk, b, ks)
    p2, a2 = ops.convk_fwd(x, wk, b, ks, chunk=8)
    assert torch.equal(p1, p2) and torch.equal(a1, a2)
    # spot check of one image against F.conv2d in float64
    ref = F.max_pool2d(torch.relu(F.conv2d(x[B - 1:].permute(0, 3, 1, 2).double().cpu(), w.double().cpu(), b.double().cpu())), 2, 2)
    check("convk fwd (last image of the last chunk)", p1[B - 1:].permute(0, 3, 1, 2), ref, 3e-6 * math.sqrt(ks * ks * Ci))
    dp = torch.randn(p1.shape, generator=g).to(DEV)
    dw1, db1, dw2, db2 = (torch.empty_like(w), torch.empty_like(b), torch.empty_like(w), torch.empty_like(b))
    dx1 = ops.convk_bwd(x, dp, a1, wk, dw1, db1, ks)
    dx2 = ops.convk_bwd(x, dp, a1, wk, dw2, db2, ks, chunk=8)
    torch.cuda.synchronize()
    assert torch.equal(dx1, dx2) and torch.equal(db1, db2)
    check("convk wgrad across chunkings", dw1, dw2, 1e-5)
    assert bool(torch.isfinite(dw1).all()) and float(dx1.abs().max()) > 0


def test_nchw_to_nhwc4():
    ops = _ops()
    x = torch.randn(2, 3, 9, 11)
    y = ops.nchw_to_nhwc4(x.to(DEV)).cpu()
    assert torch.equal(y[..., :3], x.permute(0, 2, 3, 1)) and float(y[..., 3].abs().max()) == 0.0


# ----------------------------------------------------------------------------- point-wise / reductions
@pytest.mark.parametrize("C", [32, 256, 288])        # <= 256: the one-pass register form; above: the two-pass loop
def test_l2norm_fwd_bwd(C):
    ops = _ops()
    g = torch.Generator().manual_seed(3)
    u = torch.randn(37, C, generator=g)
    u[5] = 0.0   # an all-zero pixel: norm 0 -> vn 0, finite gradient
    ur = u.double().requires_grad_(True)
    vr = ur / (ur.norm(p=2, dim=1, keepdim=True).expand_as(ur) + 1e-12)
    dv = torch.randn(37, C, generator=g)
    vr.backward(dv.double())
    vn, norm = ops.l2norm_fwd(u.to(DEV), 0.0, 0)
    du = ops.l2norm_bwd(dv.to(DEV), vn, norm, 0.0, 0)
    torch.cuda.synchronize()
    check("l2norm fwd", vn, vr, 2e-6)
    mask = torch.ones(37, dtype=torch.bool); mask[5] = False
    check("l2norm bwd", du[mask.to(DEV)], ur.grad[mask], 5e-6)
    assert bool(torch.isfinite(du).all())


def test_l2norm_second_output_and_dropout_add():
    """The attention dropout on v (models/model.py:185) is fused twice: forward as a second output of the L2-norm
    pass (fp32, or bf16 on the bf16 path), backward as y += dropout(x) in one pass (vqa_dropout_add).  Both must equal
    the stand-alone vqa_dropout."""
    ops = _ops()
    g = torch.Generator().manual_seed(13)
    u = torch.randn(300, 64, generator=g).to(DEV)
    vn, norm = ops.l2norm_fwd(u, 0.0, 0)
    ref = ops.dropout(vn, 0.3, 777)
    vn2, _, vd = ops.l2norm_fwd(u, 0.0, 0, drop2=(0.3, 777, torch.float32))
    vn3, _, vb = ops.l2norm_fwd(u, 0.0, 0, drop2=(0.3, 777, torch.bfloat16))
    torch.cuda.synchronize()
    assert torch.equal(vn, vn2) and torch.equal(vn, vn3) and torch.equal(vd, ref) and torch.equal(vb, ref.to(torch.bfloat16))
    g2 = torch.randn(300 * 64 + 3, generator=g).to(DEV)            # + a tail that is not a multiple of 4
    acc = torch.randn(300 * 64 + 3, generator=g).to(DEV)
    want = acc + ops.dropout(g2, 0.3, 4242)
    ops.dropout_add(g2, acc, 0.3, 4242)
    torch.cuda.synchronize()
    assert float((acc - want).abs().max()) <= 1e-6 * float(want.abs().max())      # fused multiply-add vs two roundings


@pytest.mark.parametrize("B,Hp,Wp,C,G,p_v,p", [(3, 5, 7, 64, 2, 0.3, 0.2), (2, 4, 4, 256, 1, 0.0, 0.0), (5, 3, 9, 128, 4, 0.5, 0.0),
                                              (2, 3, 3, 512, 2, 0.3, 0.1), (1, 7, 5, 320, 3, 0.0, 0.4)])
def test_l2norm_bwd_joined_equals_the_three_kernel_form(B, Hp, Wp, C, G, p_v, p):
    """vqa_l2norm_bwd_joined (d loss / d vn joined in the kernel from probs x dcomb and dropout-mask x dv_in) against
    att_apply_bwd -> dropout_add -> l2norm_bwd through the [B*P][C] tensor: fp32, bf16 and channel-blocked bf16 outputs."""
    ops = _ops()
    g = torch.Generator().manual_seed(B * 100 + C)
    P = Hp * Wp
    u = torch.randn(B * P, C, generator=g).to(DEV)
    vn, norm = ops.l2norm_fwd(u, 0.0, 0)
    probs = torch.softmax(torch.randn(B, G, P, generator=g), dim=2).to(DEV)
    ld = G * C + 8
    dcomb = torch.randn(B, ld, generator=g).to(DEV)
    dv_in = torch.randn(B * P, C, generator=g).to(DEV)
    dscore, dvn = ops.att_apply_bwd(dcomb, ld, probs, vn.view(B, P, C))
    dscore2, none = ops.att_apply_bwd(dcomb, ld, probs, vn.view(B, P, C), want_dvn=False)
    assert none is None
    dvn = dvn.view(B * P, C)
    if p_v > 0:
        ops.dropout_add(dv_in, dvn, p_v, 991)
    else:
        dvn += dv_in
    torch.cuda.synchronize()
    assert torch.equal(dscore, dscore2)
    for kw in ({"out_dtype": torch.float32}, {"out_dtype": torch.bfloat16}, {"c16_hw": (Hp, Wp)}):
        want = ops.l2norm_bwd(dvn, vn, norm, p, 55, **kw)
        got = ops.l2norm_bwd_joined(dcomb, ld, probs, dv_in, p_v, 991, vn, norm, p, 55, **kw)
        torch.cuda.synchronize()
        if got.dtype == torch.float32:      # the join is a fused multiply-add where dropout_add rounds twice
            assert float((got - want).abs().max()) <= 2e-6 * float(want.abs().max()), kw
        else:
            assert float((got.float() - want.float()).abs().max()) <= 2 ** -7 * float(want.float().abs().max()), kw
            assert float((got != want).float().mean()) < 0.02, kw


def test_embed_tanh_fwd_bwd():
    ops = _ops()
    g = torch.Generator().manual_seed(4)
    V, E, B, T = 20, 12, 5, 4
    emb = torch.randn(V, E, generator=g)
    q = torch.randint(0, V, (B, T), generator=g)
    q[0, 1] = 0
    er = emb.double().requires_grad_(True)
    xr = torch.tanh(F.embedding(q, er, padding_idx=0)).transpose(0, 1)   # [T,B,E]
    dx = torch.randn(T, B, E, generator=g)
    xr.backward(dx.double())
    bad = torch.zeros(1, dtype=torch.int32, device=DEV)
    x = ops.embed_tanh_fwd(q.to(DEV), emb.to(DEV), 0.0, 0, bad)
    demb = torch.full((V, E), 7.0, device=DEV)          # every row is written: no pre-zeroing needed
    ops.embed_tanh_bwd(q.to(DEV), x, dx.to(DEV), demb, 0.0, 0)
    torch.cuda.synchronize()
    check("embed fwd", x, xr, 2e-6)
    check("embed bwd", demb, er.grad, 5e-6)
    assert float(demb[0].abs().max()) == 0.0 and int(bad) == 0


@pytest.mark.parametrize("V,E,B,T,hot", [(5000, 300, 64, 14, 0), (40, 64, 300, 9, 0), (3000, 96, 700, 6, 1500), (7, 1100, 50, 5, 0)])
def test_embed_bwd_binned_equals_the_scanning_kernel(V, E, B, T, hot):
    """with a workspace the slots are binned by token and every vocabulary row sums its own sorted bin (O(B*T + V));
    without one every row scans all B*T slots.  Same summation order, so the same bits -- incl. a token in more than 1 024 slots
    (bin overflow: that row scans), a vocabulary much larger than the batch, bad ids, train-mode dropout."""
    ops = _ops()
    g = torch.Generator().manual_seed(V + B)
    emb = torch.randn(V, E, generator=g)
    q = torch.randint(0, V, (B, T), generator=g)
    if hot:
        q.view(-1)[torch.randperm(B * T, generator=g)[:hot]] = 3
    q[1, 2], q[2, 0] = V + 1, -4
    dx = torch.randn(T, B, E, generator=g).to(DEV)
    bad = torch.zeros(1, dtype=torch.int32, device=DEV)
    x = ops.embed_tanh_fwd(q.to(DEV), emb.to(DEV), 0.3, 99, bad)
    d1 = torch.full((V, E), 7.0, device=DEV)
    d2 = torch.full((V, E), -7.0, device=DEV)
    ops.embed_tanh_bwd(q.to(DEV), x, dx, d1, 0.3, 99, binned=True)
    ops.embed_tanh_bwd(q.to(DEV), x, dx, d2, 0.3, 99, binned=False)
    torch.cuda.synchronize()
    assert torch.equal(d1, d2) and float(d1[0].abs().max()) == 0.0 and bool(torch.isfinite(d1).all())


def test_embed_bwd_is_deterministic_and_counts_bad_tokens():
    """Many slots per vocabulary row (B*T >> V, more than one 256-slot scan round, > 256 hits of one token in a
    round impossible by construction): the per-row sums run in slot order, so two launches agree bit for bit;
    ids outside [0, V) are counted, read as a zero row and receive no gradient."""
    ops = _ops()
    g = torch.Generator().manual_seed(8)
    V, E, B, T = 7, 300, 150, 9
    emb = torch.randn(V, E, generator=g)
    q = torch.randint(0, V, (B, T), generator=g)
    q[3, 2], q[10, 0], q[11, 8] = V, -1, V + 5
    bad = torch.zeros(1, dtype=torch.int32, device=DEV)
    x = ops.embed_tanh_fwd(q.to(DEV), emb.to(DEV), 0.0, 0, bad)
    dx = torch.randn(T, B, E, generator=g)
    d1, d2 = torch.empty(V, E, device=DEV), torch.empty(V, E, device=DEV)
    ops.embed_tanh_bwd(q.to(DEV), x, dx.to(DEV), d1, 0.0, 0)
    ops.embed_tanh_bwd(q.to(DEV), x, dx.to(DEV), d2, 0.0, 0)
    torch.cuda.synchronize()
    assert int(bad) == 3 and torch.equal(d1, d2)
    ok = (q >= 0) & (q < V)
    qs = torch.where(ok, q, torch.zeros_like(q))
    er = emb.double().requires_grad_(True)
    xr = (torch.tanh(F.embedding(qs, er, padding_idx=0)) * ok[..., None]).transpose(0, 1)
    xr.backward(dx.double())
    check("embed fwd (bad ids -> zero rows)", x, xr, 2e-6)
    check("embed bwd (deterministic)", d1, er.grad, 1e-5)


def test_lstm_cell_fwd_bwd():
    ops = _ops()
    g = torch.Generator().manual_seed(6)
    B, H, t = 6, 16, 2
    q_len = torch.tensor([5, 3, 1, 2, 4, 3])
    xg, hg = torch.randn(B, 4 * H, generator=g), torch.randn(B, 4 * H, generator=g)
    c0, h0 = torch.randn(B, H, generator=g), torch.randn(B, H, generator=g)
    xr = (xg + hg).double().requires_grad_(True)
    cr = c0.double().requires_grad_(True)
    hr = h0.double().requires_grad_(True)
    i, f, gg, o = xr.split(H, dim=1)
    cn = torch.sigmoid(f) * cr + torch.sigmoid(i) * torch.tanh(gg)
    hn = torch.sigmoid(o) * torch.tanh(cn)
    m = (q_len > t).double().unsqueeze(1)
    c1 = m * cn + (1 - m) * cr
    h1 = m * hn + (1 - m) * hr
    dh, dc = torch.randn(B, H, generator=g), torch.randn(B, H, generator=g)
    (c1 * dc.double() + h1 * dh.double()).sum().backward()

    gates = torch.empty(B, 4 * H, device=DEV)
    c_out, h_out = torch.empty(B, H, device=DEV), torch.empty(B, H, device=DEV)
    cf = torch.zeros(B, 2 * H + 4, device=DEV)
    ops.lstm_cell_fwd(xg.to(DEV), hg.to(DEV), c0.to(DEV), h0.to(DEV), q_len.to(DEV), t, gates, c_out, h_out,
                      cf[:, 4:], 2 * H + 4)
    dhd, dcd = dh.clone().to(DEV), dc.clone().to(DEV)
    dg = torch.empty(B, 4 * H, device=DEV)
    ops.lstm_cell_bwd(gates, c0.to(DEV), c_out, q_len.to(DEV), t, dhd, dcd, dg)
    torch.cuda.synchronize()
    check("lstm cell c", c_out, c1, 2e-6)
    check("lstm cell h", h_out, h1, 2e-6)
    check("lstm cell c_final", cf[:, 4:4 + H], c1, 2e-6)
    check("lstm cell dgates", dg, xr.grad, 5e-6)
    check("lstm cell dc_in", dcd, cr.grad, 5e-6)
    check("lstm cell dh passthrough", dhd, hr.grad, 5e-6)


def _lstm_reference(xg, w_hh, q_len, reverse, dcn):
    """fp64 masked LSTM recurrence of one direction (the oracle's loop) with autograd: returns the saved gate
    activations [T,B,4H] (zero rows where inactive), the state chains in the library's slot convention, c_n, and the
    gradients w.r.t. the gate pre-activations [T,B,4H] for d loss / d c_n = dcn."""
    T, B, H4 = xg.shape
    H = H4 // 4
    w = w_hh.double()
    pres = [None] * T
    h = torch.zeros(B, H, dtype=torch.float64)
    c = torch.zeros(B, H, dtype=torch.float64)
    Hs, Cs = torch.zeros(T + 1, B, H, dtype=torch.float64), torch.zeros(T + 1, B, H, dtype=torch.float64)
    gates = torch.zeros(T, B, 4 * H, dtype=torch.float64)
    for t in (range(T - 1, -1, -1) if reverse else range(T)):
        x_t = xg[t].double().clone().requires_grad_(True)
        pres[t] = x_t
        pre = x_t + h @ w.t()
        i, f, g, o = pre.split(H, dim=1)
        i, f, o, g = torch.sigmoid(i), torch.sigmoid(f), torch.sigmoid(o), torch.tanh(g)
        cn = f * c + i * g
        hn = o * torch.tanh(cn)
        m = (q_len > t).double().unsqueeze(1)
        gates[t] = (m * torch.cat([i, f, g, o], dim=1)).detach()
        c = m * cn + (1 - m) * c
        h = m * hn + (1 - m) * h
        so = t if reverse else t + 1
        Hs[so], Cs[so] = h.detach(), c.detach()
    (c * dcn.double()).sum().backward()
    dg = torch.stack([p.grad if p.grad is not None else torch.zeros(B, 4 * H, dtype=torch.float64) for p in pres])
    return gates, Hs, Cs, c.detach(), dg


@pytest.mark.parametrize("B,H,T,ndir", [(6, 32, 4, 2), (70, 64, 5, 2), (130, 96, 3, 1), (256, 1024, 3, 2), (3, 32, 1, 2)])
@pytest.mark.parametrize("graph", [False, True])
def test_lstm_sequence_fwd_bwd(B, H, T, ndir, graph):
    """vqa_lstm_seq_fwd / vqa_lstm_seq_bwd (one call per sequence, every direction per launch, plain launches or a
    cached hipGraph) against an fp64 masked recurrence with autograd: saved gates, state chains, c_n written into a
    strided output, and the gate pre-activation gradients of BPTT; ragged lengths incl. 1 and T."""
    ops = _ops()
    assert ops.lstm_step_supported(H) and not ops.lstm_step_supported(20)
    g = torch.Generator().manual_seed(B + H + T)
    q_len = torch.randint(1, T + 1, (B,), generator=g)
    q_len[0], q_len[-1] = T, 1
    d = lambda x: x.to(DEV)
    dirs_f, dirs_b, refs, keep = [], [], [], []
    cf = torch.zeros(B, ndir * H + 4, device=DEV)
    for k in range(ndir):
        w_hh = torch.randn(4 * H, H, generator=g) / math.sqrt(H)
        xg = torch.randn(T, B, 4 * H, generator=g)
        dcn = torch.randn(B, H, generator=g)
        refs.append(_lstm_reference(xg, w_hh, q_len, bool(k), dcn))
        t_ = dict(w_hh=d(w_hh), xg=d(xg), gates=torch.full((T, B, 4 * H), 9.0, device=DEV),
                  Hs=torch.full((T + 1, B, H), 9.0, device=DEV), Cs=torch.full((T + 1, B, H), 9.0, device=DEV),
                  c_final=cf[:, 4 + k * H:], reverse=bool(k))
        t_["Hs"][T if k else 0].zero_()
        t_["Cs"][T if k else 0].zero_()
        dirs_f.append(t_)
        dirs_b.append(dict(w_hh=t_["w_hh"], gates=t_["gates"], Hs=t_["Hs"], Cs=t_["Cs"],
                           dgates=torch.full((T, B, 4 * H), 9.0, device=DEV), dh=torch.zeros(B, H, device=DEV),
                           dc=d(dcn.clone()), reverse=bool(k)))
        keep.append(dcn)
    ql = d(q_len)
    before = ops.lstm_graph_stats()
    for rep in range(2):                       # the second round replays the cached graphs (same buffers)
        for k in range(ndir):
            dirs_b[k]["dh"].zero_()
            dirs_b[k]["dc"].copy_(keep[k])
        ops.lstm_seq_fwd(dirs_f, ql, B, T, H, cf_ld=ndir * H + 4, use_graph=graph)
        ops.lstm_seq_bwd(dirs_b, ql, B, T, H, use_graph=graph)
    torch.cuda.synchronize()
    after = ops.lstm_graph_stats()
    if graph:
        assert after[0] - before[0] >= 2 and after[1] - before[1] <= 2, (before, after)   # >= 2 replays, <= 2 builds
    else:
        assert after[:3] == before[:3]
    assert float(cf[:, :4].abs().max()) == 0.0
    for k in range(ndir):
        gates, Hs, Cs, cn, dg = refs[k]
        check(f"lstm seq gates dir{k}", dirs_f[k]["gates"], gates, 5e-6)
        check(f"lstm seq Hs dir{k}", dirs_f[k]["Hs"], Hs, 5e-6)
        check(f"lstm seq Cs dir{k}", dirs_f[k]["Cs"], Cs, 5e-6)
        check(f"lstm seq c_n dir{k}", cf[:, 4 + k * H:4 + (k + 1) * H], cn, 5e-6)
        check(f"lstm seq dgates dir{k}", dirs_b[k]["dgates"], dg, 2e-5)


def test_lstm_sequence_matches_unfused_path():
    """The fused sequence kernels against the unfused entry points they replace (vqa_gemm + vqa_lstm_cell_fwd, resp.
    vqa_lstm_cell_bwd + vqa_gemm(accumulate)) at the bench shape B=256, H=1024."""
    ops = _ops()
    B, H, T = 256, 1024, 4
    g = torch.Generator().manual_seed(3)
    q_len = torch.randint(1, T + 1, (B,), generator=g).to(DEV)
    w_hh = (torch.randn(4 * H, H, generator=g) / math.sqrt(H)).to(DEV)
    xg = torch.randn(T, B, 4 * H, generator=g).to(DEV)
    dcn = torch.randn(B, H, generator=g).to(DEV)
    z = lambda *s_: torch.zeros(*s_, device=DEV)
    f = dict(w_hh=w_hh, xg=xg, gates=z(T, B, 4 * H), Hs=z(T + 1, B, H), Cs=z(T + 1, B, H), c_final=None, reverse=False)
    ops.lstm_seq_fwd([f], q_len, B, T, H, use_graph=False)
    b = dict(w_hh=w_hh, gates=f["gates"], Hs=f["Hs"], Cs=f["Cs"], dgates=z(T, B, 4 * H), dh=z(B, H), dc=dcn.clone())
    ops.lstm_seq_bwd([b], q_len, B, T, H, use_graph=False)
    gates, Hs, Cs, hg = z(T, B, 4 * H), z(T + 1, B, H), z(T + 1, B, H), z(B, 4 * H)
    for t in range(T):
        ops.gemm(Hs[t], w_hh, hg, B, 4 * H, H)
        ops.lstm_cell_fwd(xg[t], hg, Cs[t], Hs[t], q_len, t, gates[t], Cs[t + 1], Hs[t + 1], None, 0)
    dg, dh, dc = z(T, B, 4 * H), z(B, H), dcn.clone()
    for t in range(T - 1, -1, -1):
        ops.lstm_cell_bwd(gates[t], Cs[t], Cs[t + 1], q_len, t, dh, dc, dg[t])
        if t:
            ops.gemm(dg[t], w_hh, dh, B, H, 4 * H, transB=False, lda=4 * H, ldb=H, accumulate=True)
    torch.cuda.synchronize()
    check("seq vs unfused gates", f["gates"], gates.double(), 2e-6)
    check("seq vs unfused Cs", f["Cs"], Cs.double(), 2e-6)
    check("seq vs unfused dgates", b["dgates"], dg.double(), 2e-5)
    check("seq vs unfused dc", b["dc"], dc.double(), 2e-5)


@pytest.mark.parametrize("B,P,C,G", [(2, 676, 256, 2), (2, 70, 130, 3), (3, 100, 72, 1), (2, 17, 64, 4), (1, 65, 8, 2)])
def test_attention_apply_fwd_shapes(B, P, C, G):
    """softmax over positions + weighted sum, both the 16-byte-load form (C % 4 == 0) and the scalar one, channel
    blocks that are not full, position counts around the 64-position unroll, an output row stride wider than G*C."""
    ops = _ops()
    g = torch.Generator().manual_seed(B * 1000 + P + C)
    score = torch.randn(B, G, P, generator=g) * 2
    vn = torch.randn(B, P, C, generator=g)
    pr = torch.softmax(score.double(), dim=-1)
    ref = torch.einsum("bgp,bpc->bgc", pr, vn.double()).reshape(B, G * C)
    ld = G * C + 12
    out = torch.full((B, ld), 7.0, device=DEV)
    probs = ops.att_apply_fwd(score.to(DEV), vn.to(DEV), out, ld)
    torch.cuda.synchronize()
    check(f"att_apply_fwd probs {B,P,C,G}", probs, pr, 3e-6)
    check(f"att_apply_fwd out {B,P,C,G}", out[:, :G * C], ref, 3e-6)
    assert bool((out[:, G * C:] == 7.0).all())


def test_attention_score_and_apply():
    ops = _ops()
    g = torch.Generator().manual_seed(8)
    B, P, C, mid, G = 3, 9, 32, 24, 2
    xs = torch.relu(torch.randn(B * P, mid, generator=g))
    wx, bx = torch.randn(G, mid, generator=g), torch.randn(G, generator=g)
    vn = torch.randn(B, P, C, generator=g)
    xr = xs.double().requires_grad_(True)
    wr, br, vr = wx.double().requires_grad_(True), bx.double().requires_grad_(True), vn.double().requires_grad_(True)
    sc = (xr @ wr.t() + br).reshape(B, P, G).permute(0, 2, 1)          # [B,G,P]
    pr = torch.softmax(sc, dim=-1)
    out = torch.einsum("bgp,bpc->bgc", pr, vr).reshape(B, G * C)
    dout = torch.randn(B, G * C + 8, generator=g)
    (out * dout[:, :G * C].double()).sum().backward()

    score = ops.att_score_fwd(xs.to(DEV), wx.to(DEV), bx.to(DEV), B, P, 0.0, 0)
    outd = torch.zeros(B, G * C + 8, device=DEV)
    probs = ops.att_apply_fwd(score, vn.to(DEV), outd, G * C + 8)
    dscore, dvn = ops.att_apply_bwd(dout.to(DEV), G * C + 8, probs, vn.to(DEV))
    xs_io = xs.clone().to(DEV)
    dwx_part, dq_part, RS = ops.att_score_bwd(dscore, wx.to(DEV), xs_io, B, P, 0.0, 0)
    dwx = torch.empty(G * mid, device=DEV)
    ops.colsum(dwx_part, B * RS, G * mid, dwx)
    dq = torch.empty(B, mid, device=DEV)
    ops.sum_parts(dq_part, dq, B, RS, mid)
    dbx = torch.empty(G, device=DEV)
    ops.sum_bgp(dscore, dbx)
    torch.cuda.synchronize()
    check("att score", score, sc, 3e-6)
    check("att probs", probs, pr, 3e-6)
    check("att weighted", outd[:, :G * C], out, 3e-6)
    check("att dvn", dvn, vr.grad, 1e-5)
    # xr.grad is zero where xs == 0 only through the relu of the caller; here dxpre masks xs > 0
    check("att dxpre", xs_io, xr.grad * (xs > 0).double(), 1e-5)
    check("att dwx", dwx.view(G, mid), wr.grad, 1e-5)
    # softmax gradients sum to zero over positions: dbx is exactly 0 in exact arithmetic
    assert float(dbx.abs().max()) < 1e-5 and float(br.grad.abs().max()) < 1e-12
    check("att dq(+)", dq, (xr.grad * (xs > 0).double()).reshape(B, P, mid).sum(1), 1e-5)


@pytest.mark.parametrize("mid,G", [(256, 1), (512, 2), (768, 2), (1024, 2), (1024, 1), (1024, 3)])
def test_attention_score_fwd_rows_in_flight(mid, G):
    """fp32 x with mid a multiple of 256 (the reference's 1024) takes att_score_fwd_rows_kernel for G <= 2 (whole row + the next
    row's loads in flight, weights in registers); G = 3 stays on the general kernel.  Against float64, and with dropout
    against the stand-alone dropout pass on the same (seed, element) stream."""
    ops = _ops()
    g = torch.Generator().manual_seed(mid + G)
    B, P = 7, 333                                   # 2331 rows: more than one row per wave, a ragged tail
    xs = torch.relu(torch.randn(B * P, mid, generator=g))
    wx, bx = torch.randn(G, mid, generator=g), torch.randn(G, generator=g)
    want = (xs.double() @ wx.double().t() + bx.double()).reshape(B, P, G).permute(0, 2, 1)
    xd, wd, bd = xs.to(DEV), wx.to(DEV), bx.to(DEV)
    score = ops.att_score_fwd(xd, wd, bd, B, P, 0.0, 0)
    torch.cuda.synchronize()
    check(f"att score rows mid={mid} G={G}", score, want, 3e-6)
    dropped = ops.dropout(xd, 0.3, 991)
    s_drop = ops.att_score_fwd(xd, wd, bd, B, P, 0.3, 991)
    s_ref = ops.att_score_fwd(dropped, wd, bd, B, P, 0.0, 0)
    torch.cuda.synchronize()
    assert float((s_drop - s_ref).abs().max()) <= 1e-6 * float(s_ref.abs().max())


def test_softce_loss_score_and_grad():
    from oracle import vqa_oracle as O
    ops = _ops()
    g = torch.Generator().manual_seed(9)
    B, A = 7, 50
    logits = torch.randn(B, A, generator=g) * 3
    a_idx = torch.zeros(B, 3, dtype=torch.int64)
    a_val = torch.zeros(B, 3, dtype=torch.int64)
    for b in range(B):
        k = 1 + b % 3
        a_idx[b, :k] = torch.randperm(A, generator=g)[:k] + 1
        a_val[b, :k] = torch.randint(1, 6, (k,), generator=g)
    a_idx[2, 0] = int(logits[2].argmax()) + 1            # one sure hit for the score
    lr = logits.double().requires_grad_(True)
    loss = O.soft_ce_loss(lr, a_idx, a_val)
    loss.backward()
    ld = A + 2
    ld_buf = torch.zeros(B, ld, device=DEV); ld_buf[:, :A] = logits.to(DEV)
    dl = torch.zeros(B, ld, device=DEV)
    loss_rows, score_rows = ops.softce(ld_buf, ld, a_idx.to(DEV), a_val.to(DEV), A, 1.0 / B, dl, ld)
    torch.cuda.synchronize()
    check("softce loss", loss_rows.sum(), loss, 3e-6)
    check("softce dlogits", dl[:, :A], lr.grad, 1e-5)
    check("vqa score", score_rows.sum(), O.batch_accuracy(logits, a_idx, a_val), 1e-6)


def test_colsum_masked_and_adam_and_misc():
    from oracle import vqa_oracle as O
    ops = _ops()
    g = torch.Generator().manual_seed(10)
    x = torch.randn(1000, 70, generator=g)
    mask = torch.randint(0, 5, (1000, 70), generator=g).to(torch.uint8)
    out = torch.ones(70, device=DEV)
    ops.colsum(x.to(DEV), 1000, 70, out, mask=mask.to(DEV), accumulate=True)
    torch.cuda.synchronize()
    check("colsum masked+acc", out, (x.double() * (mask != 4)).sum(0) + 1, 1e-5)
    # Adam against the oracle (== torch.optim.Adam, tests/test_oracle_golden.py)
    p, gr = torch.randn(1000, generator=g), torch.randn(1000, generator=g)
    m, v = torch.zeros(1000), torch.zeros(1000)
    pd, md, vd = p.clone().to(DEV), m.clone().to(DEV), v.clone().to(DEV)
    for step in (1, 2, 3):
        lr = O.learning_rate(5e-4, step - 1)
        O.adam_step(p, gr, m, v, step, lr)
        ops.adam(pd, gr.to(DEV), md, vd, lr, step)
    torch.cuda.synchronize()
    check("adam", pd, p, 1e-6)
    # relu+dropout backward with p = 0, add
    y, dy = torch.randn(999, generator=g), torch.randn(999, generator=g)
    dx = torch.empty(999, device=DEV)
    ops.relu_drop_bwd(y.to(DEV), dy.to(DEV), dx, 0.0, 0)
    s = torch.empty(999, device=DEV)
    ops.add(y.to(DEV), dy.to(DEV), s)
    torch.cuda.synchronize()
    check("relu bwd", dx, dy * (y > 0), 0.0)
    check("add", s, y + dy, 0.0)


def test_dropout_statistics_and_determinism():
    ops = _ops()
    x = torch.ones(1 << 20, device=DEV)
    y1 = ops.dropout(x, 0.3, 1234)
    y2 = ops.dropout(x, 0.3, 1234)
    y3 = ops.dropout(x, 0.3, 1235)
    torch.cuda.synchronize()
    assert torch.equal(y1, y2)                       # same seed, same mask (backward regenerates it)
    assert not torch.equal(y1, y3)
    keep = float((y1 > 0).float().mean())
    assert abs(keep - 0.7) < 3e-3, keep
    vals = torch.unique(y1)
    assert len(vals) == 2 and abs(float(vals.max()) - 1 / 0.7) < 1e-6
    assert abs(float(y1.mean()) - 1.0) < 5e-3        # inverted dropout keeps the mean


# ----------------------------------------------------------------------------- first conv block (dedicated path)
@pytest.mark.parametrize("B,Ci,H,W,Co", [(2, 3, 20, 24, 32), (3, 3, 31, 28, 64), (2, 2, 16, 16, 64), (1, 1, 9, 12, 32),
                                         (2, 3, 224, 224, 64), (1, 3, 62, 448, 64)])   # 448 wide: > 64 KB of LDS (configs[3])
def test_conv0_dedicated_fwd_wgrad(B, Ci, H, W, Co):
    ops = _ops()
    assert ops.conv0_supported(Ci, H, W, Co, 1)
    assert not ops.conv0_supported(Ci, H, W, Co, 2) and not ops.conv0_supported(Ci, H, W + 1, Co, 1)
    assert not ops.conv0_supported(4, H, W, Co, 1)
    g = torch.Generator().manual_seed(B * 100 + H + Co)
    x = torch.randn(B, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, 3, 3, generator=g) / math.sqrt(9 * Ci)
    b = torch.randn(Co, generator=g) * 0.1
    wr, br = w.double().requires_grad_(True), b.double().requires_grad_(True)
    yr = F.max_pool2d(torch.relu(F.conv2d(x.double(), wr, br)), 2, 2)
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy.double())
    xd = x.to(DEV)
    pooled, amax = ops.conv0_fwd(xd, w.to(DEV), b.to(DEV))
    dw, db = torch.empty(Co, Ci, 3, 3, device=DEV), torch.empty(Co, device=DEV)
    ops.conv0_wgrad(xd, _nhwc(dy).to(DEV), amax, dw, db)
    torch.cuda.synchronize()
    check(f"conv0 fwd {B,Ci,H,W,Co}", pooled.permute(0, 3, 1, 2), yr, 1e-5)
    assert bool(((pooled == 0) == (amax == 4)).all())
    check("conv0 wgrad", dw, wr.grad, 3e-5)
    check("conv0 bias grad", db, br.grad, 3e-5)
    # same arg-max routing as the generic implicit-GEMM path
    wf, _ = ops.conv_pack_weights(w.to(DEV), 4, need_wd=False)   # generic path: channels padded to 4
    p2, a2 = ops.conv_fwd(ops.nchw_to_nhwc4(xd), wf, b.to(DEV), 1)
    torch.cuda.synchronize()
    assert float((p2 - pooled).abs().max()) < 1e-5 and float((a2 != amax).float().mean()) < 1e-3


# ----------------------------------------------------------------------------- every tile configuration
def test_conv_batch_chunking(knob):
    """Batches whose tensors would pass 4 GiB are walked in chunks inside the C ABI (32-bit offsets per launch);
    VQA_CONV_CHUNK forces that path on small tensors: forward and dgrad are bit-identical to one launch, wgrad's
    chunks are extra split-K slabs of the same reduce."""
    ops = _ops()
    g = torch.Generator().manual_seed(77)
    B, H, Ci, Co = 5, 26, 32, 64
    x = torch.randn(B, H, H, Ci, generator=g).to(DEV)
    w = (torch.randn(Co, Ci, 3, 3, generator=g) / math.sqrt(9 * Ci)).to(DEV)
    b = (torch.randn(Co, generator=g) * 0.1).to(DEV)
    wf, wd = ops.conv_pack_weights(w, Ci)

    def run():
        pooled, am = ops.conv_fwd(x, wf, b, 1)
        dp = torch.sin(pooled * 3.0) + 0.1
        dw, db = torch.empty_like(w), torch.empty_like(b)
        ops.conv_wgrad(x, dp, am, dw, db, 1)
        dx = ops.conv_dgrad(dp, am, wd, x.shape, 1)
        torch.cuda.synchronize()
        return pooled, am, dw, db, dx

    ref = run()
    knob("VQA_CONV_CHUNK", "2")      # 2 + 2 + 1 images
    got = run()
    assert torch.equal(ref[0], got[0]) and torch.equal(ref[1], got[1]) and torch.equal(ref[4], got[4])
    check("chunked wgrad dw", got[2], ref[2].double(), 1e-5)
    check("chunked wgrad db", got[3], ref[3].double(), 1e-5)


@pytest.mark.parametrize("w192,w384", [("0", "0"), ("1", "0"), ("0", "1")])
def test_wgrad_tall_tiles_forced(w192, w384, knob):
    """The tall wgrad tiles (192x128 where 9*CiP = 576, 384x128 where 9*CiP = 1152: 8 MFMA waves, one workgroup
    per CU) and the 96- / 128-row tiles they replace are all parity-checked: the two variables force every choice."""
    knob("VQA_WGRAD_192", w192)
    knob("VQA_WGRAD_384", w384)
    test_conv_relu_pool_fwd_bwd(2, 58, 58, 64, 128, 1)
    test_conv_relu_pool_fwd_bwd(1, 30, 30, 128, 256, 1)


@pytest.mark.parametrize("persistent", ["0", "1"])
def test_persistent_tiles_forced(persistent, knob):
    """VQA_PERSISTENT forces the persistent-tile kernels (normally chosen for short-K GEMMs only) on or off."""
    knob("VQA_PERSISTENT", persistent)
    test_conv_relu_pool_fwd_bwd(2, 58, 58, 64, 128, 1)
    test_conv_relu_pool_fwd_bwd(3, 31, 29, 3, 8, 1)
    test_gemm_layouts(1030, 260, 3584, False, True)
    test_gemm_layouts(300, 200, 100, False, False)
    test_gemm_epilogue_bias_rowgroup_relu_accumulate()


@pytest.mark.parametrize("big", ["0", "1", "2", "3"])
def test_tile_configurations_forced(big, knob):
    """The 256-row / 8-MFMA-wave tile configurations are normally chosen by problem size (only the bench
    shapes reach them); VQA_BIG_TILES forces each choice so that every compiled kernel is parity-checked:
    0 = 128-row tiles everywhere, 1 = 256x128 / 256x64 conv forward + dgrad, 2 = 256x128 generic GEMM, 3 = 256-row conv forward + dgrad tiles with 8 loader waves (1024 threads)."""
    knob("VQA_BIG_TILES", big)
    test_conv_relu_pool_fwd_bwd(2, 58, 58, 64, 128, 1)
    test_conv_relu_pool_fwd_bwd(1, 30, 30, 128, 256, 1)
    test_conv_relu_pool_fwd_bwd(2, 40, 40, 64, 64, 1)
    test_gemm_layouts(1030, 260, 3584, False, True)
    test_gemm_layouts(1030, 260, 3584, False, False)
    test_gemm_layouts(300, 200, 100, True, True)
