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


X3_CASES = [  # B, H, W, Ci, Co, stride
    (2, 58, 58, 64, 128, 1),   # conv1 family: 192x128 tiles forward / wgrad, 256x64 dgrad
    (1, 38, 38, 128, 256, 1),  # conv2 family
    (3, 34, 70, 64, 64, 1),    # Co = 64: the 256x64 forward tile; wide rows (row cursor wraps rows and images)
    (2, 38, 42, 32, 96, 1),    # partial N tile (Co = 96), one K-step per tap
    (2, 69, 73, 32, 64, 2),    # stride 2
    (5, 36, 36, 96, 160, 1),   # three K-steps per tap, 1.25 N tiles, rows % 192 != 0
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
