"""The arithmetic of the fp32x3 kernels (csrc/x3_core.hpp) restated on the CPU: an fp32 value is split exactly into three
bf16 terms, and a dot product accumulated in fp32 from the six partial products of weight >= 2^-16 is as accurate as the
plain fp32 dot product.  (The GPU tests measure the kernels themselves; this one pins the algorithm they implement.)"""
import torch


def split3(x):
    hi = x.to(torch.bfloat16)
    r1 = x - hi.float()
    mid = r1.to(torch.bfloat16)
    lo = (r1 - mid.float()).to(torch.bfloat16)
    return hi, mid, lo


def test_split_is_exact_and_ordered():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1 << 16, generator=g) * torch.exp2(torch.randint(-30, 30, (1 << 16,), generator=g).float())
    hi, mid, lo = split3(x)
    assert torch.equal(hi.double() + mid.double() + lo.double(), x.double())
    assert bool((mid.double().abs() <= hi.double().abs() * 2.0 ** -8).all())
    assert bool((lo.double().abs() <= hi.double().abs() * 2.0 ** -16).all())


def test_six_partial_products_match_fp32_accuracy():
    g = torch.Generator().manual_seed(1)
    K = 1152                                              # 9 taps x 128 channels: conv2's reduction length
    a, b = torch.randn(256, K, generator=g), torch.randn(K, 64, generator=g)
    ref = a.double() @ b.double()
    pa, pb = split3(a), split3(b)
    pairs = ((2, 0), (0, 2), (1, 1), (1, 0), (0, 1), (0, 0))          # (lo,hi) (hi,lo) (mid,mid) (mid,hi) (hi,mid) (hi,hi)
    # every partial product of two bf16 values is exact in fp32; the accumulation is fp32 (emulated: fp32 matmuls, fp32 adds)
    acc = torch.zeros(256, 64)
    for i, j in pairs:
        acc = acc + pa[i].float() @ pb[j].float()
    err_x3 = float((acc.double() - ref).abs().max() / ref.abs().max())
    err_f32 = float(((a @ b).double() - ref).abs().max() / ref.abs().max())
    dropped = sum((pa[i].double() @ pb[j].double()) for i, j in ((1, 2), (2, 1), (2, 2)))
    rel_dropped = float(dropped.abs().max() / ref.abs().max())
    print(f"six-product error {err_x3:.3e}, fp32 matmul error {err_f32:.3e}, dropped terms {rel_dropped:.3e}")
    assert rel_dropped < 2.0 ** -22                       # far below one fp32 rounding of the result
    assert err_x3 < 3.0 * err_f32 + 1e-7
