"""CPU check of the arithmetic behind mmf_amil_desc::gemm = MMF_GEMM_BF16X3 (csrc/mmf_gemm_split.h), restated with
torch.bfloat16 (round to nearest even, what v_cvt_pk_bf16_f32 does); the first plane is taken from the value clamped to the
largest finite bf16, so that a finite fp32 value never splits into an infinite plane:

  * an fp32 value is the EXACT sum of three bf16 values a0 = bf16(clamp(a)), a1 = bf16(a - a0), a2 = bf16(a - a0 - a1);
  * every product of two bf16 values is exact in fp32;
  * the six leading products, accumulated in fp32 smallest first, give a dot product whose error against fp64 is that of
    a plain fp32 dot product (the three dropped products are below 2^-24 |a b|).
"""
import numpy as np
import torch


def split3(a):
    a0 = a.clamp(-3.3895313892515355e38, 3.3895313892515355e38).to(torch.bfloat16).float()      # split_pair's v_med3_f32 + cvt
    r1 = a - a0
    a1 = r1.to(torch.bfloat16).float()
    r2 = r1 - a1
    a2 = r2.to(torch.bfloat16).float()
    return a0, a1, a2


def _samples(n, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, generator=g)
    scale = torch.exp2(torch.randint(-60, 60, (n,), generator=g).float())
    edge = torch.tensor([0.0, -0.0, 1.0, -1.0, 1.0 + 2.0 ** -23, 3.0e38, -3.0e38, 1.1754944e-38, 1e-40, 2.0 ** -149, 255.99998, 0.1,
                         3.4028234e38, -3.4028234e38, 3.39e38, 3.3895e38])      # FLT_MAX and the values nearest-rounding would turn into inf
    return torch.cat([x, x * scale, edge])


def test_three_bf16_values_sum_to_the_fp32_value_exactly():
    a = _samples(200_000, 1)
    a0, a1, a2 = split3(a)
    # exact wherever the third plane is a normal bf16 number (|a| >= 2^-110); below that the planes underflow and the
    # split is off by less than 2^-126 in absolute terms (fp32 subnormal territory)
    big = (a == 0) | (a.abs() >= 2.0 ** -110)
    s = a0.double() + a1.double() + a2.double()
    assert torch.equal(s[big], a.double()[big])                                     # exact, not merely to fp32 rounding
    assert ((s - a.double())[~big].abs() < 2.0 ** -126).all()
    assert torch.isfinite(a0).all() and torch.isfinite(a1).all() and torch.isfinite(a2).all()   # no plane overflows, FLT_MAX included
    # the residuals shrink by 2^-8 per plane (round to nearest: half an ulp of an 8-bit significand); only the clamped values
    # (beyond the largest bf16) leave up to one ulp
    nz = (a != 0) & (a.abs() <= 3.3895313892515355e38)
    assert (a1[nz].abs() <= a[nz].abs() * 2.0 ** -8).all() and (a2[nz].abs() <= a[nz].abs() * 2.0 ** -16).all()


def test_products_of_planes_are_exact_in_fp32():
    a0, a1, a2 = split3(_samples(50_000, 2)[:50_000])
    b0, b1, b2 = split3(_samples(50_000, 3)[:50_000] * 0.04)
    for x in (a0, a1, a2):
        for y in (b0, b1, b2):
            p = x.double() * y.double()
            ok = torch.isfinite(p) & ((p == 0) | (p.abs() > 1e-36))            # stay clear of fp32 underflow
            assert torch.equal((x * y)[ok].double(), p[ok])


def test_six_product_dot_matches_fp32_dot_accuracy():
    g = torch.Generator().manual_seed(4)
    M, K, N = 512, 1024, 128
    a = torch.randn(M, K, generator=g)
    b = torch.randn(N, K, generator=g) * 0.04
    ref = a.double() @ b.double().T
    A, B = split3(a), split3(b)
    terms = [(0, 2), (2, 0), (1, 1), (0, 1), (1, 0), (0, 0)]                       # smallest first, as the kernels issue them
    c = torch.zeros(M, N)
    for k in range(0, K, 16):                                                       # one 16-k chunk per MFMA chain
        for i, j in terms:
            c = c + A[i][:, k:k + 16] @ B[j][:, k:k + 16].T
    plain = torch.zeros(M, N)
    for k in range(0, K, 16):
        plain = plain + a[:, k:k + 16] @ b[:, k:k + 16].T
    scale = ref.abs().max()
    e_split = float((c.double() - ref).abs().max() / scale)
    e_plain = float((plain.double() - ref).abs().max() / scale)
    assert e_split <= 2 * e_plain + 1e-7, (e_split, e_plain)
    # and the three dropped products really are negligible: adding them back changes nothing at fp32 resolution
    drop = sum(A[i] @ B[j].T for i, j in ((1, 2), (2, 1), (2, 2)))
    assert float(drop.abs().max() / scale) < 2.0 ** -22
