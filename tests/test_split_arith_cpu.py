"""CPU restatement of the operand split every SPLIT kernel rests on (sr3_common.h: split3_pair; the device self-test
sr3_selftest_split3 checks the same property on the GPU): a finite fp32 value whose residuals stay normal is EXACTLY the sum of three
bf16 terms, x = h + m + l, h = rn_bf16(x), m = rn_bf16(x - h), l = rn_bf16(x - h - m), every residual exact in fp32 -- and the six
products the kernels keep (hh, hm, mh, mm, hl, lh) leave out terms of at most 2^-23 of the product.  numpy only."""
import numpy as np


def rn_bf16(x):
    """fp32 -> bf16 (round to nearest even) -> fp32, on the bit patterns (what v_cvt_pk_bf16_f32 does for finite inputs)."""
    u = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return r.astype(np.uint32).view(np.float32)


def split3(x):
    x = np.asarray(x, dtype=np.float32)
    h = rn_bf16(x)
    r1 = (x - h).astype(np.float32)
    m = rn_bf16(r1)
    r2 = (r1 - m).astype(np.float32)
    l = rn_bf16(r2)
    return h, m, l, r1, r2


def _patterns(n, seed=0):
    g = np.random.default_rng(seed)
    bits = g.integers(0, 2 ** 32, size=n, dtype=np.uint64)
    mant, sign = bits & 0x007FFFFF, bits & 0x80000000
    expo = (27 + (bits >> 23) % 200) << 23                       # exponents 27..226: every residual stays a normal number
    vals = [(sign | expo | mant), (sign | expo),                                   # random significands, powers of two
            (sign | (np.uint64(100) << 23) | (mant & 0x7F0000) | 0x7FFF),          # just below a bf16 tie
            (sign | (np.uint64(140) << 23) | (mant & 0x7F0000) | 0x8001),          # just above
            (sign | (np.uint64(90) << 23) | (mant & 0x7F0000) | 0x8000),           # exact ties (round to even)
            (sign | (np.uint64(127) << 23) | 0x7FFFFF)]                            # 24 set significand bits
    return np.concatenate(vals).astype(np.uint32).view(np.float32)


def test_three_bf16_terms_reproduce_an_fp32_value_exactly():
    x = _patterns(200000)
    h, m, l, r1, r2 = split3(x)
    # the residuals are exact (float64 arithmetic agrees with the fp32 subtraction)
    assert np.array_equal(r1.astype(np.float64), x.astype(np.float64) - h.astype(np.float64))
    assert np.array_equal(r2.astype(np.float64), r1.astype(np.float64) - m.astype(np.float64))
    # and the third term takes what is left: nothing remains
    assert np.array_equal(l, r2)
    assert np.array_equal(h.astype(np.float64) + m.astype(np.float64) + l.astype(np.float64), x.astype(np.float64))
    # every term is a bf16 number (low 16 bits clear)
    for t in (h, m, l):
        assert not (t.view(np.uint32) & 0xFFFF).any()
    # magnitudes: each term at most 2^-8 of the one before (half an ulp of an 8-bit significand, with the tie margin)
    nz = x != 0
    assert (np.abs(m[nz]) <= np.abs(x[nz]) * 2.0 ** -8).all() and (np.abs(l[nz]) <= np.abs(x[nz]) * 2.0 ** -16).all()


def test_six_kept_products_drop_at_most_2_to_minus_23_of_the_product():
    g = np.random.default_rng(3)
    a = (g.standard_normal(100000) * np.exp(g.standard_normal(100000))).astype(np.float32)
    b = (g.standard_normal(100000) * np.exp(g.standard_normal(100000))).astype(np.float32)
    ah, am, al, _, _ = split3(a)
    bh, bm, bl, _, _ = split3(b)
    f = np.float64
    kept = (ah.astype(f) * bh + ah.astype(f) * bm + am.astype(f) * bh + am.astype(f) * bm + ah.astype(f) * bl + al.astype(f) * bh)
    exact = a.astype(f) * b.astype(f)
    dropped = np.abs(exact - kept)                    # = |am bl + al bm + al bl|
    assert (dropped <= np.abs(exact) * 2.0 ** -23).all(), float((dropped / np.abs(exact)).max())
