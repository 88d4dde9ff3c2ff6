"""CPU: the error bound the split-fp16 convolution mode is built on (amt_conv_f16x3.h), checked
with numpy's IEEE half type standing where v_cvt_f16_f32 / the f16 MFMA operands do.

    x 2^s = h + l 2^-11,  h = f16(x 2^s),  l = f16((x 2^s - h) 2^11)
  * the pair represents x to <= max(2^-23 |x|, 2^-25): the f32 rounding unit, with an absolute floor
    where l leaves the half normal range (operands are scaled so that their bound sits near 2^13:
    the floor is 2^-38 of the bound);
  * h*h', h*l', l*h' are exact in f32 (11 x 11 significand bits), so a product a b evaluated as
    hh' + 2^-11 (hl' + lh') differs from the exact product by the two representation errors
    (<= 2^-23 each) and the dropped ll' term (<= 2^-22): <= 2^-21 |ab| in the worst case, ~1e-7 rms --
    a few f32 ulps per product, and a K = 2048 contraction accumulated in f32 ends up as close to the
    float64 result as a plain f32 dot product.
"""
import numpy as np


def split(x):
    x = np.asarray(x, np.float32)
    h = x.astype(np.float16)
    h = np.where(np.abs(x) >= np.float32(2.0 ** -14), h, np.float16(0))        # host flush (weights)
    r = (x - h.astype(np.float32)) * np.float32(2048.0)
    l = r.astype(np.float16)
    l = np.where(np.abs(r) >= np.float32(2.0 ** -14), l, np.float16(0))
    return h, l


def test_pair_represents_f32_to_2pow23():
    rng = np.random.default_rng(0)
    for scale in (1.0, 8.0, 1e-2, 4000.0):
        x = (rng.standard_normal(200000) * scale).astype(np.float32)
        x = x[np.abs(x) < 8192.0]                           # the kernel scales operands below 2^13
        h, l = split(x)
        rec = h.astype(np.float64) + l.astype(np.float64) / 2048.0
        x64 = x.astype(np.float64)
        err = np.abs(rec - x64)
        assert np.all(err <= np.maximum(2.0 ** -23 * np.abs(x64), 2.0 ** -25) * 1.0001), scale
        big = np.abs(x) >= 2.0 ** -3                        # l stays normal: the bound is purely relative
        if big.any():
            assert (err[big] / np.abs(x64[big])).max() <= 2.0 ** -23 * 1.0001
        # x 2^s - h is exactly representable (Sterbenz): the residual is formed without rounding
        r32 = (x - h.astype(np.float32))
        assert np.array_equal(r32.astype(np.float64), x.astype(np.float64) - h.astype(np.float64))


def test_three_term_product_error():
    rng = np.random.default_rng(1)
    a = (rng.standard_normal(100000) * 3).astype(np.float32)
    b = (rng.standard_normal(100000) * 9).astype(np.float32)
    keep = (np.abs(a) >= 0.125) & (np.abs(b) >= 0.125)      # away from the absolute floor of the low terms
    a, b = a[keep], b[keep]
    ah, al = split(a)
    bh, bl = split(b)
    f = np.float64
    # every partial product is exact in f32
    for p, q in ((ah, bh), (ah, bl), (al, bh)):
        prod64 = p.astype(f) * q.astype(f)
        assert np.array_equal((p.astype(np.float32) * q.astype(np.float32)).astype(f), prod64)
    approx = ah.astype(f) * bh.astype(f) + (ah.astype(f) * bl.astype(f) + al.astype(f) * bh.astype(f)) / 2048.0
    exact = a.astype(f) * b.astype(f)
    rel = np.abs(approx - exact) / np.abs(exact)
    assert rel.max() <= 2.0 ** -21, rel.max()
    assert np.sqrt(np.mean(rel ** 2)) <= 1.2e-7


def test_dot_product_as_accurate_as_f32():
    """A K = 2048 contraction (the dominant layer's depth) evaluated with the three-term products and
    f32 accumulation is as close to the float64 result as a plain f32 dot product is."""
    rng = np.random.default_rng(2)
    K = 2048
    errs_split, errs_f32 = [], []
    for _ in range(200):
        a = rng.random(K).astype(np.float32)                # sigmoid-like activations
        b = (rng.standard_normal(K) * 0.05).astype(np.float32)
        ah, al = split(a * np.float32(4096.0))              # power-of-two operand scales, undone below
        bh, bl = split(b * np.float32(128.0))
        hi = np.float32(0)
        lo = np.float32(0)
        p_hh = ah.astype(np.float32) * bh.astype(np.float32)
        p_x = ah.astype(np.float32) * bl.astype(np.float32) + al.astype(np.float32) * bh.astype(np.float32)
        hi = np.add.reduce(p_hh, dtype=np.float32)
        lo = np.add.reduce(p_x, dtype=np.float32)
        got = (np.float32(hi) + np.float32(lo) * np.float32(1.0 / 2048.0)) * np.float32(1.0 / (4096.0 * 128.0))
        ref = float(np.dot(a.astype(np.float64), b.astype(np.float64)))
        f32 = float(np.add.reduce(a * b, dtype=np.float32))
        scale = float(np.abs(a.astype(np.float64) * b.astype(np.float64)).sum())
        errs_split.append(abs(float(got) - ref) / scale)
        errs_f32.append(abs(f32 - ref) / scale)
    assert np.mean(errs_split) <= 2.0 * np.mean(errs_f32) + 1e-9
    assert max(errs_split) < 1e-6
