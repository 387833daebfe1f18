"""The arithmetic claim of the split-operand kernels (csrc/go2nn_bx3.h, csrc/go2nn_mlp3.h), restated in numpy and checked without a GPU: an fp32 value is the EXACT sum of three
bf16 planes (round-to-nearest-even at each level, the residuals need no rounding), and the six products the kernels keep differ from the full product by less than 2^-22 of it."""
import numpy as np


def bf16_rne(x):
    """fp32 -> the nearest bf16 (ties to even), returned as fp32 (v_cvt_pk_bf16_f32's rounding)"""
    u = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def planes(x):
    x = np.asarray(x, dtype=np.float32)
    h = bf16_rne(x); r = x - h            # (fp32 subtraction, as bx3_split4 does it)
    m = bf16_rne(r); s = r - m
    l = bf16_rne(s)
    return h, m, l, s - l


def test_three_bf16_planes_hold_every_bit_of_an_fp32_value():
    rng = np.random.default_rng(0)
    mags = np.exp2(rng.uniform(-60, 60, 200000)).astype(np.float32)
    x = np.concatenate([(rng.standard_normal(200000) * 3).astype(np.float32), mags * rng.choice([-1, 1], 200000).astype(np.float32),
                        np.array([0.0, -0.0, 1.0, -1.0, 1.0 + 2.0 ** -23, 1.0 - 2.0 ** -24, 3.0e38, -3.0e38, 2.0 ** -100, 0.1, 255.5, 256.0 - 2.0 ** -15], dtype=np.float32)])
    h, m, l, rest = planes(x)
    assert np.all(rest == 0.0)                                                    # nothing is left behind the third plane
    assert np.all((h.astype(np.float64) + m.astype(np.float64) + l.astype(np.float64)) == x.astype(np.float64))
    # each residual is exact in fp32 (Sterbenz-like: the plane is within half a bf16 ulp of what it rounds)
    assert np.all((x.astype(np.float64) - h.astype(np.float64)) == (x - h).astype(np.float64))
    nz = x != 0
    assert np.all(np.abs(m[nz]) <= np.abs(x[nz]) * 2.0 ** -8) and np.all(np.abs(l[nz]) <= np.abs(x[nz]) * 2.0 ** -16)


def test_the_six_kept_terms_are_within_an_fp32_rounding_of_the_product():
    rng = np.random.default_rng(1)
    a = (rng.standard_normal(100000) * np.exp2(rng.uniform(-10, 10, 100000))).astype(np.float32)
    b = (rng.standard_normal(100000) * np.exp2(rng.uniform(-10, 10, 100000))).astype(np.float32)
    ah, am, al, _ = (v.astype(np.float64) for v in planes(a)); bh, bm, bl, _ = (v.astype(np.float64) for v in planes(b))
    kept = ah * bh + ah * bm + am * bh + am * bm + ah * bl + al * bh           # go2nn_bx3.h: hi hi + hi mid + mid hi + mid mid + hi lo + lo hi
    full = a.astype(np.float64) * b.astype(np.float64)
    dropped = np.abs(full - kept)                                               # = |mid lo + lo mid + lo lo|
    assert np.all(dropped <= np.abs(full) * 2.0 ** -22)                         # below half an ulp of the fp32 product (2^-24 relative) times 4 at worst
    assert np.median(dropped[full != 0] / np.abs(full[full != 0])) < 2.0 ** -25
