"""CPU check of the arithmetic behind the split-operand kernels (csrc/split_bf16.h, mlp2_split_kernel, gram_split_kernel): an fp32
number as three bf16 pieces, and what the six piece products that are issued leave out.  numpy only - the kernels themselves are
measured against fp64 in the GPU suite."""
import numpy as np


def bf16_rne(x):
    """fp32 -> the nearest bf16 (ties to even), returned as fp32 - what v_cvt_pk_bf16_f32 does for finite inputs"""
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    r = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return r.astype(np.uint32).view(np.float32)


def split3(x):
    x = np.asarray(x, np.float32)
    h = bf16_rne(x)
    r = (x - h).astype(np.float32)   # exact in fp32 (the test below checks that)
    m = bf16_rne(r)
    s = (r - m).astype(np.float32)
    return h, m, bf16_rne(s), r, s


def _samples(rng, n):
    mant = rng.standard_normal(n).astype(np.float32)
    x = (mant * np.float32(2.0) ** rng.integers(-60, 60, n).astype(np.float32)).astype(np.float32)
    special = np.array([1.0, -1.0, 1.0 + 2.0 ** -8, 1.0 + 2.0 ** -9, 1.0 - 2.0 ** -9, 1.0 + 2.0 ** -8 + 2.0 ** -16, 1.0 + 2.0 ** -23,
                        3.0e38, -3.0e38, 1.5e-30, 0.0, 255.0, 257.0, 65535.0, 16777215.0], np.float32)
    return np.concatenate([x, special])


def test_three_bf16_pieces_carry_every_bit_of_an_fp32_number():
    x = _samples(np.random.default_rng(1), 200000)
    h, m, l, r, s = split3(x)
    x64, h64, m64 = x.astype(np.float64), h.astype(np.float64), m.astype(np.float64)
    assert np.array_equal(r.astype(np.float64), x64 - h64)         # the subtractions are exact in fp32
    assert np.array_equal(s.astype(np.float64), x64 - h64 - m64)
    total = h64 + m64 + l.astype(np.float64)
    miss = np.abs(total - x64)
    # 8 + 8 + 8 significand bits: the pieces sum to x, or (signs of the residuals permitting a 25th bit) to x within half a unit of
    # its LAST place times 2^-1 - never more than 2^-25 |x|
    assert np.all(miss <= np.abs(x64) * 2.0 ** -25)
    assert np.mean(miss == 0) > 0.95
    # magnitudes of the pieces: what makes the weights of the nine piece products 1, 2^-8, 2^-16, ...
    ok = x != 0
    assert np.all(np.abs(r[ok]) <= np.abs(x[ok]) * np.float32(2.0 ** -8)) and np.all(np.abs(s[ok]) <= np.abs(x[ok]) * np.float32(2.0 ** -16))


def test_the_six_issued_piece_products_leave_out_less_than_one_fp32_rounding():
    rng = np.random.default_rng(2)
    x, w = _samples(rng, 100000), _samples(rng, 100000)
    keep = (np.abs(x) < 1e18) & (np.abs(w) < 1e18) & (np.abs(x) > 1e-12) & (np.abs(w) > 1e-12)  # (piece products inside fp32's normal range)
    x, w = x[keep], w[keep]
    xh, xm, xl, _, _ = split3(x)
    wh, wm, wl, _, _ = split3(w)
    f = lambda a: a.astype(np.float64)  # noqa: E731
    issued = f(xl) * f(wh) + f(xh) * f(wl) + f(xm) * f(wm) + f(xm) * f(wh) + f(xh) * f(wm) + f(xh) * f(wh)
    exact = f(x) * f(w)
    assert np.all(np.abs(issued - exact) <= np.abs(exact) * 2.0 ** -23)   # dropped: x_m w_l + x_l w_m + x_l w_l (+ the 25th bits)
    # every issued product is exact as an fp32 accumulator input: two 8-bit significands
    for a, b in ((xl, wh), (xh, wl), (xm, wm), (xm, wh), (xh, wm), (xh, wh)):
        p = f(a) * f(b)
        assert np.array_equal(p.astype(np.float32).astype(np.float64), p)


def test_a_split_dot_product_is_no_further_from_fp64_than_the_fp32_chain():
    """the kernels' claim in numbers a CPU can check: 500-term dot products, the six piece products accumulated in fp32 per 32 k
    (an MFMA's k) then across chunks, against numpy's k-ordered fp32 chain"""
    rng = np.random.default_rng(3)
    n, k = 400, 500
    a = (rng.standard_normal((n, k)) * 10.0 ** rng.uniform(-4, 2, (n, k))).astype(np.float32)
    w = rng.standard_normal(k).astype(np.float32)
    ref = a.astype(np.float64) @ w.astype(np.float64)
    mag = np.abs(a).astype(np.float64) @ np.abs(w).astype(np.float64)
    chain = np.zeros(n, np.float32)
    for j in range(k):
        chain = (chain + a[:, j] * w[j]).astype(np.float32)   # (unfused multiply-add: at least as many roundings as the fma chain)
    ah, am, al, _, _ = split3(a)
    wh, wm, wl, _, _ = split3(w)
    acc = np.zeros(n, np.float32)
    for c0 in range(0, k, 32):
        sl = slice(c0, min(k, c0 + 32))
        for x_, w_ in ((al, wh), (ah, wl), (am, wm), (am, wh), (ah, wm), (ah, wh)):
            part = (x_[:, sl].astype(np.float64) @ w_[sl].astype(np.float64))   # inside an instruction: wider than fp32
            acc = (acc.astype(np.float64) + part).astype(np.float32)
    err_split, err_chain = np.abs(acc.astype(np.float64) - ref) / mag, np.abs(chain.astype(np.float64) - ref) / mag
    assert err_split.max() <= err_chain.max() and err_split.mean() <= err_chain.mean()
