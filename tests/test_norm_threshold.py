"""CPU tier: the zero-normal test of the kernels without a square root (gi_core.hpp: f32_norm3_below_eps).
`f32_norm3(a, b, c) < SP_EPSILON_FLOAT` -- float32 sum of squares, square root in double, compared with 1e-5 -- is the same
predicate as `sum of squares < 0x2EDBE6FF` (as float32): the square root is monotone and correctly rounded."""
import math

import numpy as np

LIM = np.uint32(0x2EDBE6FF).view(np.float32)


def test_threshold_is_the_first_float_whose_root_reaches_eps():
    assert math.sqrt(float(LIM)) >= 1e-5
    assert math.sqrt(float(np.nextafter(LIM, np.float32(0)))) < 1e-5


def test_both_predicates_agree_around_the_threshold_and_on_random_vectors():
    bits = np.arange(int(LIM.view(np.uint32)) - 200000, int(LIM.view(np.uint32)) + 200000, dtype=np.uint32)
    sq = bits.view(np.float32)
    assert np.array_equal(np.sqrt(sq.astype(np.float64)) < 1e-5, sq < LIM)
    rng = np.random.default_rng(0)
    v = (rng.normal(size=(400000, 3)) * 10.0 ** rng.uniform(-8, 1, (400000, 1))).astype(np.float32)
    s = v[:, 0] * v[:, 0] + v[:, 1] * v[:, 1] + v[:, 2] * v[:, 2]  # float32 products and sum, like the kernel
    assert s.dtype == np.float32
    assert np.array_equal(np.sqrt(s.astype(np.float64)) < 1e-5, s < LIM)
    assert 1000 < (s < LIM).sum() < 399000
