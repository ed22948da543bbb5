"""The draw contract (include/recogym_rng.h): Philox4x32-10 known answers, the numpy restatement
used to inject draws into the reference, and the oracle's MT19937 legacy sampling vs numpy."""
import ctypes as C

import numpy as np
from numpy.random.mtrand import RandomState

import ref_harness as rh
from oracle import oracle as orc

# Random123 kat_vectors, philox4x32 with 10 rounds: (counter, key) -> output
KAT = [
    ((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
     (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


def test_philox_known_answers():
    for c, k, want in KAT:
        assert tuple(int(x) for x in orc.philox(c, k)) == want
        assert rh.philox4x32_10(c, k) == want


def test_python_restatement_matches_c_header():
    rs = np.random.RandomState(0)
    L = orc.lib()
    for _ in range(200):
        c = rs.randint(0, 2 ** 32, size=4, dtype=np.uint64)
        k = rs.randint(0, 2 ** 32, size=2, dtype=np.uint64)
        w = orc.philox(c, k)
        assert tuple(int(x) for x in w) == rh.philox4x32_10(c, k)
        a, b = int(w[0]), int(w[1])
        assert L.rgo_uniform(a, b) == rh.uniform(a, b)
        for n in (1, 2, 3, 10, 1000, 99999, 2 ** 29 - 1):
            assert L.rgo_bounded(a, b, n) == rh.bounded(a, b, n)
            assert rh.bounded(a, b, n) < n


def test_uniform_range_and_resolution():
    L = orc.lib()
    assert L.rgo_uniform(0, 0) == 0.0
    top = L.rgo_uniform(0xFFFFFFFF, 0xFFFFFFFF)
    assert top < 1.0 and top == 1.0 - 2.0 ** -53
    assert L.rgo_bounded(0xFFFFFFFF, 0xFFFFFFFF, 10) == 9
    assert L.rgo_bounded(0, 0, 10) == 0


def test_mt19937_legacy_sampling_matches_numpy():
    n = 1000
    for seed, bound in ((42, 10), (7, 1000), (123456, 100000), (0, 1), (99, 2 ** 31)):
        d = np.zeros(n); g = np.zeros(n); i = np.zeros(n, dtype=np.uint32)
        orc.lib().rgo_mt_probe(seed, bound, n, d.ctypes.data_as(C.c_void_p),
                               g.ctypes.data_as(C.c_void_p), i.ctypes.data_as(C.c_void_p))
        rs = RandomState(seed)
        assert (d == rs.random_sample(n)).all()
        assert (g == rs.normal(size=n)).all()
        want = np.array([rs.choice(bound) for _ in range(n)])
        assert (i.astype(np.int64) == want).all()


def test_ff_range():
    # ctr in (0.004478, 0.029312), ff(0) = 0.009422
    L = orc.lib()
    assert abs(L.rgo_ff(0.0) - 0.009422) < 1e-6
    assert abs(L.rgo_ff(-1e3) - 0.004478) < 1e-6
    assert abs(L.rgo_ff(1e3) - 0.029312) < 1e-6
