"""CPU tests of the oracle itself: the checker must be pinned before it checks anything.

Pins: (i) SURVEY.md 8c anchors recorded from the reference's own kernel code,
(ii) an independent O(n^2) evaluation of the closed-form contract, (iii) structured
known answers derived from the contract, (iv) the committed golden vectors."""
import json
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def test_survey_anchors(orc):
    for a in _load("survey_anchors.json")["anchors"]:
        n, q, psi = a["n"], int(a["q"]), int(a["psi"])
        assert orc.find_prime(a["bits"], n) == q
        assert orc.min_root(q, n) == psi
        if a["out_first"]:
            tw, pre = orc.make_tables(q, psi, n)
            x = orc.fill_splitmix(n * a["frames"], 42, q)
            y = orc.forward(x, q, tw, pre, n)
            assert [int(v) for v in y[:len(a["out_first"])]] == [int(v) for v in a["out_first"]]


def test_golden_vectors(orc):
    for c in _load("forward_vectors.json")["cases"]:
        n, q, psi = c["n"], int(c["q"]), int(c["psi"])
        tw, pre = orc.make_tables(q, psi, n)
        x = orc.fill_splitmix(n * c["frames"], c["seed"], q)
        if "input" in c:
            assert [int(v) for v in x] == [int(v) for v in c["input"]]
        y = orc.forward(x, q, tw, pre, n)
        assert "%016x" % orc.fnv1a_words(y) == c["fnv1a_words"]
        assert [int(v) for v in y[:8]] == [int(v) for v in c["first8"]]
        assert [int(v) for v in y[-8:]] == [int(v) for v in c["last8"]]
        if "output" in c:
            assert [int(v) for v in y] == [int(v) for v in c["output"]]


@pytest.mark.parametrize("n,bits", [(2, 20), (4, 30), (8, 61), (32, 30), (64, 60), (256, 61), (1024, 30), (2048, 60), (8192, 61)])
def test_forward_matches_naive(orc, n, bits):
    q = orc.find_prime(bits, n)
    psi = orc.min_root(q, n)
    tw, pre = orc.make_tables(q, psi, n)
    rng = np.random.default_rng(n * 7 + bits)
    x = rng.integers(0, q, size=n, dtype=np.uint64)
    assert np.array_equal(orc.forward(x, q, tw, pre, n), orc.naive_forward(x, q, psi, n))


def test_table_contract(orc):
    n, q = 64, orc.find_prime(60, 64)
    psi = orc.min_root(q, n)
    tw, pre = orc.make_tables(q, psi, n)
    for j in range(n):
        e = int("{:06b}".format(j)[::-1], 2)
        assert int(tw[j]) == pow(psi, e, q)
        assert int(pre[j]) == (int(tw[j]) << 64) // q
    assert pow(psi, n, q) == q - 1 and (q - 1) % (2 * n) == 0 and orc.is_prime(q)


@pytest.mark.parametrize("n,bits", [(32, 30), (1024, 30), (4096, 60)])
def test_structured_known_answers(orc, n, bits):
    q = orc.find_prime(bits, n)
    psi = orc.min_root(q, n)
    tw, pre = orc.make_tables(q, psi, n)
    lg = n.bit_length() - 1
    zero = np.zeros(n, dtype=np.uint64)
    assert not orc.forward(zero, q, tw, pre, n).any()
    delta = zero.copy(); delta[0] = 1
    assert (orc.forward(delta, q, tw, pre, n) == 1).all()
    xmono = zero.copy(); xmono[1] = 1
    y = orc.forward(xmono, q, tw, pre, n)
    for k in range(0, n, max(1, n // 64)):
        assert int(y[int("{:0{w}b}".format(k, w=lg)[::-1], 2)]) == pow(psi, 2 * k + 1, q)
    allmax = np.full(n, q - 1, dtype=np.uint64)
    y = orc.forward(allmax, q, tw, pre, n)
    assert (y < q).all()
    # all q-1 = -1 * (all ones): NTT is linear
    ones = orc.forward(np.ones(n, dtype=np.uint64), q, tw, pre, n)
    assert np.array_equal((y.astype(object) + ones.astype(object)) % q, np.zeros(n, dtype=object))


def test_lazy_inputs_reduce_identically(orc):
    """inputs anywhere in [0,4q) give the transform of their residues (ntt.cpp:331-332)"""
    n = 256
    q = orc.find_prime(60, n)
    psi = orc.min_root(q, n)
    tw, pre = orc.make_tables(q, psi, n)
    rng = np.random.default_rng(5)
    x = rng.integers(0, q, size=n, dtype=np.uint64)
    k = rng.integers(0, 4, size=n, dtype=np.uint64)
    lifted = x + k * np.uint64(q)
    assert np.array_equal(orc.forward(lifted, q, tw, pre, n), orc.forward(x, q, tw, pre, n))


def test_in2_supplies_upper_half(orc):
    """frame = in[0..n/2) || in2[n/2..n)  (ntt.cpp:584-590)"""
    n = 64
    q = orc.find_prime(30, n)
    psi = orc.min_root(q, n)
    tw, pre = orc.make_tables(q, psi, n)
    rng = np.random.default_rng(9)
    a = rng.integers(0, q, size=2 * n, dtype=np.uint64)
    b = rng.integers(0, q, size=2 * n, dtype=np.uint64)
    merged = a.copy()
    for f in range(2):
        merged[f * n + n // 2:(f + 1) * n] = b[f * n + n // 2:(f + 1) * n]
    assert np.array_equal(orc.forward(a, q, tw, pre, n, x2=b), orc.forward(merged, q, tw, pre, n))


@pytest.mark.parametrize("n,bits", [(2, 20), (64, 30), (1024, 60), (4096, 61)])
def test_inverse_round_trip(orc, n, bits):
    q = orc.find_prime(bits, n)
    psi = orc.min_root(q, n)
    tw, pre = orc.make_tables(q, psi, n)
    itw, _ = orc.make_inv_tables(q, psi, n)
    rng = np.random.default_rng(n)
    x = rng.integers(0, q, size=3 * n, dtype=np.uint64)
    y = orc.forward(x, q, tw, pre, n)
    assert np.array_equal(orc.inverse(y, q, itw, n), x)


@pytest.mark.parametrize("n,bits", [(8, 30), (64, 60), (256, 61)])
def test_polymul_matches_schoolbook(orc, n, bits):
    q = orc.find_prime(bits, n)
    psi = orc.min_root(q, n)
    tw, pre = orc.make_tables(q, psi, n)
    itw, _ = orc.make_inv_tables(q, psi, n)
    rng = np.random.default_rng(n + 1)
    a = rng.integers(0, q, size=n, dtype=np.uint64)
    b = rng.integers(0, q, size=n, dtype=np.uint64)
    c = orc.inverse(orc.pointwise(orc.forward(a, q, tw, pre, n), orc.forward(b, q, tw, pre, n), q), q, itw, n)
    assert np.array_equal(c, orc.schoolbook(a, b, q, n))


def test_forward_mt_equals_single_thread(orc):
    n = 1024
    q = orc.find_prime(60, n)
    psi = orc.min_root(q, n)
    tw, pre = orc.make_tables(q, psi, n)
    x = orc.fill_splitmix(n * 37, 7, q)
    assert np.array_equal(orc.forward_mt(x, q, tw, pre, n, 5), orc.forward(x, q, tw, pre, n))


def test_quotient_estimate_of_the_final_reduction_is_never_high_and_at_most_one_low():
    """The arithmetic of reduce_final_est (csrc/modarith.hpp) restated with numpy float32: for q >= 2^58 and v in [0,16q),
    k' = trunc(float(v >> 32) * est_inv) with est_inv = a float slightly below 2^32/q (agx_ntt.cpp: scaled by 1 - 2^-20,
    rounded toward zero) must satisfy floor(v/q) - 1 <= k' <= floor(v/q), so that v - k'q lies in [0,2q).  Checked at every
    multiple of q +-1, at the ends of the range and on random values, for moduli at both ends of [2^58, 2^60)."""
    import numpy as np

    rng = np.random.default_rng(5)
    for q in (2**58 + 2**13 + 1, 2**59 - 2**20 + 1, 1152921504606830593, 2**60 - 2**14 + 1 - 2**15, 2**60 - 1):
        d = 4294967296.0 / q * (1.0 - 1.0 / 1048576.0)
        f = np.float32(d)
        if float(f) > d:
            f = np.nextafter(f, np.float32(0))
        vs = [0, 1, 16 * q - 1, 2**64 - 1 if 16 * q > 2**64 - 1 else 16 * q - 2]
        for k in range(1, 16):
            vs += [k * q - 1, k * q, k * q + 1, k * q + (q >> 1)]
        vs += [int(x) for x in rng.integers(0, 16 * q, size=20000, dtype=np.uint64)] if 16 * q < 2**64 else [int(rng.integers(0, 2**63)) * 2 + int(rng.integers(0, 2)) for _ in range(20000)]
        vs = [v for v in vs if 0 <= v < min(16 * q, 2**64)]
        hi = np.array([v >> 32 for v in vs], dtype=np.uint32)
        kq = (hi.astype(np.float32) * f).astype(np.uint32)          # v_cvt_f32_u32 (RNE), v_mul_f32 (RNE), v_cvt_u32_f32 (truncate)
        for v, k1 in zip(vs, kq.tolist()):
            k = v // q
            assert k - 1 <= k1 <= k, (q, v, k, k1)
