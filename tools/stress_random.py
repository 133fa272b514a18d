"""tools/stress_random.py -- randomized parity stress on the GPU box (beyond the fixed pytest cases): for --seconds, random (n = 32 ... 32768 --
the wave-packed kernels of the small sizes included since round 4 --, modulus size 17 ... 62 bits, 1-3 primes, launch size 1 ... 5,000 frames -- on both
sides of the forward-companion threshold and of every frames-per-wave boundary --, in / out of place, dense or [poly][prime][n] strided, inputs anywhere
in [0,4q)): forward against the oracle, inverse round trip, the fused product by X (a negacyclic shift) and the square of X^(n/2) (= -1) computed in place
(a == b == c).  Uses the oracle as checker, so it is test tooling, not product code.  Round 3: 1,913 cases in 150 s, all bit-exact.
Usage: python3 tools/stress_random.py [seconds]"""
import sys, time
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import agilex_ntt_amd as agx
from oracle import oracle as orc
from gpu_util import DeviceHelper, rand_coeffs
orc.build()
dev = DeviceHelper(torch)
rng = np.random.default_rng(int(time.time()))
t0 = time.time(); cases = 0
SECONDS = float(sys.argv[1]) if len(sys.argv) > 1 else 150.0
while time.time() - t0 < SECONDS:
    n = 1 << int(rng.integers(5, 16))
    bits = int(rng.choice([17, 20, 28, 30, 31, 33, 45, 58, 59, 60, 61, 62]))
    if bits <= n.bit_length() + 1: bits = n.bit_length() + 3
    primes = int(rng.integers(1, 4))
    frames_target = int(rng.choice([1, 3, 17, 65, 300, 1100, 4200, 5000]))
    batch = max(1, min(frames_target // primes, (96 << 20) // (8 * n * primes)))
    qs = []
    k = 0
    while len(qs) < primes:
        try:
            q = orc.find_prime(bits, n, k)
        except Exception:
            break
        if q < 2 * n or (q - 1) % (2 * n) or q in qs: break
        qs.append(q); k += 1
    if len(qs) < primes: continue
    psis = [orc.min_root(q, n) for q in qs]
    plan = agx.Plan(n, qs, psi=psis)
    x = np.concatenate([rand_coeffs(rng, batch * n, q, hi_mult=int(rng.integers(1, 5)) if bits < 62 else 3) for q in qs])
    d = dev.to_device(x)
    oop = bool(rng.integers(0, 2))
    strided = bool(rng.integers(0, 4) == 0)      # a quarter of the cases: the same frames laid out [poly][prime][n] (prime_stride = n, poly_stride = P n)
    if strided:
        xs = np.ascontiguousarray(x.reshape(primes, batch, n).transpose(1, 0, 2)).reshape(-1)
        d = dev.to_device(xs)
        o = dev.empty(x.size) if oop else d
        plan.forward_strided(d.data_ptr(), o.data_ptr(), batch, n, primes * n, dev.stream)
        got = np.ascontiguousarray(dev.to_host(o).reshape(batch, primes, n).transpose(1, 0, 2)).reshape(-1)
        o = dev.to_device(got)      # the dense form of the result for the inverse below
    else:
        o = dev.empty(x.size) if oop else d
        plan.forward(d.data_ptr(), o.data_ptr(), batch, dev.stream)
        got = dev.to_host(o)
    for p, q in enumerate(qs):
        tw, pre = orc.make_tables(q, psis[p], n)
        sl = slice(p * batch * n, (p + 1) * batch * n)
        want = orc.forward_mt(x[sl].copy(), q, tw, pre, n, 16) if hasattr(orc, "forward_mt") and batch > 64 else orc.forward(x[sl], q, tw, pre, n)
        assert np.array_equal(got[sl], want), ("fwd", n, bits, primes, batch, oop)
    plan.inverse(o.data_ptr(), o.data_ptr(), batch, dev.stream)
    back = dev.to_host(o)
    for p, q in enumerate(qs):
        sl = slice(p * batch * n, (p + 1) * batch * n)
        assert np.array_equal(back[sl], x[sl] % np.uint64(q)), ("inv", n, bits, primes, batch)
    # product linearity check: (a*b) with b = delta_1 (X) shifts a negacyclically
    if batch <= 300:
        a = np.concatenate([rand_coeffs(rng, batch * n, q) for q in qs])
        b = np.zeros_like(a); b[1::n] = 1
        da, db, dc = dev.to_device(a), dev.to_device(b), dev.empty(a.size)
        plan.polymul(da.data_ptr(), db.data_ptr(), dc.data_ptr(), 0, batch, dev.stream)      # no scratch: every size from 32 up has a one-launch product
        c = dev.to_host(dc).reshape(-1, n); ar = a.reshape(-1, n)
        for p, q in enumerate(qs):
            rows = slice(p * batch, (p + 1) * batch)
            want = np.concatenate([(np.uint64(q) - ar[rows, -1:]) % np.uint64(q), ar[rows, :-1]], axis=1)
            assert np.array_equal(c[rows], want), ("mul", n, bits, primes, batch)
        # squaring in place: (X^(n/2))^2 = X^n = -1
        h = np.zeros_like(a); h[n // 2::n] = 1
        dh = dev.to_device(h)
        plan.polymul(dh.data_ptr(), dh.data_ptr(), dh.data_ptr(), 0, batch, dev.stream)
        sq = dev.to_host(dh).reshape(-1, n)
        for p, q in enumerate(qs):
            rows = slice(p * batch, (p + 1) * batch)
            assert (sq[rows, 0] == np.uint64(q - 1)).all() and not sq[rows, 1:].any(), ("square", n, bits, primes, batch)
    plan.close(); cases += 1
print("stress OK:", cases, "cases in", round(time.time() - t0), "s")
