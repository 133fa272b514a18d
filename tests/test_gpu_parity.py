"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI,
must be bit-exact against the CPU oracle on the same inputs."""
import json
import os

import numpy as np
import pytest

from gpu_util import rand_coeffs, tables_for

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ALL_SIZES = [2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768]


# registry ids of the product library (what a default or a call-shape selector can reach); every other id is an A/B entry that
# only lib/libagxntt_diag.so carries: tests/test_gpu_diag.py re-runs the id-parametrised tests of this file in a child process
# bound to that library
PRODUCT_IDS = ({93, 92, 91, 159, 164, 117, 119, 120, 121, 122, 123} | set(range(150, 159)) | set(range(130, 142))
               | set(range(200, 215)) | set(range(230, 235)) | set(range(240, 245)) | set(range(250, 258)) | set(range(260, 268)))


def _select(agx, plan, config):
    """explicit registry entry (AGX_VARIANT_REGBLOCK_BASE + id); A/B ids are skipped unless the diag library is loaded"""
    if config is None or config == "default":
        return
    if config not in PRODUCT_IDS and not agx.LIB_PATH.endswith("libagxntt_diag.so"):
        plan.close()
        pytest.skip(f"registry id {config} lives in lib/libagxntt_diag.so (covered by tests/test_gpu_diag.py)")
    plan.set_variant(agx.VARIANT_REGBLOCK_BASE + config)


def _plan_from_oracle_tables(agx, orc, n, bits, count, inverse=True):
    tabs = tables_for(orc, n, bits, count)
    tw = np.stack([t[2] for t in tabs])
    pre = np.stack([t[3] for t in tabs])
    tables = [tw, pre]
    if inverse:
        inv = [orc.make_inv_tables(t[0], t[1], n) for t in tabs]
        tables += [np.stack([i[0] for i in inv]), np.stack([i[1] for i in inv])]
    return agx.Plan(n, [t[0] for t in tabs], tables=tuple(tables)), tabs


def _oracle_forward_rns(orc, x, tabs, n, batch):
    """x: [P][batch][n] flat"""
    out = np.empty_like(x)
    for p, (q, _, tw, pre) in enumerate(tabs):
        sl = slice(p * batch * n, (p + 1) * batch * n)
        out[sl] = orc.forward(x[sl], q, tw, pre, n)
    return out


@pytest.mark.parametrize("variant", ["radix2", "regblock"])
@pytest.mark.parametrize("n", ALL_SIZES)
def test_forward_bit_exact(agx, orc, dev, n, variant):
    bits = 30 if n == 1024 else (61 if n in (8, 8192) else 60)
    batch = 5 if n <= 4096 else 3          # ragged: not a multiple of polys-per-block
    primes = 2 if n <= 8192 else 1
    plan, tabs = _plan_from_oracle_tables(agx, orc, n, bits, primes)
    plan.set_variant(agx.VARIANT_LDS_RADIX2 if variant == "radix2" else agx.VARIANT_REGBLOCK)
    rng = np.random.default_rng(n * 3 + primes)
    x = np.concatenate([rand_coeffs(rng, batch * n, t[0], hi_mult=4) for t in tabs])  # lazy range [0,4q)
    d_in = dev.to_device(x)
    d_out = dev.empty(x.size)
    plan.forward(d_in.data_ptr(), d_out.data_ptr(), batch, dev.stream)
    got = dev.to_host(d_out)
    want = _oracle_forward_rns(orc, x, tabs, n, batch)
    assert np.array_equal(got, want)
    assert np.array_equal(dev.to_host(d_in), x), "input must not be modified"
    # in place
    plan.forward(d_in.data_ptr(), d_in.data_ptr(), batch, dev.stream)
    assert np.array_equal(dev.to_host(d_in), want)
    plan.close()


def test_golden_vectors_on_gpu(agx, orc, dev):
    with open(os.path.join(GOLDEN, "forward_vectors.json")) as f:
        cases = json.load(f)["cases"]
    for c in cases:
        n, q, psi = c["n"], int(c["q"]), int(c["psi"])
        plan = agx.Plan(n, [q], psi=[psi])
        assert plan.psi(0) == psi
        x = orc.fill_splitmix(n * c["frames"], c["seed"], q)
        d = dev.to_device(x)
        plan.forward(d.data_ptr(), d.data_ptr(), c["frames"], dev.stream)
        y = dev.to_host(d)
        assert "%016x" % orc.fnv1a_words(y) == c["fnv1a_words"], c["n"]
        assert [int(v) for v in y[:8]] == [int(v) for v in c["first8"]]
        assert [int(v) for v in y[-8:]] == [int(v) for v in c["last8"]]
        if "output" in c:
            assert [int(v) for v in y] == [int(v) for v in c["output"]]
        plan.close()


def test_survey_anchor_words_on_gpu(agx, orc, dev):
    """the output words SURVEY.md 8c recorded from the reference's own kernel code"""
    with open(os.path.join(GOLDEN, "survey_anchors.json")) as f:
        anchors = json.load(f)["anchors"]
    for a in anchors:
        if not a["out_first"]:
            continue
        n, q = a["n"], int(a["q"])
        plan = agx.Plan(n, [q])  # library picks the least root itself
        assert plan.psi(0) == int(a["psi"])
        x = orc.fill_splitmix(n * a["frames"], 42, q)
        d = dev.to_device(x)
        plan.forward(d.data_ptr(), d.data_ptr(), a["frames"], dev.stream)
        y = dev.to_host(d)
        assert [int(v) for v in y[:len(a["out_first"])]] == [int(v) for v in a["out_first"]]
        plan.close()


@pytest.mark.parametrize("n,bits,frames", [(32, 30, 2), (1024, 30, 1), (1024, 30, 7), (4096, 60, 3), (16384, 60, 2), (32768, 60, 1)])
def test_one_shot_host_call(agx, orc, n, bits, frames):
    """agx_ntt_forward_host = ntt_input_kernel + fwd_ntt_kernel<0> + ntt_output_kernel;
    lower half of each frame from `in`, upper half from `in2` (ntt.cpp:584-590)"""
    (q, psi, tw, pre), = tables_for(orc, n, bits)
    rng = np.random.default_rng(n + frames)
    a = rand_coeffs(rng, frames * n, q)
    b = rand_coeffs(rng, frames * n, q)
    got = agx.forward_host(a, b, q, tw, pre, n, frames)
    assert np.array_equal(got, orc.forward(a, q, tw, pre, n, x2=b))
    got_same = agx.forward_host(a, a, q, tw, pre, n, frames)
    assert np.array_equal(got_same, orc.forward(a, q, tw, pre, n))


def test_host_streaming_pipeline(agx, orc):
    """more frames than one staging chunk (32 MiB): chunks rotate over three slots/streams; in2
    supplies the upper halves; result identical to the oracle frame for frame"""
    n, frames = 4096, 2600                      # 81 MiB of frames -> 3 chunks (1024, 1024, 552)
    q = agx.find_primes(60, n)[0]
    plan = agx.Plan(n, [q])
    tw, pre = orc.make_tables(q, plan.psi(0), n)
    rng = np.random.default_rng(123)
    a = rand_coeffs(rng, frames * n, q)
    b = rand_coeffs(rng, frames * n, q)
    got = plan.forward_host_stream(a, b, frames)
    want = orc.forward_mt(np.concatenate([np.concatenate([a[f * n:f * n + n // 2], b[f * n + n // 2:(f + 1) * n]]) for f in range(frames)]),
                          q, tw, pre, n, 8)
    assert np.array_equal(got, want)
    same = plan.forward_host_stream(a, a, frames)
    assert np.array_equal(same, orc.forward_mt(a, q, tw, pre, n, 8))
    two = agx.Plan(n, agx.find_primes(60, n, 2))
    with pytest.raises(agx.AgxError) as ei:
        two.forward_host_stream(a, a, 1)
    assert ei.value.status == 5
    plan.close()
    two.close()


def test_reference_smoke_inputs_are_reproduced(agx, orc):
    """main.cpp:49-55 feeds placeholder tables that break the precon contract; the reference
    then computes 64-bit wrap-around garbage.  Same operation sequence here, so the
    garbage matches the oracle bit for bit as well."""
    n = 16384
    i = np.arange(n, dtype=np.uint64)
    got = agx.forward_host(i, i + np.uint64(1), 65537, i + np.uint64(2), i + np.uint64(3), n, 1)
    want = orc.forward(i, 65537, i + np.uint64(2), i + np.uint64(3), n, x2=i + np.uint64(1))
    assert np.array_equal(got, want)


@pytest.mark.parametrize("n", [32, 1024, 4096])
def test_structured_known_answers(agx, dev, n):
    q = agx.find_primes(30 if n == 1024 else 60, n)[0]
    plan = agx.Plan(n, [q])
    psi = plan.psi(0)
    lg = n.bit_length() - 1
    x = np.zeros(4 * n, dtype=np.uint64)
    x[n] = 1                 # frame 1: delta_0
    x[2 * n + 1] = 1         # frame 2: X
    x[3 * n:] = q - 1        # frame 3: all q-1
    d = dev.to_device(x)
    plan.forward(d.data_ptr(), d.data_ptr(), 4, dev.stream)
    y = dev.to_host(d).reshape(4, n)
    assert not y[0].any()
    assert (y[1] == 1).all()
    for k in range(n):
        assert int(y[2][int("{:0{w}b}".format(k, w=lg)[::-1], 2)]) == pow(psi, 2 * k + 1, q)
    assert (y[3] < q).all()
    plan.close()


@pytest.mark.parametrize("bits", [30, 60, 61, 62])
@pytest.mark.parametrize("config", ["default", 91, 92, 93, 159, 147, 161, 70])
def test_n4096_kernel_registry_variants(agx, orc, dev, bits, config):
    """every registered n=4096 kernel (exact 91, fast 92, 16q-lazy 93 and its forward companion 159; the A/B twins 147 / 161 and the
    trace twin 70 of lib/libagxntt_diag.so: tests/test_gpu_diag.py) against the oracle, for 30-, 60-, 61- and 62-bit moduli; a form
    whose lazy range does not fit the modulus must be refused, and the default must fall back to a legal one"""
    n, batch, primes = 4096, 3, 2
    plan, tabs = _plan_from_oracle_tables(agx, orc, n, bits, primes, inverse=False)
    if config != "default":
        # fast forms need q <= 2^61, the 16q-lazy form q <= 2^60: anything else must be refused
        illegal = (config == 92 and bits == 62) or (config in (93, 159, 147, 161, 70) and bits >= 61)
        if illegal:
            if config not in PRODUCT_IDS and not agx.LIB_PATH.endswith("libagxntt_diag.so"):
                plan.close()
                pytest.skip("diag-library id")
            with pytest.raises(agx.AgxError) as ei:
                plan.set_variant(agx.VARIANT_REGBLOCK_BASE + config)
            assert ei.value.status == 2
            plan.close()
            return
        _select(agx, plan, config)
    rng = np.random.default_rng(bits * 100 + (config if config != "default" else 7))
    x = np.concatenate([rand_coeffs(rng, batch * n, t[0], hi_mult=4 if bits < 62 else 3) for t in tabs])
    d = dev.to_device(x)
    plan.forward(d.data_ptr(), d.data_ptr(), batch, dev.stream)
    assert np.array_equal(dev.to_host(d), _oracle_forward_rns(orc, x, tabs, n, batch))
    plan.close()


@pytest.mark.parametrize("n,bits,config", [(4096, 60, None), (4096, 61, None), (4096, 62, None), (4096, 60, 93), (4096, 59, 93), (4096, 58, 93), (4096, 45, 93),
                                           (16384, 60, 117), (16384, 59, 117), (16384, 58, 117), (16384, 45, 117), (16384, 61, None), (16384, 62, None),
                                           (32768, 60, 119), (32768, 59, 119), (32768, 58, 119), (32768, 45, 119), (32768, 61, None), (32768, 62, None),
                                           (1024, 60, None), (2048, 60, None), (8192, 60, None)])
def test_extreme_coefficients(agx, orc, dev, n, bits, config):
    """worst-case lazy ranges: all coefficients 4q-1 / q-1 / 0 under the largest 60-, 61- and
    62-bit moduli (16q-lazy, fast and exact forms respectively); configs 93 / 117 / 119 = the kernels with the tail-free schedule
    whose outputs reach 16q before the quotient-estimate reduction (n = 4096, 16384, 32768), at 60/59-bit moduli (estimate path;
    59 bits = the smallest top words it sees), at 58 and 45 bits (q < 2^58: the four-step fallback)"""
    q = orc.find_prime(bits, n)
    psi = orc.min_root(q, n)
    tw, pre = orc.make_tables(q, psi, n)
    plan = agx.Plan(n, [q], psi=[psi])
    if config is not None:
        plan.set_variant(agx.VARIANT_REGBLOCK_BASE + config)
    top = min(4 * q - 1, 2**64 - 1)
    x = np.concatenate([np.full(n, top, dtype=np.uint64), np.full(n, q - 1, dtype=np.uint64), np.zeros(n, dtype=np.uint64),
                        np.where(np.arange(n) % 2 == 0, np.uint64(top), np.uint64(0))])
    d = dev.to_device(x)
    plan.forward(d.data_ptr(), d.data_ptr(), 4, dev.stream)
    assert np.array_equal(dev.to_host(d), orc.forward(x, q, tw, pre, n))
    plan.close()


@pytest.mark.parametrize("n", [1024, 2048, 4096, 8192, 16384, 32768])
@pytest.mark.parametrize("bits", [60, 61])
def test_inverse_extreme_inputs(agx, orc, dev, n, bits):
    """the inverse's lazy sums at their worst: every input 4q-1 (the largest value the contract admits),
    q-1, alternating 4q-1 / 0, under the largest 60-bit (16q-lazy inverse) and 61-bit (fast form) moduli;
    expected = the oracle's inverse of the residues"""
    q = orc.find_prime(bits, n)
    psi = orc.min_root(q, n)
    plan = agx.Plan(n, [q], psi=[psi])
    top = 4 * q - 1
    x = np.concatenate([np.full(n, top, dtype=np.uint64), np.full(n, q - 1, dtype=np.uint64),
                        np.where(np.arange(n) % 2 == 0, np.uint64(top), np.uint64(0)),
                        np.where(np.arange(n) % 3 == 0, np.uint64(top), np.uint64(top - q))])
    itw = orc.make_inv_tables(q, psi, n)[0]
    want = orc.inverse(x % np.uint64(q), q, itw, n)
    d = dev.to_device(x)
    plan.inverse(d.data_ptr(), d.data_ptr(), 4, dev.stream)
    assert np.array_equal(dev.to_host(d), want)
    plan.close()


@pytest.mark.parametrize("n", [4096, 16384, 32768])
@pytest.mark.parametrize("bits", [61, 62])
def test_inverse_fast_and_exact_forms(agx, orc, dev, n, bits):
    """inverse kernels in their fast (61-bit q) and exact (62-bit q) arithmetic -- at n=32768 the one-launch
    pair kernel -- against the oracle's inverse, in place and out of place, plus the round trip"""
    batch, primes = 3, 2
    plan, tabs = _plan_from_oracle_tables(agx, orc, n, bits, primes)
    rng = np.random.default_rng(n * 3 + bits)
    r = np.concatenate([rand_coeffs(rng, batch * n, t[0]) for t in tabs])
    want = np.concatenate([orc.inverse(r[p * batch * n:(p + 1) * batch * n], tabs[p][0],
                                       orc.make_inv_tables(tabs[p][0], tabs[p][1], n)[0], n) for p in range(primes)])
    d_r, d_o = dev.to_device(r), dev.empty(r.size)
    plan.inverse(d_r.data_ptr(), d_o.data_ptr(), batch, dev.stream)
    assert np.array_equal(dev.to_host(d_o), want)
    plan.inverse(d_r.data_ptr(), d_r.data_ptr(), batch, dev.stream)
    assert np.array_equal(dev.to_host(d_r), want)
    plan.forward(d_r.data_ptr(), d_r.data_ptr(), batch, dev.stream)
    assert np.array_equal(dev.to_host(d_r), r)
    plan.close()


@pytest.mark.parametrize("n", ALL_SIZES)
def test_inverse_round_trip_and_oracle(agx, orc, dev, n):
    bits = 30 if n == 1024 else 60
    batch, primes = 3, 2
    plan, tabs = _plan_from_oracle_tables(agx, orc, n, bits, primes)
    rng = np.random.default_rng(n + 11)
    x = np.concatenate([rand_coeffs(rng, batch * n, t[0]) for t in tabs])
    d_x = dev.to_device(x)
    d_y = dev.empty(x.size)
    plan.forward(d_x.data_ptr(), d_y.data_ptr(), batch, dev.stream)
    y = dev.to_host(d_y)
    d_z = dev.empty(x.size)
    plan.inverse(d_y.data_ptr(), d_z.data_ptr(), batch, dev.stream)
    assert np.array_equal(dev.to_host(d_z), x)
    # against the oracle's inverse on arbitrary (not forward-image) data
    r = np.concatenate([rand_coeffs(rng, batch * n, t[0]) for t in tabs])
    d_r = dev.to_device(r)
    plan.inverse(d_r.data_ptr(), d_r.data_ptr(), batch, dev.stream)
    want = np.concatenate([orc.inverse(r[p * batch * n:(p + 1) * batch * n], tabs[p][0],
                                       orc.make_inv_tables(tabs[p][0], tabs[p][1], n)[0], n) for p in range(primes)])
    assert np.array_equal(dev.to_host(d_r), want)
    assert np.array_equal(y, _oracle_forward_rns(orc, x, tabs, n, batch))
    plan.close()


@pytest.mark.parametrize("n,bits", [(64, 30), (1024, 60), (4096, 61)])
def test_pointwise_and_polymul(agx, orc, dev, n, bits):
    batch, primes = 3, 2
    plan, tabs = _plan_from_oracle_tables(agx, orc, n, bits, primes)
    rng = np.random.default_rng(n + 5)
    a = np.concatenate([rand_coeffs(rng, batch * n, t[0]) for t in tabs])
    b = np.concatenate([rand_coeffs(rng, batch * n, t[0]) for t in tabs])
    d_a, d_b, d_c, d_s = dev.to_device(a), dev.to_device(b), dev.empty(a.size), dev.empty(a.size)
    plan.pointwise(d_a.data_ptr(), d_b.data_ptr(), d_c.data_ptr(), batch, dev.stream)
    want = np.concatenate([orc.pointwise(a[p * batch * n:(p + 1) * batch * n], b[p * batch * n:(p + 1) * batch * n], tabs[p][0])
                           for p in range(primes)])
    assert np.array_equal(dev.to_host(d_c), want)
    plan.polymul(d_a.data_ptr(), d_b.data_ptr(), d_c.data_ptr(), d_s.data_ptr(), batch, dev.stream)
    c = dev.to_host(d_c)
    for p in range(primes):
        for f in range(batch):
            sl = slice((p * batch + f) * n, (p * batch + f + 1) * n)
            assert np.array_equal(c[sl], orc.schoolbook(a[sl], b[sl], tabs[p][0], n))
    # c aliasing a
    plan.polymul(d_a.data_ptr(), d_b.data_ptr(), d_a.data_ptr(), d_s.data_ptr(), batch, dev.stream)
    assert np.array_equal(dev.to_host(d_a), c)
    plan.close()


@pytest.mark.parametrize("n,bits", [(256, 60), (1024, 30), (4096, 60), (4096, 61), (4096, 62), (16384, 60)])
def test_lazy_outputs(agx, orc, dev, n, bits):
    """agx_ntt_forward_lazy: values in [0,4q) congruent to the transform; the inverse and the
    pointwise product take them as they are"""
    batch = 3
    plan, tabs = _plan_from_oracle_tables(agx, orc, n, bits, 1)
    q = tabs[0][0]
    rng = np.random.default_rng(n + bits)
    x = rand_coeffs(rng, batch * n, q)
    d_x, d_y = dev.to_device(x), dev.empty(x.size)
    plan.forward_lazy(d_x.data_ptr(), d_y.data_ptr(), batch, dev.stream)
    y = dev.to_host(d_y)
    want = orc.forward(x, q, tabs[0][2], tabs[0][3], n)
    assert (y.astype(object) < 4 * q).all()
    assert np.array_equal(y % np.uint64(q), want)
    d_z = dev.empty(x.size)
    plan.inverse(d_y.data_ptr(), d_z.data_ptr(), batch, dev.stream)
    assert np.array_equal(dev.to_host(d_z), x)
    plan.pointwise(d_y.data_ptr(), d_y.data_ptr(), d_z.data_ptr(), batch, dev.stream)
    assert np.array_equal(dev.to_host(d_z), orc.pointwise(want, want, q))
    plan.close()


def test_strided_poly_major_layout(agx, orc, dev):
    """[poly][prime][n] callers: prime_stride = n, poly_stride = P*n"""
    n, batch, primes = 4096, 4, 3
    plan, tabs = _plan_from_oracle_tables(agx, orc, n, 60, primes)
    rng = np.random.default_rng(77)
    x = np.empty((batch, primes, n), dtype=np.uint64)
    for p in range(primes):
        x[:, p, :] = rand_coeffs(rng, batch * n, tabs[p][0]).reshape(batch, n)
    d = dev.to_device(x.reshape(-1))
    plan.forward_strided(d.data_ptr(), d.data_ptr(), batch, n, primes * n, dev.stream)
    y = dev.to_host(d).reshape(batch, primes, n)
    for p, (q, _, tw, pre) in enumerate(tabs):
        assert np.array_equal(y[:, p, :].reshape(-1), orc.forward(np.ascontiguousarray(x[:, p, :]).reshape(-1), q, tw, pre, n))
    plan.inverse_strided(d.data_ptr(), d.data_ptr(), batch, n, primes * n, dev.stream)
    assert np.array_equal(dev.to_host(d).reshape(batch, primes, n), x)
    plan.close()


@pytest.mark.parametrize("n", [16384, 32768])
def test_large_frames_strided_in_and_out_of_place(agx, orc, dev, n):
    """n=16384/32768 have dedicated forward kernels per call shape (pair kernel in place, fused-split
    out of place); both must honour strides: [poly][prime][n] layout, 2 primes, ragged batch"""
    batch, primes = 3, 2
    plan, tabs = _plan_from_oracle_tables(agx, orc, n, 60, primes)
    rng = np.random.default_rng(n + 3)
    x = np.empty((batch, primes, n), dtype=np.uint64)
    for p in range(primes):
        x[:, p, :] = rand_coeffs(rng, batch * n, tabs[p][0], hi_mult=4).reshape(batch, n)
    want = np.empty_like(x)
    for p, (q, _, tw, pre) in enumerate(tabs):
        want[:, p, :] = orc.forward(np.ascontiguousarray(x[:, p, :]).reshape(-1), q, tw, pre, n).reshape(batch, n)
    d_in = dev.to_device(x.reshape(-1))
    d_out = dev.empty(x.size)
    plan.forward_strided(d_in.data_ptr(), d_out.data_ptr(), batch, n, primes * n, dev.stream)     # out of place
    assert np.array_equal(dev.to_host(d_out).reshape(batch, primes, n), want)
    assert np.array_equal(dev.to_host(d_in), x.reshape(-1))
    plan.forward_strided(d_in.data_ptr(), d_in.data_ptr(), batch, n, primes * n, dev.stream)      # in place
    assert np.array_equal(dev.to_host(d_in).reshape(batch, primes, n), want)
    plan.inverse_strided(d_in.data_ptr(), d_in.data_ptr(), batch, n, primes * n, dev.stream)
    assert np.array_equal(dev.to_host(d_in).reshape(batch, primes, n), x % np.array([t[0] for t in tabs], dtype=np.uint64)[None, :, None])
    plan.close()


@pytest.mark.parametrize("bits", [61, 62])
@pytest.mark.parametrize("n", [16384, 32768])
def test_large_frames_fast_and_exact_forms(agx, orc, dev, n, bits):
    """the per-call-shape forward kernels of n=16384/32768 (pair kernels in place, fused-split out of
    place) in their fast (61-bit q) and exact (62-bit q) arithmetic, against the oracle"""
    batch, primes = 3, 2
    plan, tabs = _plan_from_oracle_tables(agx, orc, n, bits, primes, inverse=False)
    rng = np.random.default_rng(n + bits)
    x = np.concatenate([rand_coeffs(rng, batch * n, t[0], hi_mult=4 if bits < 62 else 3) for t in tabs])
    want = _oracle_forward_rns(orc, x, tabs, n, batch)
    d_in = dev.to_device(x)
    d_out = dev.empty(x.size)
    plan.forward(d_in.data_ptr(), d_out.data_ptr(), batch, dev.stream)     # out of place
    assert np.array_equal(dev.to_host(d_out), want)
    plan.forward(d_in.data_ptr(), d_in.data_ptr(), batch, dev.stream)      # in place
    assert np.array_equal(dev.to_host(d_in), want)
    plan.close()


@pytest.mark.parametrize("n,batch", [(32, 1000), (512, 77), (4096, 8), (4096, 4200), (16384, 300)])
def test_calls_are_graph_capturable(agx, orc, dev, n, batch):
    """the device-pointer calls allocate nothing and never synchronise, so a stream capture can
    record them: forward + inverse captured once into a HIP graph, replayed on new data (n=16384 with more frames than
    workgroups: its inverse is a loop kernel, and a captured launch must take the stateless fixed-stride form)"""
    import torch

    q = agx.find_primes(60, n)[0]
    plan = agx.Plan(n, [q])
    tw, pre = orc.make_tables(q, plan.psi(0), n)
    buf = dev.empty(batch * n)
    mid = dev.empty(batch * n)
    out = dev.empty(batch * n)
    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        plan.forward(buf.data_ptr(), mid.data_ptr(), batch, side.cuda_stream)      # warm-up outside capture
        side.synchronize()
        with torch.cuda.graph(graph, stream=side):
            s = torch.cuda.current_stream().cuda_stream
            plan.forward(buf.data_ptr(), mid.data_ptr(), batch, s)
            plan.inverse(mid.data_ptr(), out.data_ptr(), batch, s)
    torch.cuda.current_stream().wait_stream(side)
    rng = np.random.default_rng(4)
    for _ in range(2):
        x = rand_coeffs(rng, batch * n, q)
        buf.copy_(torch.from_numpy(x.view(np.int64).copy()))
        graph.replay()
        dev.sync()
        assert np.array_equal(dev.to_host(mid), orc.forward(x, q, tw, pre, n))
        assert np.array_equal(dev.to_host(out), x)
    plan.close()


def test_forward_companion_on_both_sides_of_its_threshold(agx, orc, dev):
    """n = 4096, 60-bit moduli: forward launches of >= 4,096 frames run on the streamed 128-thread kernel (registry id 159, the forward companion
    of the R = 3 default), smaller ones on the 512-thread kernel; one plan, launch sizes on both sides of the threshold (and exactly on it, split
    over two primes), in place and out of place, every result against the oracle, inverse round trip through the main kernel"""
    n = 4096
    for primes, batch in ((1, 4095), (1, 4096), (2, 2047), (2, 2048), (3, 1400)):
        plan, tabs = _plan_from_oracle_tables(agx, orc, n, 60, primes)
        rng = np.random.default_rng(primes * 10000 + batch)
        x = np.concatenate([rand_coeffs(rng, batch * n, t[0], hi_mult=4) for t in tabs])
        want = np.concatenate([orc.forward_mt(x[p * batch * n:(p + 1) * batch * n].copy(), t[0], t[2], t[3], n, 8) for p, t in enumerate(tabs)])
        d_x, d_y = dev.to_device(x), dev.empty(x.size)
        plan.forward(d_x.data_ptr(), d_y.data_ptr(), batch, dev.stream)
        assert np.array_equal(dev.to_host(d_y), want), (primes, batch, "out of place")
        plan.forward(d_x.data_ptr(), d_x.data_ptr(), batch, dev.stream)
        assert np.array_equal(dev.to_host(d_x), want), (primes, batch, "in place")
        plan.inverse(d_x.data_ptr(), d_x.data_ptr(), batch, dev.stream)
        back = dev.to_host(d_x)
        for p, t in enumerate(tabs):
            sl = slice(p * batch * n, (p + 1) * batch * n)
            assert np.array_equal(back[sl], x[sl] % np.uint64(t[0])), (primes, batch, "round trip")
        plan.close()


def test_empty_batch_and_errors(agx, dev):
    q = agx.find_primes(60, 4096)[0]
    plan = agx.Plan(4096, [q])
    d = dev.empty(4096)
    plan.forward(d.data_ptr(), d.data_ptr(), 0, dev.stream)  # no-op, no launch
    with pytest.raises(agx.AgxError) as ei:
        plan.forward(0, d.data_ptr(), 1, dev.stream)
    assert ei.value.status == 1
    tw, pre = agx.make_tables(q, plan.psi(0), 4096)
    fwd_only = agx.Plan(4096, [q], tables=(tw[None, :], pre[None, :]))
    with pytest.raises(agx.AgxError) as ei:
        fwd_only.inverse(d.data_ptr(), d.data_ptr(), 1, dev.stream)
    assert ei.value.status == 9
    # poly-mul: every size has a one-launch kernel now; only a plan forced onto the radix-2 kernels takes the three-launch path and needs
    # caller scratch (NULL -> 1), which must be disjoint from the operands (-> 5)
    big = agx.Plan(16, [agx.find_primes(60, 16)[0]])
    big.set_variant(agx.VARIANT_LDS_RADIX2)
    e = dev.empty(512)
    for scratch, status in ((0, 1), (e.data_ptr(), 5)):
        with pytest.raises(agx.AgxError) as ei:
            big.polymul(e.data_ptr(), e.data_ptr(), e.data_ptr(), scratch, 1, dev.stream)
        assert ei.value.status == status
    # forward-only plans have no product either; pointwise / fill validate their pointers like the transforms do
    with pytest.raises(agx.AgxError) as ei:
        fwd_only.polymul(d.data_ptr(), d.data_ptr(), d.data_ptr(), 0, 1, dev.stream)
    assert ei.value.status == 9
    with pytest.raises(agx.AgxError) as ei:
        plan.pointwise(d.data_ptr(), 0, d.data_ptr(), 1, dev.stream)
    assert ei.value.status == 1
    with pytest.raises(agx.AgxError) as ei:
        plan.fill_synthetic(0, 1, 0, 42, dev.stream)
    assert ei.value.status == 1
    plan.close()
    fwd_only.close()
    big.close()


def test_one_shot_plan_cache_follows_the_callers_tables(agx, orc):
    """agx_ntt_forward_host keeps the plan of its last call: repeating a call reuses it, and a call with another root
    (same n and modulus, different tables), another modulus or another n must NOT -- each result against the oracle"""
    rng = np.random.default_rng(77)
    n = 1024
    q = orc.find_prime(30, n)
    psi1 = orc.min_root(q, n)
    psi2 = pow(psi1, 3, q)                      # another primitive 2n-th root (3 is odd)
    cases = [(n, q, psi1), (n, q, psi1), (n, q, psi2), (n, orc.find_prime(30, n, 1), None), (2048, orc.find_prime(60, 2048), None), (n, q, psi1)]
    for (nn, qq, psi) in cases:
        psi = psi if psi is not None else orc.min_root(qq, nn)
        tw, pre = orc.make_tables(qq, psi, nn)
        x = rand_coeffs(rng, 5 * nn, qq)
        got = agx.forward_host(x, x, qq, tw, pre, nn, 5)
        assert np.array_equal(got, orc.forward(x, qq, tw, pre, nn)), (nn, qq, psi)


def test_fill_synthetic_is_shard_invariant(agx, dev):
    n, primes, batch = 1024, 2, 6
    qs = agx.find_primes(60, n, primes)
    plan = agx.Plan(n, qs)
    whole = dev.empty(primes * batch * n)
    plan.fill_synthetic(whole.data_ptr(), batch, 0, 42, dev.stream)
    w = dev.to_host(whole).reshape(primes, batch, n)
    for p in range(primes):
        assert (w[p] < qs[p]).all()
    part = dev.empty(primes * 2 * n)
    plan.fill_synthetic(part.data_ptr(), 2, 3, 42, dev.stream)  # polys 3,4
    assert np.array_equal(dev.to_host(part).reshape(primes, 2, n), w[:, 3:5, :])
    assert len(np.unique(w)) > w.size * 0.99
    plan.close()


# ---------------------------------------------------------------------------------------
# BASELINE.json configurations at full size, through size-independent properties
# ---------------------------------------------------------------------------------------
def _mod_add(a, b, q):
    s = a + b  # < 2^61 + 2^61 fits
    return np.where(s >= q, s - np.uint64(q), s)


def test_config3_full_size_properties(agx, orc, dev):
    """n=4096, 4-prime RNS, batch=4096 (the roofline run): linearity, round trip, and a sample
    of frames against the oracle"""
    n, primes, batch = 4096, 4, 4096
    qs = agx.find_primes(60, n, primes)
    plan = agx.Plan(n, qs)
    total = primes * batch * n
    d_a, d_b = dev.empty(total), dev.empty(total)
    plan.fill_synthetic(d_a.data_ptr(), batch, 0, 1, dev.stream)
    plan.fill_synthetic(d_b.data_ptr(), batch, 0, 2, dev.stream)
    a, b = dev.to_host(d_a), dev.to_host(d_b)
    s = np.concatenate([_mod_add(a[p * batch * n:(p + 1) * batch * n], b[p * batch * n:(p + 1) * batch * n], qs[p]) for p in range(primes)])
    d_s = dev.to_device(s)
    d_fa, d_fb = dev.empty(total), dev.empty(total)
    plan.forward(d_a.data_ptr(), d_fa.data_ptr(), batch, dev.stream)
    plan.forward(d_b.data_ptr(), d_fb.data_ptr(), batch, dev.stream)
    plan.forward(d_s.data_ptr(), d_s.data_ptr(), batch, dev.stream)
    fa, fb, fs = dev.to_host(d_fa), dev.to_host(d_fb), dev.to_host(d_s)
    for p in range(primes):
        sl = slice(p * batch * n, (p + 1) * batch * n)
        assert (fa[sl] < qs[p]).all()
        assert np.array_equal(_mod_add(fa[sl], fb[sl], qs[p]), fs[sl]), "NTT(a+b) != NTT(a)+NTT(b)"
    plan.inverse(d_fa.data_ptr(), d_fa.data_ptr(), batch, dev.stream)
    assert np.array_equal(dev.to_host(d_fa), a), "INTT(NTT(a)) != a"
    for p in range(primes):
        tw, pre = orc.make_tables(qs[p], plan.psi(p), n)
        for f in (0, 1, 2047, 4095):
            sl = slice((p * batch + f) * n, (p * batch + f + 1) * n)
            assert np.array_equal(orc.forward(a[sl], qs[p], tw, pre, n), fa[sl])
            assert np.array_equal(orc.forward(b[sl], qs[p], tw, pre, n), fb[sl])
    plan.close()


def test_config2_single_60bit_forward_inverse(agx, orc, dev):
    """n=4096, one 60-bit modulus, batch=1 forward+inverse, bit-exact vs CPU"""
    n = 4096
    q = agx.find_primes(60, n)[0]
    plan = agx.Plan(n, [q])
    tw, pre = orc.make_tables(q, plan.psi(0), n)
    x = orc.fill_splitmix(n, 42, q)
    d = dev.to_device(x)
    plan.forward(d.data_ptr(), d.data_ptr(), 1, dev.stream)
    assert np.array_equal(dev.to_host(d), orc.forward(x, q, tw, pre, n))
    plan.inverse(d.data_ptr(), d.data_ptr(), 1, dev.stream)
    assert np.array_equal(dev.to_host(d), x)
    plan.close()


def test_config4_shape_one_gpu_slice(agx, orc, dev):
    """n=16384, 8-prime RNS; one GPU's slice is cut down to batch 256 here (full batch 65536 is
    64 GiB over 8 GPUs): round trip plus sampled frames against the oracle"""
    n, primes, batch = 16384, 8, 256
    qs = agx.find_primes(60, n, primes)
    plan = agx.Plan(n, qs)
    total = primes * batch * n
    d_a = dev.empty(total)
    plan.fill_synthetic(d_a.data_ptr(), batch, 0, 3, dev.stream)
    a = dev.to_host(d_a)
    d_f = dev.empty(total)
    plan.forward(d_a.data_ptr(), d_f.data_ptr(), batch, dev.stream)
    f = dev.to_host(d_f)
    for p in (0, 7):
        tw, pre = orc.make_tables(qs[p], plan.psi(p), n)
        for k in (0, 255):
            sl = slice((p * batch + k) * n, (p * batch + k + 1) * n)
            assert np.array_equal(orc.forward(a[sl], qs[p], tw, pre, n), f[sl])
    plan.inverse(d_f.data_ptr(), d_f.data_ptr(), batch, dev.stream)
    assert np.array_equal(dev.to_host(d_f), a)
    plan.close()


def test_config4_full_per_gpu_slice_on_device(agx, orc, dev):
    """BASELINE configs[3] at its real per-GPU size: n=16384, 8 primes, 65536/8 = 8192 polynomials
    = 8 GiB in place.  Checked on the device (round trip; forward out of place == in place) and on a
    few frames copied back for the oracle."""
    import torch

    n, primes, batch = 16384, 8, 8192
    qs = agx.find_primes(60, n, primes)
    plan = agx.Plan(n, qs)
    total = primes * batch * n
    a = dev.empty(total)
    plan.fill_synthetic(a.data_ptr(), batch, 0, 7, dev.stream)
    ref = a.clone()
    fwd_oop = dev.empty(total)
    plan.forward(a.data_ptr(), fwd_oop.data_ptr(), batch, dev.stream)      # fused-split kernels
    plan.forward(a.data_ptr(), a.data_ptr(), batch, dev.stream)            # pair kernel, in place
    dev.sync()
    assert torch.equal(a, fwd_oop)
    for p in (0, 5):
        tw, pre = orc.make_tables(qs[p], plan.psi(p), n)
        for f in (0, 4097, 8191):
            lo = (p * batch + f) * n
            x = ref[lo:lo + n].cpu().numpy().view(np.uint64)
            assert np.array_equal(a[lo:lo + n].cpu().numpy().view(np.uint64), orc.forward(x, qs[p], tw, pre, n))
    del fwd_oop
    plan.inverse(a.data_ptr(), a.data_ptr(), batch, dev.stream)
    dev.sync()
    assert torch.equal(a, ref)
    plan.close()


def test_config5_polymul_32768(agx, orc, dev):
    """n=32768 full poly-mul (NTT -> pointwise -> INTT): convolution theorem checked through
    X^j * b = negacyclic shift of b, and against the oracle pipeline on one frame"""
    n, batch = 32768, 4
    q = agx.find_primes(60, n)[0]
    plan = agx.Plan(n, [q])
    rng = np.random.default_rng(99)
    b = rand_coeffs(rng, batch * n, q)
    a = np.zeros(batch * n, dtype=np.uint64)
    shifts = [0, 1, 12345, n - 1]
    for f, j in enumerate(shifts):
        a[f * n + j] = 1
    d_a, d_b, d_c, d_s = dev.to_device(a), dev.to_device(b), dev.empty(a.size), dev.empty(a.size)
    plan.polymul(d_a.data_ptr(), d_b.data_ptr(), d_c.data_ptr(), d_s.data_ptr(), batch, dev.stream)
    c = dev.to_host(d_c)
    for f, j in enumerate(shifts):
        bf = b[f * n:(f + 1) * n]
        want = np.concatenate([(np.uint64(q) - bf[n - j:]) % np.uint64(q), bf[:n - j]]) if j else bf
        assert np.array_equal(c[f * n:(f + 1) * n], want)
    tw, pre = orc.make_tables(q, plan.psi(0), n)
    itw, _ = orc.make_inv_tables(q, plan.psi(0), n)
    a2 = rand_coeffs(rng, n, q)
    d_a2 = dev.to_device(np.concatenate([a2, a[n:]]))
    plan.polymul(d_a2.data_ptr(), d_b.data_ptr(), d_c.data_ptr(), d_s.data_ptr(), batch, dev.stream)
    want = orc.inverse(orc.pointwise(orc.forward(a2, q, tw, pre, n), orc.forward(b[:n], q, tw, pre, n), q), q, itw, n)
    assert np.array_equal(dev.to_host(d_c)[:n], want)
    plan.close()


def _oracle_polymul(orc, a, b, q, psi, n):
    """INTT(NTT(a) o NTT(b)) through the oracle's own transforms, frame by frame"""
    tw, pre = orc.make_tables(q, psi, n)
    itw, _ = orc.make_inv_tables(q, psi, n)
    fa = orc.forward(a % np.uint64(q), q, tw, pre, n)
    fb = orc.forward(b % np.uint64(q), q, tw, pre, n)
    return orc.inverse(orc.pointwise(fa, fb, q), q, itw, n)


@pytest.mark.parametrize("bits", [60, 61, 62])
@pytest.mark.parametrize("n", [1024, 2048, 4096, 8192, 16384, 32768])
def test_polymul_every_fused_kernel(agx, orc, dev, n, bits):
    """agx_ntt_polymul at every size with a one-launch kernel, in the 16q-lazy (60-bit), fast (61-bit) and exact (62-bit)
    arithmetic: polymul_rb2 (two frames in registers) for n <= 8192 and polymul_rb2_park (one frame in registers, the other
    parked in c's frame) of the R = 5 kernels at n = 16384 / 32768.  Expected: schoolbook product for n <= 2048, the oracle's
    NTT pipeline above that; c distinct, c aliasing a, c aliasing b."""
    batch, primes = 3, 2
    plan, tabs = _plan_from_oracle_tables(agx, orc, n, bits, primes)
    rng = np.random.default_rng(n * 7 + bits)
    a = np.concatenate([rand_coeffs(rng, batch * n, t[0]) for t in tabs])
    b = np.concatenate([rand_coeffs(rng, batch * n, t[0]) for t in tabs])
    want = np.empty_like(a)
    for p, t in enumerate(tabs):
        for f in range(batch):
            sl = slice((p * batch + f) * n, (p * batch + f + 1) * n)
            want[sl] = orc.schoolbook(a[sl], b[sl], t[0], n) if n <= 2048 else _oracle_polymul(orc, a[sl], b[sl], t[0], t[1], n)
    d_a, d_b, d_c, d_s = dev.to_device(a), dev.to_device(b), dev.empty(a.size), dev.empty(a.size)
    plan.polymul(d_a.data_ptr(), d_b.data_ptr(), d_c.data_ptr(), d_s.data_ptr(), batch, dev.stream)
    assert np.array_equal(dev.to_host(d_c), want)
    assert np.array_equal(dev.to_host(d_a), a) and np.array_equal(dev.to_host(d_b), b)      # operands untouched
    plan.polymul(d_a.data_ptr(), d_b.data_ptr(), d_a.data_ptr(), d_s.data_ptr(), batch, dev.stream)   # c aliasing a
    assert np.array_equal(dev.to_host(d_a), want)
    d_a = dev.to_device(a)
    plan.polymul(d_a.data_ptr(), d_b.data_ptr(), d_b.data_ptr(), d_s.data_ptr(), batch, dev.stream)   # c aliasing b
    assert np.array_equal(dev.to_host(d_b), want)
    plan.close()


@pytest.mark.parametrize("n,config", [(16384, None), (32768, None), (16384, 160), (16384, 120), (32768, 123)])
def test_one_launch_product_with_aliasing_and_no_scratch(agx, orc, dev, n, config):
    """the parked-operand fused product (polymul_rb2_park) at n = 16384 / 32768 -- the R = 5 defaults park NTT(first) thread-privately
    in c's own frame, registry id 37 (R = 4, A/B) parks it in natural order and redistributes through LDS; 120 / 123: the fast /
    exact forms under a 60-bit modulus: no caller scratch, c distinct / aliasing a / aliasing b, operands in [0,4q), two primes"""
    batch, primes = 5, 2
    plan, tabs = _plan_from_oracle_tables(agx, orc, n, 60, primes)
    _select(agx, plan, config)
    rng = np.random.default_rng(3700)
    a = np.concatenate([rand_coeffs(rng, batch * n, t[0], hi_mult=4) for t in tabs])
    b = np.concatenate([rand_coeffs(rng, batch * n, t[0], hi_mult=4) for t in tabs])
    want = np.empty_like(a)
    for p, t in enumerate(tabs):
        for f in range(batch):
            sl = slice((p * batch + f) * n, (p * batch + f + 1) * n)
            want[sl] = _oracle_polymul(orc, a[sl], b[sl], t[0], t[1], n)
    d_a, d_b, d_c = dev.to_device(a), dev.to_device(b), dev.empty(a.size)
    plan.polymul(d_a.data_ptr(), d_b.data_ptr(), d_c.data_ptr(), 0, batch, dev.stream)          # scratch = NULL
    assert np.array_equal(dev.to_host(d_c), want)
    assert np.array_equal(dev.to_host(d_a), a) and np.array_equal(dev.to_host(d_b), b)
    plan.polymul(d_a.data_ptr(), d_b.data_ptr(), d_a.data_ptr(), 0, batch, dev.stream)          # c aliasing a
    assert np.array_equal(dev.to_host(d_a), want)
    d_a = dev.to_device(a)
    plan.polymul(d_a.data_ptr(), d_b.data_ptr(), d_b.data_ptr(), 0, batch, dev.stream)          # c aliasing b
    assert np.array_equal(dev.to_host(d_b), want)
    plan.close()


def test_polymul_lazy_operands(agx, orc, dev):
    """operands anywhere in [0,4q) (the transforms' input contract) through the fused product at n=4096, 60-bit q:
    the 16q-lazy forward results feed the Barrett product unreduced"""
    n, batch = 4096, 2
    plan, tabs = _plan_from_oracle_tables(agx, orc, n, 60, 1)
    q, psi = tabs[0][0], tabs[0][1]
    rng = np.random.default_rng(4242)
    a = rand_coeffs(rng, batch * n, q, hi_mult=4)
    b = np.concatenate([np.full(n, 4 * q - 1, dtype=np.uint64), rand_coeffs(rng, n, q, hi_mult=4)])
    want = np.concatenate([_oracle_polymul(orc, a[f * n:(f + 1) * n], b[f * n:(f + 1) * n], q, psi, n) for f in range(batch)])
    d_a, d_b, d_c = dev.to_device(a), dev.to_device(b), dev.empty(a.size)
    plan.polymul(d_a.data_ptr(), d_b.data_ptr(), d_c.data_ptr(), 0, batch, dev.stream)
    assert np.array_equal(dev.to_host(d_c), want)
    plan.close()


# every entry of the kernel registry with the size it serves and the largest modulus its arithmetic admits
# (exact: 62 bits, fast: 61, 16q-lazy: 60); defaults are reached by the other tests, this one reaches the rest
REGISTRY = [
    (91, 4096, 62), (92, 4096, 61), (93, 4096, 60), (159, 4096, 60), (147, 4096, 60), (161, 4096, 60), (70, 4096, 60),
    # streamed single-frame kernels (lazy, fast, exact): n = 1024 / 2048 / 8192, 16384 (117 + forward companion 164), 32768; A/B twins 115 / 160 / 114
    (150, 1024, 60), (151, 1024, 61), (152, 1024, 62), (153, 2048, 60), (154, 2048, 61), (155, 2048, 62), (156, 8192, 60), (157, 8192, 61), (158, 8192, 62),
    (117, 16384, 60), (164, 16384, 60), (120, 16384, 61), (122, 16384, 62), (115, 16384, 60), (160, 16384, 60),
    (119, 32768, 60), (121, 32768, 61), (123, 32768, 62), (114, 32768, 60),
    # 32-bit arithmetic: tier 2 (every q < 2^30), tier 1 (every q < 2^31)
    (130, 1024, 30), (131, 2048, 30), (132, 4096, 30), (133, 8192, 30), (134, 16384, 30), (135, 32768, 30),
    (136, 1024, 31), (137, 2048, 31), (138, 4096, 31), (139, 8192, 31), (140, 16384, 31), (141, 32768, 31),
    # wave-packed kernels of n = 32 ... 512 (csrc/wp_kernels.hpp): 16q-lazy / fast / exact per size, then the 32-bit tiers
    (200, 32, 60), (201, 32, 61), (202, 32, 62), (203, 64, 60), (204, 64, 61), (205, 64, 62), (206, 128, 60), (207, 128, 61), (208, 128, 62),
    (209, 256, 60), (210, 256, 61), (211, 256, 62), (212, 512, 60), (213, 512, 61), (214, 512, 62),
    (230, 32, 30), (231, 64, 30), (232, 128, 30), (233, 256, 30), (234, 512, 30), (240, 32, 31), (241, 64, 31), (242, 128, 31), (243, 256, 31), (244, 512, 31),
    # n = 2 ... 16: one lane per frame (fast / exact; 32-bit tiers)
    (250, 2, 61), (251, 2, 62), (252, 4, 61), (253, 4, 62), (254, 8, 61), (255, 8, 62), (256, 16, 61), (257, 16, 62),
    (260, 2, 30), (261, 4, 30), (262, 8, 30), (263, 16, 30), (264, 2, 31), (265, 4, 31), (266, 8, 31), (267, 16, 31),
    # A/B shapes of the wave-packed kernels (lib/libagxntt_diag.so)
    (215, 32, 60), (220, 512, 60), (221, 512, 60), (222, 256, 60), (224, 32, 60), (235, 32, 30), (236, 512, 30),
]


@pytest.mark.parametrize("config,n,max_bits", REGISTRY)
def test_every_registry_entry_at_its_own_size(agx, orc, dev, config, n, max_bits):
    """each registered kernel configuration selected explicitly (AGX_VARIANT_REGBLOCK_BASE + id) at the size it
    serves, under the largest modulus its arithmetic form admits and under a 30-bit one: forward in place and
    out of place against the oracle, then (where the entry has them) inverse and fused product"""
    batch = 3
    for bits in (max_bits, 30):
        plan, tabs = _plan_from_oracle_tables(agx, orc, n, bits, 1)
        _select(agx, plan, config)
        q, psi, tw, pre = tabs[0]
        rng = np.random.default_rng(config * 131 + bits)
        x = rand_coeffs(rng, batch * n, q, hi_mult=4 if bits < 62 else 3)
        want = orc.forward(x, q, tw, pre, n)
        d_x, d_y = dev.to_device(x), dev.empty(x.size)
        plan.forward(d_x.data_ptr(), d_y.data_ptr(), batch, dev.stream)
        assert np.array_equal(dev.to_host(d_y), want), (config, bits, "out of place")
        assert np.array_equal(dev.to_host(d_x), x)
        plan.forward(d_x.data_ptr(), d_x.data_ptr(), batch, dev.stream)
        assert np.array_equal(dev.to_host(d_x), want), (config, bits, "in place")
        plan.inverse(d_x.data_ptr(), d_x.data_ptr(), batch, dev.stream)
        assert np.array_equal(dev.to_host(d_x), x % np.uint64(q)), (config, bits, "inverse")
        a, b = rand_coeffs(rng, batch * n, q), rand_coeffs(rng, batch * n, q)
        d_a, d_b, d_c, d_s = dev.to_device(a), dev.to_device(b), dev.empty(a.size), dev.empty(a.size)
        plan.polymul(d_a.data_ptr(), d_b.data_ptr(), d_c.data_ptr(), d_s.data_ptr(), batch, dev.stream)
        wantc = np.concatenate([_oracle_polymul(orc, a[f * n:(f + 1) * n], b[f * n:(f + 1) * n], q, psi, n) for f in range(batch)])
        assert np.array_equal(dev.to_host(d_c), wantc), (config, bits, "polymul")
        plan.close()


SMALL_SIZES = [2, 4, 8, 16, 32, 64, 128, 256, 512]


@pytest.mark.parametrize("bits", [60, 61, 62, 31, 30, 20])
@pytest.mark.parametrize("n", SMALL_SIZES)
def test_wave_packed_small_sizes(agx, orc, dev, n, bits):
    """n = 2 ... 512 (n = 32 is in the reference's size table, include/kernel/ntt.h:11-12; below it one lane holds a frame): several frames share a wave
    (csrc/wp_kernels.hpp), so the frame counts here straddle the frames-per-wave and frames-per-workgroup boundaries (a wave whose
    last frames do not exist folds them back for loads and masks their stores).  Forward out of place and in place on inputs in
    [0,4q), lazy outputs, inverse on arbitrary data, extreme coefficients, the one-launch product with every aliasing and NO scratch
    -- all against the oracle, for every arithmetic form (16q-lazy / fast / exact, 32-bit tier 1 / tier 2)."""
    primes = 2
    plan, tabs = _plan_from_oracle_tables(agx, orc, n, bits, primes)
    rng = np.random.default_rng(n * 7 + bits)
    for batch in (1, 3, 17, 64, 67, 259):
        hi = 4 if bits < 62 else 3
        x = np.concatenate([rand_coeffs(rng, batch * n, t[0], hi_mult=hi) for t in tabs])
        # extreme coefficients in the first frames: 0, q-1, 4q-1 (the top of the lazy input range)
        for p, t in enumerate(tabs):
            x[p * batch * n:p * batch * n + n:3] = np.uint64(t[0] - 1)
            if batch > 1:
                x[(p * batch + 1) * n:(p * batch + 2) * n:2] = np.uint64(hi * t[0] - 1)
        want = _oracle_forward_rns(orc, x, tabs, n, batch)
        d_x, d_y = dev.to_device(x), dev.empty(x.size)
        plan.forward(d_x.data_ptr(), d_y.data_ptr(), batch, dev.stream)
        assert np.array_equal(dev.to_host(d_y), want), (n, bits, batch, "forward out of place")
        assert np.array_equal(dev.to_host(d_x), x), "input must not be modified"
        plan.forward_lazy(d_x.data_ptr(), d_y.data_ptr(), batch, dev.stream)
        y = dev.to_host(d_y)
        for p, t in enumerate(tabs):
            sl = slice(p * batch * n, (p + 1) * batch * n)
            assert (y[sl].astype(object) < 4 * t[0]).all()
            assert np.array_equal(y[sl] % np.uint64(t[0]), want[sl]), (n, bits, batch, "lazy")
        plan.inverse(d_y.data_ptr(), d_y.data_ptr(), batch, dev.stream)      # lazy values are legal inverse inputs
        xr = np.concatenate([x[p * batch * n:(p + 1) * batch * n] % np.uint64(t[0]) for p, t in enumerate(tabs)])
        assert np.array_equal(dev.to_host(d_y), xr), (n, bits, batch, "round trip")
        plan.forward(d_x.data_ptr(), d_x.data_ptr(), batch, dev.stream)
        assert np.array_equal(dev.to_host(d_x), want), (n, bits, batch, "forward in place")
        # inverse of arbitrary (not forward-image) data against the oracle's definitional inverse
        r = np.concatenate([rand_coeffs(rng, batch * n, t[0]) for t in tabs])
        d_r, d_z = dev.to_device(r), dev.empty(r.size)
        plan.inverse(d_r.data_ptr(), d_z.data_ptr(), batch, dev.stream)
        wanti = np.concatenate([orc.inverse(r[p * batch * n:(p + 1) * batch * n], t[0], orc.make_inv_tables(t[0], t[1], n)[0], n) for p, t in enumerate(tabs)])
        assert np.array_equal(dev.to_host(d_z), wanti), (n, bits, batch, "inverse")
    # one-launch product, scratch = NULL: c distinct, c = a, c = b, squaring with and without aliasing
    batch = 21
    a = np.concatenate([rand_coeffs(rng, batch * n, t[0]) for t in tabs])
    b = np.concatenate([rand_coeffs(rng, batch * n, t[0]) for t in tabs])

    def product(u, v):
        return np.concatenate([_oracle_polymul(orc, u[(p * batch + f) * n:(p * batch + f + 1) * n], v[(p * batch + f) * n:(p * batch + f + 1) * n], t[0], t[1], n)
                               for p, t in enumerate(tabs) for f in range(batch)])

    want_ab, want_aa = product(a, b), product(a, a)
    d_a, d_b, d_c = dev.to_device(a), dev.to_device(b), dev.empty(a.size)
    plan.polymul(d_a.data_ptr(), d_b.data_ptr(), d_c.data_ptr(), 0, batch, dev.stream)
    assert np.array_equal(dev.to_host(d_c), want_ab), (n, bits, "product")
    plan.polymul(d_a.data_ptr(), d_a.data_ptr(), d_c.data_ptr(), 0, batch, dev.stream)
    assert np.array_equal(dev.to_host(d_c), want_aa), (n, bits, "square")
    d_t = dev.to_device(a)
    plan.polymul(d_t.data_ptr(), d_b.data_ptr(), d_t.data_ptr(), 0, batch, dev.stream)
    assert np.array_equal(dev.to_host(d_t), want_ab), (n, bits, "c = a")
    d_t = dev.to_device(b)
    plan.polymul(d_a.data_ptr(), d_t.data_ptr(), d_t.data_ptr(), 0, batch, dev.stream)
    assert np.array_equal(dev.to_host(d_t), want_ab), (n, bits, "c = b")
    d_t = dev.to_device(a)
    plan.polymul(d_t.data_ptr(), d_t.data_ptr(), d_t.data_ptr(), 0, batch, dev.stream)
    assert np.array_equal(dev.to_host(d_t), want_aa), (n, bits, "square in place")
    if n <= 128:      # against the definition too
        sl = slice(0, n)
        assert np.array_equal(want_ab[sl], orc.schoolbook(a[sl], b[sl], tabs[0][0], n))
    plan.close()


@pytest.mark.parametrize("n", SMALL_SIZES)
def test_wave_packed_strided_layouts(agx, orc, dev, n):
    """[poly][prime][n] callers at the small sizes (prime_stride = n, poly_stride = P n: the frames of a wave are NOT adjacent), and a
    padded dense layout (poly_stride = n + 8), forward and inverse, 60- and 30-bit moduli"""
    for bits in (60, 30):
        batch, primes = 37, 3
        plan, tabs = _plan_from_oracle_tables(agx, orc, n, bits, primes)
        rng = np.random.default_rng(n + bits)
        x = np.empty((batch, primes, n), dtype=np.uint64)
        for p in range(primes):
            x[:, p, :] = rand_coeffs(rng, batch * n, tabs[p][0]).reshape(batch, n)
        d = dev.to_device(x.reshape(-1))
        plan.forward_strided(d.data_ptr(), d.data_ptr(), batch, n, primes * n, dev.stream)
        y = dev.to_host(d).reshape(batch, primes, n)
        for p, (q, _, tw, pre) in enumerate(tabs):
            assert np.array_equal(y[:, p, :].reshape(-1), orc.forward(np.ascontiguousarray(x[:, p, :]).reshape(-1), q, tw, pre, n)), (n, bits, p)
        plan.inverse_strided(d.data_ptr(), d.data_ptr(), batch, n, primes * n, dev.stream)
        assert np.array_equal(dev.to_host(d).reshape(batch, primes, n), x)
        # padded frames: poly_stride = n + 8, prime_stride = batch (n + 8); the pad words must come back untouched
        pad = n + 8
        z = np.full((primes, batch, pad), 0xDEADBEEFCAFEF00D, dtype=np.uint64)
        for p in range(primes):
            z[p, :, :n] = x[:, p, :]
        d = dev.to_device(z.reshape(-1))
        plan.forward_strided(d.data_ptr(), d.data_ptr(), batch, batch * pad, pad, dev.stream)
        got = dev.to_host(d).reshape(primes, batch, pad)
        assert (got[:, :, n:] == np.uint64(0xDEADBEEFCAFEF00D)).all()
        for p in range(primes):
            assert np.array_equal(got[p, :, :n], y[:, p, :])
        plan.close()


@pytest.mark.parametrize("bits", [60, 61, 62, 30])
@pytest.mark.parametrize("n", [1024, 2048, 4096, 8192, 16384, 32768])
def test_polymul_squaring_with_and_without_aliasing(agx, orc, dev, n, bits):
    """agx_ntt_polymul(a, a, c) and (a, a, a) (the parked product read its own parked transform back as the second
    operand when a == b == c; squaring now has its own kernel there); against the oracle's product"""
    batch = 3
    plan, tabs = _plan_from_oracle_tables(agx, orc, n, bits, 1)
    q, psi = tabs[0][0], tabs[0][1]
    rng = np.random.default_rng(n + bits + 99)
    a = rand_coeffs(rng, batch * n, q)
    want = np.concatenate([_oracle_polymul(orc, a[f * n:(f + 1) * n], a[f * n:(f + 1) * n], q, psi, n) for f in range(batch)])
    d_a, d_c = dev.to_device(a), dev.empty(a.size)
    plan.polymul(d_a.data_ptr(), d_a.data_ptr(), d_c.data_ptr(), 0, batch, dev.stream)
    assert np.array_equal(dev.to_host(d_c), want), (n, bits, "a == b, c distinct")
    assert np.array_equal(dev.to_host(d_a), a)
    plan.polymul(d_a.data_ptr(), d_a.data_ptr(), d_a.data_ptr(), 0, batch, dev.stream)
    assert np.array_equal(dev.to_host(d_a), want), (n, bits, "a == b == c")
    plan.close()


@pytest.mark.parametrize("config,n,batch", [(117, 16384, 333), (119, 32768, 290)])
def test_loop_kernels_more_frames_than_workgroups(agx, orc, dev, config, n, batch):
    """the loop kernels (a resident grid walking over the frames -- 117 / 119: the n = 16384 / 32768 defaults, inverse by the
    ticket-drawing loop kernel) on more frames than the chip holds workgroups, a frame count that is not a multiple of the grid,
    two primes, in place: forward and inverse against the oracle"""
    primes = 2
    plan, tabs = _plan_from_oracle_tables(agx, orc, n, 60, primes)
    _select(agx, plan, config)
    rng = np.random.default_rng(config + 1000)
    x = np.concatenate([rand_coeffs(rng, batch * n, t[0], hi_mult=4) for t in tabs])
    d = dev.to_device(x)
    plan.forward(d.data_ptr(), d.data_ptr(), batch, dev.stream)
    assert np.array_equal(dev.to_host(d), _oracle_forward_rns(orc, x, tabs, n, batch))
    r = np.concatenate([rand_coeffs(rng, batch * n, t[0], hi_mult=4) for t in tabs])
    want = np.concatenate([orc.inverse(r[p * batch * n:(p + 1) * batch * n] % np.uint64(tabs[p][0]), tabs[p][0],
                                       orc.make_inv_tables(tabs[p][0], tabs[p][1], n)[0], n) for p in range(primes)])
    d_r = dev.to_device(r)
    plan.inverse(d_r.data_ptr(), d_r.data_ptr(), batch, dev.stream)
    assert np.array_equal(dev.to_host(d_r), want)
    plan.close()


@pytest.mark.parametrize("config", [117, 119])
def test_dynamic_loop_kernels_on_two_streams_of_one_plan(agx, orc, dev, config):
    """the ticket-drawing loop kernels (the inverse of the n = 16384 / 32768 defaults) launched back to back on two streams of ONE
    plan, so that launches overlap: the plan keeps one ticket pair per stream, so both results must be right, repeatedly"""
    import torch

    n, batch = (16384, 700) if config == 117 else (32768, 400)
    plan, tabs = _plan_from_oracle_tables(agx, orc, n, 60, 1)
    _select(agx, plan, config)
    q, psi, tw, pre = tabs[0]
    rng = np.random.default_rng(3737)
    xa, xb = rand_coeffs(rng, batch * n, q), rand_coeffs(rng, batch * n, q)
    wa, wb = orc.forward(xa, q, tw, pre, n), orc.forward(xb, q, tw, pre, n)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for rep in range(3):
        da, db = dev.to_device(xa), dev.to_device(xb)
        dev.sync()
        plan.forward(da.data_ptr(), da.data_ptr(), batch, s1.cuda_stream)
        plan.forward(db.data_ptr(), db.data_ptr(), batch, s2.cuda_stream)
        plan.inverse(da.data_ptr(), da.data_ptr(), batch, s1.cuda_stream)
        plan.inverse(db.data_ptr(), db.data_ptr(), batch, s2.cuda_stream)
        plan.forward(da.data_ptr(), da.data_ptr(), batch, s1.cuda_stream)
        plan.forward(db.data_ptr(), db.data_ptr(), batch, s2.cuda_stream)
        dev.sync()
        assert np.array_equal(dev.to_host(da), wa) and np.array_equal(dev.to_host(db), wb), rep
    plan.close()


@pytest.mark.parametrize("n,config,batch", [(32768, None, 1100), (16384, None, 2100), (32768, 114, 1100), (16384, 115, 2100)])
def test_large_frames_on_many_rounds_of_workgroups(agx, orc, dev, n, config, batch):
    """n = 32768 / 16384 on several rounds of workgroups per CU: the whole-frame R = 5 defaults and the two-halves-in-turn kernels
    they replaced (registry ids 54 / 53, A/B): forward in place and inverse against the oracle"""
    plan, tabs = _plan_from_oracle_tables(agx, orc, n, 60, 1)
    _select(agx, plan, config)
    q, psi, tw, pre = tabs[0]
    rng = np.random.default_rng(n + batch)
    x = rand_coeffs(rng, batch * n, q, hi_mult=4)
    d = dev.to_device(x)
    plan.forward(d.data_ptr(), d.data_ptr(), batch, dev.stream)
    assert np.array_equal(dev.to_host(d), orc.forward(x, q, tw, pre, n))
    plan.inverse(d.data_ptr(), d.data_ptr(), batch, dev.stream)
    assert np.array_equal(dev.to_host(d), x % np.uint64(q))
    r = rand_coeffs(rng, batch * n, q, hi_mult=4)
    d_r = dev.to_device(r)
    plan.inverse(d_r.data_ptr(), d_r.data_ptr(), batch, dev.stream)
    assert np.array_equal(dev.to_host(d_r), orc.inverse(r % np.uint64(q), q, orc.make_inv_tables(q, psi, n)[0], n))
    plan.close()


# ---------------------------------------------------------------------------------------
# narrow moduli: the 32-bit arithmetic kernels (csrc/rb32_kernels.hpp)
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("bits", [17, 20, 30, 31])
@pytest.mark.parametrize("n", [1024, 2048, 4096, 8192, 16384, 32768])
def test_narrow_modulus_kernels(agx, orc, dev, n, bits):
    """plans whose every modulus is below 2^31 take the 32-bit Shoup / Harvey kernels (tier 2 below 2^30, tier 1 below 2^31): the
    reference's own modulus class (src/main.cpp:55 is 65537; BASELINE configs[0] is a 30-bit prime).  Against the oracle, which
    computes in 64 bits: forward out of place / in place / lazy on inputs anywhere in [0,4q), inverse of lazy inputs and round
    trip, fused product with c aliasing either operand, worst-case inputs (all 4q-1, all q-1, alternating); ragged batch."""
    batch = 5 if n <= 4096 else 3
    plan, tabs = _plan_from_oracle_tables(agx, orc, n, bits, 1)
    q, psi, tw, pre = tabs[0]
    assert q < (1 << bits) and q >= 3
    itw = orc.make_inv_tables(q, psi, n)[0]
    rng = np.random.default_rng(n * 31 + bits)
    x = rand_coeffs(rng, batch * n, q, hi_mult=4)
    want = orc.forward(x, q, tw, pre, n)
    d_x, d_y = dev.to_device(x), dev.empty(x.size)
    plan.forward(d_x.data_ptr(), d_y.data_ptr(), batch, dev.stream)
    assert np.array_equal(dev.to_host(d_y), want), "out of place"
    assert np.array_equal(dev.to_host(d_x), x), "input must not be modified"
    plan.forward_lazy(d_x.data_ptr(), d_y.data_ptr(), batch, dev.stream)
    lazy = dev.to_host(d_y)
    assert (lazy < np.uint64(4 * q)).all() and np.array_equal(lazy % np.uint64(q), want), "lazy outputs"
    plan.inverse(d_y.data_ptr(), d_y.data_ptr(), batch, dev.stream)          # lazy values are legal inverse inputs
    assert np.array_equal(dev.to_host(d_y), x % np.uint64(q)), "round trip"
    plan.forward(d_x.data_ptr(), d_x.data_ptr(), batch, dev.stream)
    assert np.array_equal(dev.to_host(d_x), want), "in place"
    r = rand_coeffs(rng, batch * n, q, hi_mult=4)
    d_r = dev.to_device(r)
    plan.inverse(d_r.data_ptr(), d_r.data_ptr(), batch, dev.stream)
    assert np.array_equal(dev.to_host(d_r), orc.inverse(r % np.uint64(q), q, itw, n)), "inverse"
    # worst cases of the lazy ranges
    top = 4 * q - 1
    e = np.concatenate([np.full(n, top, dtype=np.uint64), np.full(n, q - 1, dtype=np.uint64),
                        np.where(np.arange(n) % 2 == 0, np.uint64(top), np.uint64(0))])
    d_e = dev.to_device(e)
    plan.forward(d_e.data_ptr(), d_e.data_ptr(), 3, dev.stream)
    assert np.array_equal(dev.to_host(d_e), orc.forward(e, q, tw, pre, n)), "extreme forward"
    d_e = dev.to_device(e)
    plan.inverse(d_e.data_ptr(), d_e.data_ptr(), 3, dev.stream)
    assert np.array_equal(dev.to_host(d_e), orc.inverse(e % np.uint64(q), q, itw, n)), "extreme inverse"
    # fused product, operands in [0,4q)
    a, b = rand_coeffs(rng, batch * n, q, hi_mult=4), rand_coeffs(rng, batch * n, q, hi_mult=4)
    a[:n] = top
    b[:n] = top
    wantc = np.concatenate([orc.schoolbook(a[f * n:(f + 1) * n] % np.uint64(q), b[f * n:(f + 1) * n] % np.uint64(q), q, n) if n <= 2048
                            else _oracle_polymul(orc, a[f * n:(f + 1) * n], b[f * n:(f + 1) * n], q, psi, n) for f in range(batch)])
    d_a, d_b, d_c = dev.to_device(a), dev.to_device(b), dev.empty(a.size)
    plan.polymul(d_a.data_ptr(), d_b.data_ptr(), d_c.data_ptr(), 0, batch, dev.stream)
    assert np.array_equal(dev.to_host(d_c), wantc), "product"
    plan.polymul(d_a.data_ptr(), d_b.data_ptr(), d_a.data_ptr(), 0, batch, dev.stream)
    assert np.array_equal(dev.to_host(d_a), wantc), "product, c aliasing a"
    d_a = dev.to_device(a)
    plan.polymul(d_a.data_ptr(), d_b.data_ptr(), d_b.data_ptr(), 0, batch, dev.stream)
    assert np.array_equal(dev.to_host(d_b), wantc), "product, c aliasing b"
    plan.close()


@pytest.mark.parametrize("bits_list", [(30, 31), (17, 30, 20), (30, 60), (31, 31)])
def test_narrow_and_mixed_modulus_plans(agx, orc, dev, bits_list):
    """RNS plans mixing modulus sizes: (30, 31) -> the tier-1 32-bit kernels for every prime, (17, 30, 20) -> tier 2, (30, 60) -> the
    64-bit kernels (one modulus is wide); strided [poly][prime][n] layout with an odd offset (8-byte aligned frames only: the
    narrow kernels must then take their 8-byte access path), forward and inverse against the oracle per prime"""
    n, batch = 4096, 4
    used, tabs = set(), []
    for b in bits_list:
        k = 0
        while orc.find_prime(b, n, k) in used:
            k += 1
        q = orc.find_prime(b, n, k)
        used.add(q)
        psi = orc.min_root(q, n)
        tabs.append((q, psi) + tuple(orc.make_tables(q, psi, n)))
    primes = len(tabs)
    inv = [orc.make_inv_tables(t[0], t[1], n) for t in tabs]
    plan = agx.Plan(n, [t[0] for t in tabs], tables=(np.stack([t[2] for t in tabs]), np.stack([t[3] for t in tabs]),
                                                      np.stack([i[0] for i in inv]), np.stack([i[1] for i in inv])))
    rng = np.random.default_rng(sum(bits_list))
    poly_stride, prime_stride, off = primes * n + 6, n + 2, 1          # odd offset: frames are only 8-byte aligned
    buf = np.zeros(off + batch * poly_stride, dtype=np.uint64)
    want = np.zeros_like(buf)
    for p, t in enumerate(tabs):
        for b in range(batch):
            lo = off + b * poly_stride + p * prime_stride
            buf[lo:lo + n] = rand_coeffs(rng, n, t[0], hi_mult=4)
            want[lo:lo + n] = orc.forward(buf[lo:lo + n], t[0], t[2], t[3], n)
    d = dev.to_device(buf)
    ptr = d.data_ptr() + 8 * off
    plan.forward_strided(ptr, ptr, batch, prime_stride, poly_stride, dev.stream)
    got = dev.to_host(d)
    assert np.array_equal(got, want)          # gaps between frames untouched (zero)
    plan.inverse_strided(ptr, ptr, batch, prime_stride, poly_stride, dev.stream)
    back = dev.to_host(d)
    for p, t in enumerate(tabs):
        for b in range(batch):
            lo = off + b * poly_stride + p * prime_stride
            assert np.array_equal(back[lo:lo + n], buf[lo:lo + n] % np.uint64(t[0])), (p, b)
    plan.close()


def test_out_of_contract_tables_stay_on_the_exact_kernels(agx, orc, dev):
    """a narrow modulus with tables that do NOT honour precon = floor(w 2^64 / q) (here: precons off by one) must not take the 32-bit
    (or any lazy) kernels: the exact kernels repeat the reference's operations mod 2^64 on whatever they are fed, as the oracle does"""
    n, batch = 4096, 2
    q = orc.find_prime(30, n)
    psi = orc.min_root(q, n)
    tw, pre = orc.make_tables(q, psi, n)
    bad = pre.copy()
    bad[5] -= np.uint64(1)
    plan = agx.Plan(n, [q], tables=(tw[None, :], bad[None, :]))
    x = rand_coeffs(np.random.default_rng(5), batch * n, q)
    d = dev.to_device(x)
    plan.forward(d.data_ptr(), d.data_ptr(), batch, dev.stream)
    assert np.array_equal(dev.to_host(d), orc.forward(x, q, tw, bad, n))
    plan.close()


def test_more_streams_than_ticket_slots(agx, orc, dev):
    """a plan keeps one {ticket, retired} pair per stream for 64 streams; launches on further streams take over the pair of a stream
    whose launches have completed, or the stateless fixed-stride kernels -- every result right, on 70 streams of one plan, twice (a wrapped ticket ring let two
    launches in flight share one counter and skip frames silently)"""
    import torch

    n, batch = 16384, 300
    plan, tabs = _plan_from_oracle_tables(agx, orc, n, 60, 1)
    q, psi, tw, pre = tabs[0]
    itw = orc.make_inv_tables(q, psi, n)[0]
    rng = np.random.default_rng(64)
    r = rand_coeffs(rng, batch * n, q, hi_mult=4)
    want = orc.inverse(r % np.uint64(q), q, itw, n)
    streams = [torch.cuda.Stream() for _ in range(70)]
    for rep in range(2):
        bufs = [dev.to_device(r) for _ in range(4)]
        dev.sync()
        for i, st in enumerate(streams):
            b = bufs[i % 4]
            if i >= 4:
                st.wait_stream(streams[i - 4])          # the buffer's previous user: inverse of an inverse is not what we check
                continue
            plan.inverse(b.data_ptr(), b.data_ptr(), batch, st.cuda_stream)
        dev.sync()
        for b in bufs:
            assert np.array_equal(dev.to_host(b), want), rep
        # and one launch per stream, each on a buffer of its own turn (sequential: 70 distinct streams touch the plan's slot table)
        for i, st in enumerate(streams):
            b = dev.to_device(r)
            plan.inverse(b.data_ptr(), b.data_ptr(), batch, st.cuda_stream)
            st.synchronize()
            assert np.array_equal(dev.to_host(b), want), (rep, i)
    # 70 launches IN FLIGHT together, one per stream, each on its own buffer: a slot may only change hands when its previous stream's
    # launch has completed (round 4: slots are recycled through an event recorded behind every ticket launch), so no two running
    # launches may ever share a pair
    for rep in range(2):
        bufs = [dev.to_device(r) for _ in streams]
        dev.sync()
        for b, st in zip(bufs, streams):
            plan.inverse(b.data_ptr(), b.data_ptr(), batch, st.cuda_stream)
        dev.sync()
        for i, b in enumerate(bufs):
            assert np.array_equal(dev.to_host(b), want), ("concurrent", rep, i)
    plan.close()


def test_one_shot_calls_from_two_threads_with_different_tables(agx, orc):
    """agx_ntt_forward_host from two host threads at once, each alternating between two table sets (so the per-device plan
    cache is evicted again and again): every result against the oracle (one global slot leaked / serialised)"""
    import threading

    n, frames = 2048, 6
    cases = []
    for k, bits in enumerate((30, 60, 45, 31)):
        q = orc.find_prime(bits, n)
        psi = orc.min_root(q, n)
        tw, pre = orc.make_tables(q, psi, n)
        x = rand_coeffs(np.random.default_rng(k), frames * n, q)
        cases.append((q, tw, pre, x, orc.forward(x, q, tw, pre, n)))
    errors = []

    def worker(mine):
        try:
            for rep in range(6):
                q, tw, pre, x, want = cases[mine[rep % 2]]
                got = agx.forward_host(x, x, q, tw, pre, n, frames)
                if not np.array_equal(got, want):
                    errors.append((mine, rep))
        except Exception as exc:      # noqa: BLE001
            errors.append(repr(exc))

    threads = [threading.Thread(target=worker, args=((0, 1),)), threading.Thread(target=worker, args=((2, 3),))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert agx.lib().agx_ntt_release_caches() == 0          # and the caches can be dropped and rebuilt
    q, tw, pre, x, want = cases[0]
    assert np.array_equal(agx.forward_host(x, x, q, tw, pre, n, frames), want)


def test_harness_binary_passes(agx):
    """bin/ntt_harness (src/main.cpp: the reference-shaped ntt_input_kernel / fwd_ntt_kernel<0> / ntt_output_kernel mirror,
    agx::ntt() / agx::intt(), structured known answers, inData2 != inData, schoolbook product at n = 64 / 1024, the reference's
    17-bit modulus) as a child process on the GPU"""
    import subprocess

    exe = os.path.join(os.path.dirname(agx.LIB_PATH), "..", "bin", "ntt_harness")
    if not os.path.exists(exe):
        agx.build()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "HARNESS PASSED" in r.stdout, r.stdout[-2000:]
    assert r.stdout.count("in2 != in") == 2 and r.stdout.count("schoolbook") == 3 and "FAIL" not in r.stdout, r.stdout[-2000:]


def test_randomised_shapes_against_oracle(agx, orc, dev):
    """40 random (n, modulus size, primes, batch, in/out of place, lazy) forward + inverse cases:
    every size class, every arithmetic form, ragged batches"""
    rng = np.random.default_rng(20261004)
    for case in range(40):
        n = 1 << int(rng.integers(1, 16))
        bits = int(rng.choice([max(20, n.bit_length() + 2), 30, 45, 59, 60, 61, 62]))
        if bits <= n.bit_length() + 1:
            bits = n.bit_length() + 4
        primes = int(rng.integers(1, 4)) if n <= 8192 else 1
        batch = int(rng.integers(1, 8)) if n <= 8192 else int(rng.integers(1, 3))
        plan, tabs = _plan_from_oracle_tables(agx, orc, n, bits, primes)
        x = np.concatenate([rand_coeffs(rng, batch * n, t[0], hi_mult=int(rng.integers(1, 5))) for t in tabs])
        d_x = dev.to_device(x)
        in_place = bool(rng.integers(0, 2))
        d_y = d_x if in_place else dev.empty(x.size)
        lazy = bool(rng.integers(0, 2))
        (plan.forward_lazy if lazy else plan.forward)(d_x.data_ptr(), d_y.data_ptr(), batch, dev.stream)
        y = dev.to_host(d_y)
        want = _oracle_forward_rns(orc, x, tabs, n, batch)
        for p, t in enumerate(tabs):
            sl = slice(p * batch * n, (p + 1) * batch * n)
            if lazy:
                assert (y[sl].astype(object) < 4 * t[0]).all(), (case, n, bits)
                assert np.array_equal(y[sl] % np.uint64(t[0]), want[sl]), (case, n, bits, primes, batch)
            else:
                assert np.array_equal(y[sl], want[sl]), (case, n, bits, primes, batch, in_place)
        plan.inverse(d_y.data_ptr(), d_y.data_ptr(), batch, dev.stream)
        back = dev.to_host(d_y)
        for p, t in enumerate(tabs):
            sl = slice(p * batch * n, (p + 1) * batch * n)
            assert np.array_equal(back[sl], x[sl] % np.uint64(t[0])), (case, n, bits, "inverse")
        # the polynomial product of the case's input with a second random operand, c aliasing either operand at random
        b = np.concatenate([rand_coeffs(rng, batch * n, t[0], hi_mult=int(rng.integers(1, 5))) for t in tabs])
        d_a, d_b, d_s = dev.to_device(x), dev.to_device(b), dev.empty(x.size)
        alias = int(rng.integers(0, 3))
        d_c = (dev.empty(x.size), d_a, d_b)[alias]
        plan.polymul(d_a.data_ptr(), d_b.data_ptr(), d_c.data_ptr(), d_s.data_ptr(), batch, dev.stream)
        c = dev.to_host(d_c)
        for p, t in enumerate(tabs):
            for f in range(batch):
                sl = slice((p * batch + f) * n, (p * batch + f + 1) * n)
                wantc = orc.schoolbook(x[sl] % np.uint64(t[0]), b[sl] % np.uint64(t[0]), t[0], n) if n <= 1024 else _oracle_polymul(orc, x[sl], b[sl], t[0], t[1], n)
                assert np.array_equal(c[sl], wantc), (case, n, bits, "polymul", alias)
        plan.close()
