"""GPU tests of the library-level multi-GPU driver (include/agx_ntt.h section 5, csrc/agx_group.cpp).  A one-GPU box exercises it
with the device list [0, 0]: two shards -- two plans, two streams, two staging sets, two host threads -- on one GPU, which is the
same code path as two GPUs except for the hipSetDevice argument.  Also here: the argument checks the
device-pointer calls gained in round 4 (partial overlap of in / out) and the hipStreamPerThread rule of the ticket kernels."""
import ctypes
import threading

import numpy as np
import pytest

from gpu_util import rand_coeffs, tables_for

pytestmark = pytest.mark.gpu


def _oracle_forward(orc, x, t, n):
    return orc.forward(x, t[0], t[2], t[3], n)


@pytest.mark.parametrize("n,frames", [(4096, 1101), (16384, 301)])
def test_group_host_frames_on_two_shards_of_one_gpu(agx, orc, n, frames):
    """agx_ntt_group_forward_host on devices [0, 0]: contiguous blocks (551 + 550 / 151 + 150 frames, tens of MiB each: the streaming
    pipeline, not the small-input path), lower half of every frame from `in`, upper half from `in2` (src/kernel/ntt.cpp:584-590) with
    garbage in the halves that must not be read; bit-exact against the oracle; then the inverse brings the data back"""
    t = tables_for(orc, n, 60)[0]
    q = t[0]
    grp = agx.DeviceGroup([0, 0], n, [q], psi=[t[1]])
    assert grp.num_shards == 2 and [grp.shard(i)[0] for i in range(2)] == [0, 0]
    assert grp.shard(0)[2] != grp.shard(1)[2] and grp.shard(0)[1] != grp.shard(1)[1]      # own stream, own plan
    rng = np.random.default_rng(n + frames)
    x = rand_coeffs(rng, frames * n, q, hi_mult=4)
    a, b = x.copy().reshape(frames, n), x.copy().reshape(frames, n)
    a[:, n // 2:] = np.uint64(0xFFFFFFFFFFFFFFFF)      # never read
    b[:, :n // 2] = np.uint64(0xFFFFFFFFFFFFFFFF)
    out = np.full(frames * n, 7, dtype=np.uint64)
    got = grp.forward_host(a.reshape(-1), b.reshape(-1), frames, out=out)
    want = _oracle_forward(orc, x, t, n)
    assert np.array_equal(got, want)
    back = grp.inverse_host(got, frames)
    assert np.array_equal(back, x % np.uint64(q))
    # in2 == in, a second call on the same group (the shards keep their staging sets)
    assert np.array_equal(grp.forward_host(x, x, frames), want)
    grp.close()


def test_group_more_shards_than_frames_and_zero_frames(agx, orc):
    n = 1024
    t = tables_for(orc, n, 30)[0]
    grp = agx.DeviceGroup([0, 0, 0], n, [t[0]], psi=[t[1]])
    rng = np.random.default_rng(5)
    for frames in (0, 1, 2, 3, 4):
        x = rand_coeffs(rng, max(frames, 1) * n, t[0])[: frames * n]
        got = grp.forward_host(x, x, frames) if frames else grp.forward_host(np.zeros(1, dtype=np.uint64), np.zeros(1, dtype=np.uint64), 0, out=np.zeros(1, dtype=np.uint64))
        if frames:
            assert np.array_equal(got, _oracle_forward(orc, x, t, n)), frames
    grp.close()


def test_group_device_pointer_calls(agx, orc, dev):
    """one pointer and one batch per shard: forward, inverse and the one-launch product on two shards with DIFFERENT batch sizes,
    4 primes, every shard from its own thread on its own stream; synchronize; against the oracle"""
    import torch

    n, primes = 4096, 2
    tabs = tables_for(orc, n, 60, primes)
    grp = agx.DeviceGroup([0, 0], n, [t[0] for t in tabs], psi=[t[1] for t in tabs])
    batches = [37, 5]
    rng = np.random.default_rng(11)
    xs = [np.concatenate([rand_coeffs(rng, b * n, t[0], hi_mult=4) for t in tabs]) for b in batches]
    d_in = [dev.to_device(x) for x in xs]
    d_out = [dev.empty(x.size) for x in xs]
    torch.cuda.synchronize()
    grp.forward([d.data_ptr() for d in d_in], [d.data_ptr() for d in d_out], batches)
    grp.synchronize()
    for x, d, b in zip(xs, d_out, batches):
        want = np.concatenate([_oracle_forward(orc, x[p * b * n:(p + 1) * b * n], t, n) for p, t in enumerate(tabs)])
        assert np.array_equal(dev.to_host(d), want)
    grp.inverse([d.data_ptr() for d in d_out], [d.data_ptr() for d in d_out], batches)      # in place
    grp.synchronize()
    for x, d, b in zip(xs, d_out, batches):
        want = np.concatenate([x[p * b * n:(p + 1) * b * n] % np.uint64(t[0]) for p, t in enumerate(tabs)])
        assert np.array_equal(dev.to_host(d), want)
    # product by X (a shift with sign): c = a * X on both shards, no scratch
    xpoly = [np.zeros(x.size, dtype=np.uint64) for x in xs]
    for xp in xpoly:
        xp[1::n] = 1
    d_x = [dev.to_device(xp) for xp in xpoly]
    torch.cuda.synchronize()
    grp.polymul([d.data_ptr() for d in d_in], [d.data_ptr() for d in d_x], [d.data_ptr() for d in d_out], batches)
    grp.synchronize()
    for x, d, b in zip(xs, d_out, batches):
        got = dev.to_host(d).reshape(primes, b, n)
        for p, t in enumerate(tabs):
            src = (x[p * b * n:(p + 1) * b * n] % np.uint64(t[0])).reshape(b, n)
            want = np.roll(src, 1, axis=1)
            want[:, 0] = (np.uint64(t[0]) - want[:, 0]) % np.uint64(t[0])      # X^n = -1
            assert np.array_equal(got[p], want)
    # a zero batch on one shard is a no-op there
    grp.forward([d_in[0].data_ptr(), 0], [d_out[0].data_ptr(), 0], [batches[0], 0])
    grp.synchronize()
    grp.close()


def test_group_errors(agx, orc):
    n = 1024
    t = tables_for(orc, n, 30)[0]
    with pytest.raises(agx.AgxError) as ei:
        agx.DeviceGroup([0, 99], n, [t[0]])      # bad device id
    assert ei.value.status == 5
    with pytest.raises(agx.AgxError) as ei:
        agx.DeviceGroup([-1], n, [t[0]])
    assert ei.value.status == 5
    with pytest.raises(agx.AgxError) as ei:
        agx.DeviceGroup([0], n, [t[0] + 2])      # not = 1 mod 2n / not prime
    assert ei.value.status == 3
    two = agx.DeviceGroup([0, 0], n, [t[0], tables_for(orc, n, 30, 2)[1][0]])
    x = np.zeros(n, dtype=np.uint64)
    with pytest.raises(agx.AgxError) as ei:
        two.forward_host(x, x, 1)      # host frames: one modulus per call, as the reference (src/kernel/ntt.cpp:143-144)
    assert ei.value.status == 5
    with pytest.raises(agx.AgxError) as ei:
        two.forward([0, 0], [0, 0], [1, 1])      # null device pointers: the first failing shard's status
    assert ei.value.status == 1
    two.close()
    # a forward-only group has no inverse
    tw, pre = agx.make_tables(t[0], t[1], n)
    fwd_only = agx.DeviceGroup([0], n, [t[0]], tables=(tw[None, :], pre[None, :]))
    with pytest.raises(agx.AgxError) as ei:
        fwd_only.inverse_host(x, 1)
    assert ei.value.status == 9
    assert np.array_equal(fwd_only.forward_host(x, x, 1), x)      # NTT(0) = 0
    fwd_only.close()


def test_partial_overlap_of_in_and_out_is_rejected(agx, orc, dev):
    """include/agx_ntt.h: 'overlapping in/out' is AGX_ERR_BAD_ARGUMENT: d_out = d_in + n/2 and c = a + 8 must
    return status 5 and leave memory untouched; equal pointers (in place) and interleaved frame sets that never touch stay legal"""
    n, batch = 4096, 6
    t = tables_for(orc, n, 60)[0]
    plan = agx.Plan(n, [t[0]], psi=[t[1]])
    rng = np.random.default_rng(3)
    x = rand_coeffs(rng, 3 * batch * n, t[0])
    d = dev.to_device(x)
    base = d.data_ptr()
    for off in (n // 2, 8, n - 1, (batch - 1) * n + 5):
        for call in (lambda: plan.forward(base, base + 8 * off, batch, dev.stream),
                     lambda: plan.inverse(base, base + 8 * off, batch, dev.stream),
                     lambda: plan.forward(base + 8 * off, base, batch, dev.stream),
                     lambda: plan.polymul(base, base + 8 * 2 * batch * n, base + 8 * off, 0, batch, dev.stream),
                     lambda: plan.polymul(base + 8 * 2 * batch * n, base, base + 8 * off, 0, batch, dev.stream),
                     lambda: plan.pointwise(base, base + 8 * 2 * batch * n, base + 8 * off, batch, dev.stream)):
            with pytest.raises(agx.AgxError) as ei:
                call()
            assert ei.value.status == 5, off
    assert np.array_equal(dev.to_host(d), x), "a rejected call must not touch memory"
    # legal: disjoint (out = in + batch n), in place, and interleaved frames (poly_stride = 2n, out = in + n: frames never touch)
    plan.forward(base, base + 8 * batch * n, batch, dev.stream)
    want = _oracle_forward(orc, x[: batch * n], t, n)
    assert np.array_equal(dev.to_host(d)[batch * n:2 * batch * n], want)
    d2 = dev.to_device(x)
    plan.forward_strided(d2.data_ptr(), d2.data_ptr() + 8 * n, batch, 0, 2 * n, dev.stream)
    got = dev.to_host(d2).reshape(-1, n)
    evens = x.reshape(-1, n)[0:2 * batch:2]
    assert np.array_equal(got[1:2 * batch:2].reshape(-1), _oracle_forward(orc, evens.reshape(-1), t, n))
    assert np.array_equal(got[0:2 * batch:2], evens)
    with pytest.raises(agx.AgxError) as ei:      # ... but shifted by half a frame they do touch
        plan.forward_strided(d2.data_ptr(), d2.data_ptr() + 8 * (n + n // 2), batch, 0, 2 * n, dev.stream)
    assert ei.value.status == 5
    plan.close()


def test_stream_per_thread_from_two_host_threads(agx, orc, dev):
    """hipStreamPerThread is ONE handle that names a different stream in every host thread: two threads running the
    n = 16384 inverse -- whose default kernel draws frames from a per-stream ticket pair -- on it at the same time must both be
    right (the launcher gives that handle no pair: stateless kernel)"""
    import torch

    n, batch = 16384, 700      # more frames than resident workgroups
    t = tables_for(orc, n, 60)[0]
    plan = agx.Plan(n, [t[0]], psi=[t[1]])
    itw = orc.make_inv_tables(t[0], t[1], n)[0]
    rng = np.random.default_rng(9)
    xs = [rand_coeffs(rng, batch * n, t[0]) for _ in range(2)]
    wants = [orc.inverse(x, t[0], itw, n) for x in xs]
    ds = [dev.to_device(x) for x in xs]
    outs = [dev.empty(x.size) for x in xs]
    torch.cuda.synchronize()
    STREAM_PER_THREAD = 2      # hipStreamPerThread
    errs = []

    def work(k):
        try:
            torch.cuda.set_device(0)
            for _ in range(4):
                plan.inverse(ds[k].data_ptr(), outs[k].data_ptr(), batch, STREAM_PER_THREAD)
        except Exception as e:      # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    torch.cuda.synchronize()
    assert not errs, errs
    for k in range(2):
        assert np.array_equal(dev.to_host(outs[k]), wants[k]), k
    plan.close()


def test_harness_scaling_mode_on_two_shards(agx):
    """bin/ntt_harness --devices 0,0 --small: agx::ntt / agx::intt with a device list against the single-device result (1,101 frames of
    n = 4096, 301 of n = 16384, 3 of n = 1024 -- more shards than would fill), then BASELINE configs[2] and configs[3]'s slice (batches
    divided by 8) through agx_ntt_group_forward with per-shard HIP events; as a child process"""
    import os
    import subprocess

    exe = os.path.join(os.path.dirname(agx.LIB_PATH), "..", "bin", "ntt_harness")
    r = subprocess.run([exe, "--devices", "0,0", "--small", "--steps", "20"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "SCALING PASSED" in r.stdout and "FAIL" not in r.stdout, r.stdout[-3000:]
    assert r.stdout.count("== one device") == 3 and r.stdout.count("shard 1 (device 0)") == 2 and r.stdout.count("aggregate:") == 2, r.stdout[-3000:]
    print(r.stdout)


def test_group_calls_from_two_host_threads(agx, orc):
    """calls on ONE group from two host threads serialise (include/agx_ntt.h: one call at a time per group), calls on two groups run
    side by side; every result right, repeatedly"""
    n, frames = 4096, 700
    t = tables_for(orc, n, 60)[0]
    g1 = agx.DeviceGroup([0, 0], n, [t[0]], psi=[t[1]])
    g2 = agx.DeviceGroup([0], n, [t[0]], psi=[t[1]])
    rng = np.random.default_rng(21)
    xs = [rand_coeffs(rng, frames * n, t[0]) for _ in range(3)]
    wants = [_oracle_forward(orc, x, t, n) for x in xs]
    outs = [None] * 3
    errs = []

    def work(k, grp):
        try:
            for _ in range(3):
                outs[k] = grp.forward_host(xs[k], xs[k], frames)
        except Exception as e:      # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=work, args=(0, g1)), threading.Thread(target=work, args=(1, g1)), threading.Thread(target=work, args=(2, g2))]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errs, errs
    for k in range(3):
        assert np.array_equal(outs[k], wants[k]), k
    g1.close()
    g2.close()


def test_one_shot_call_over_a_device_list_from_the_environment(agx, orc):
    """AGX_NTT_DEVICES=0,0: agx_ntt_forward_host -- the drop-in for the reference's three calls -- deals its frames to the listed devices
    without a change in the caller (the reference's NUM_NTT_COMPUTE_UNITS replication, src/kernel/ntt.cpp:8-12, 526-536): same bits as
    on one device, in2 != in included; a device that does not exist is status 5; the cached group follows the caller's tables"""
    import os

    n, frames = 4096, 900
    t = tables_for(orc, n, 60)[0]
    rng = np.random.default_rng(8)
    x = rand_coeffs(rng, frames * n, t[0])
    a, b = x.copy().reshape(frames, n), x.copy().reshape(frames, n)
    a[:, n // 2:] = np.uint64(1)
    b[:, :n // 2] = np.uint64(2)
    want = _oracle_forward(orc, x, t, n)
    old = os.environ.get("AGX_NTT_DEVICES")
    try:
        os.environ["AGX_NTT_DEVICES"] = "0,0"
        assert np.array_equal(agx.forward_host(a.reshape(-1), b.reshape(-1), t[0], t[2], t[3], n, frames), want)
        assert np.array_equal(agx.forward_host(x, x, t[0], t[2], t[3], n, frames), want)      # cached group
        t2 = tables_for(orc, n, 59)[0]                                                         # other tables: the group is rebuilt
        x2 = x % np.uint64(t2[0])
        assert np.array_equal(agx.forward_host(x2, x2, t2[0], t2[2], t2[3], n, frames), _oracle_forward(orc, x2, t2, n))
        os.environ["AGX_NTT_DEVICES"] = "0,77"
        with pytest.raises(agx.AgxError) as ei:
            agx.forward_host(x, x, t[0], t[2], t[3], n, frames)
        assert ei.value.status == 5
    finally:
        if old is None:
            os.environ.pop("AGX_NTT_DEVICES", None)
        else:
            os.environ["AGX_NTT_DEVICES"] = old
        assert agx.lib().agx_ntt_release_caches() == 0
    assert np.array_equal(agx.forward_host(x, x, t[0], t[2], t[3], n, frames), want)      # back on the current device


def test_bench_two_rank_control_flow_rehearsed_on_one_gpu(agx):
    """bench.py --gpus 2 under torch.distributed.run with AGX_BENCH_REHEARSAL=1: both ranks on GPU 0, timing collectives over gloo -- the
    N > 1 bookkeeping of bench.main() (shard offsets, barriers, MAX over ranks, per_rank gather, one JSON line from rank 0) runs on
    hardware (it had only ever executed with one rank).  Not a multi-GPU figure; the line says "rehearsal": true."""
    import json
    import os
    import socket
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, AGX_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "3", "--batch", "1024", "--ramp-seconds", "0.1"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]      # ONE line, from rank 0
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["config"]["rehearsal"] is True and j["scaling"] == "weak"
    assert len(j["per_rank"]) == 2 and [p["rank"] for p in j["per_rank"]] == [0, 1]
    assert all(p["kernel_ms"] > 0 and p["elapsed_s"] > 0 for p in j["per_rank"])
    # value = all ranks' NTTs over the slowest rank's time
    slowest = max(p["elapsed_s"] for p in j["per_rank"])
    assert abs(j["value"] - 2 * 4 * 1024 * 20 / slowest) / j["value"] < 1e-6
    assert "secondary" not in j and "cpu_baseline" not in j      # N = 1 only
    print(lines[0][:600])
