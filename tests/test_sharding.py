"""CPU tests of the N>1 path: one process per device, contiguous blocks of independent
polynomials per rank, no collective on the data path; only timing is reduced.  Run with
world size 2 over gloo (the GPU job uses the same Group class over RCCL)."""
import os
import socket
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, json
sys.path.insert(0, os.environ["AGX_ROOT"])
import numpy as np
import agilex_ntt_amd as agx
from oracle import oracle as orc          # stands in for the device engine in this CPU test

grp = agx.Group(backend="gloo")
n, primes, total = 256, 2, 11              # ragged: 11 polynomials over 2 ranks
lo, hi = agx.shard_range(total, grp.rank, grp.world)
qs = [orc.find_prime(60, n, k) for k in range(primes)]
out = {}
for p, q in enumerate(qs):
    psi = orc.min_root(q, n)
    tw, pre = orc.make_tables(q, psi, n)
    x = orc.fill_splitmix(total * n, 100 + p, q)          # every rank derives the same global batch
    out[p] = orc.forward(x[lo * n:hi * n], q, tw, pre, n)  # and transforms only its own block
np.save(os.path.join(os.environ["AGX_OUT"], f"rank{grp.rank}.npy"), np.concatenate([out[p] for p in range(primes)]))
grp.barrier()
elapsed = grp.max_over_ranks(1.0 + grp.rank)              # slowest rank defines the step time
count = grp.sum_over_ranks(hi - lo)
value = agx.aggregate_throughput(5, 10, grp.world, elapsed)
rows = grp.gather_rows([1.0 + grp.rank, 0.25 * (grp.rank + 1), None, 1800 + grp.rank])   # elapsed, kernel_ms, watts (unreadable), sclk
if grp.rank == 0:
    json.dump({"elapsed": elapsed, "count": count, "value": value, "world": grp.world, "range": [lo, hi], "rows": rows},
              open(os.path.join(os.environ["AGX_OUT"], "rank0.json"), "w"))
grp.close()
'''


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_shard_range_partitions_every_batch(agx):
    for total in (0, 1, 7, 11, 4096, 65536):
        for world in (1, 2, 3, 8):
            blocks = [agx.shard_range(total, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == total
            for (a, b), (c, d) in zip(blocks, blocks[1:]):
                assert b == c and a <= b
            sizes = [b - a for a, b in blocks]
            assert max(sizes) - min(sizes) <= 1


def test_world_size_2_gloo(tmp_path, orc):
    import json

    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), AGX_ROOT=ROOT, AGX_OUT=str(tmp_path))
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    for p in procs:
        out, err = p.communicate(timeout=240)
        assert p.returncode == 0, err.decode()[-2000:]
    meta = json.load(open(tmp_path / "rank0.json"))
    assert meta["world"] == 2 and meta["count"] == 11 and meta["range"] == [0, 5]
    assert meta["elapsed"] == 2.0                       # MAX over ranks, not rank 0's own time
    assert meta["value"] == 5 * 10 * 2 / 2.0            # whole-job units / slowest rank's time
    # the per-rank table bench.py prints as `per_rank`: one row per rank, in rank order, None carried as NaN
    rows = meta["rows"]
    assert len(rows) == 2 and [r[0] for r in rows] == [1.0, 2.0] and [r[1] for r in rows] == [0.25, 0.5] and [r[3] for r in rows] == [1800.0, 1801.0]
    assert all(r[2] != r[2] for r in rows)
    # concatenating the ranks' blocks reproduces the single-process transform: nothing was exchanged
    n, primes, total = 256, 2, 11
    r0, r1 = np.load(tmp_path / "rank0.npy"), np.load(tmp_path / "rank1.npy")
    for p in range(primes):
        q = orc.find_prime(60, n, p)
        tw, pre = orc.make_tables(q, orc.min_root(q, n), n)
        whole = orc.forward(orc.fill_splitmix(total * n, 100 + p, q), q, tw, pre, n)
        got = np.concatenate([r0[p * 5 * n:(p + 1) * 5 * n], r1[p * 6 * n:(p + 1) * 6 * n]])
        assert np.array_equal(got, whole)


# ---- the library-level group (include/agx_ntt.h section 5): block arithmetic and argument checks need no device ----------------
def test_group_block_arithmetic_matches_the_reference_minibatches(agx):
    """agx_ntt_shard_range deals F frames to C shards in the reference's minibatch sizes -- floor(F / C) + [i < F mod C]
    (src/kernel/ntt.cpp:526-536) -- as contiguous blocks that tile [0, F): ragged counts, more shards than frames, zero frames"""
    for frames, shards in [(0, 1), (0, 5), (1, 1), (1, 8), (3, 8), (7, 8), (8, 8), (9, 8), (11, 2), (4096, 8), (4097, 8), (65536, 3), (2**40 + 5, 7)]:
        nxt = 0
        for i in range(shards):
            first, count = agx.shard_block(frames, shards, i)
            assert first == nxt
            assert count == frames // shards + (1 if i < frames % shards else 0)
            nxt = first + count
        assert nxt == frames
    import pytest

    for bad in [(10, 0, 0), (10, 2, 2), (10, 2, 7)]:
        with pytest.raises(agx.AgxError) as ei:
            agx.shard_block(*bad)
        assert ei.value.status == 5
    assert agx.lib().agx_ntt_shard_range(10, 2, 0, None, None) == 1


def test_group_calls_fail_loudly_without_a_device(agx):
    """no GPU here: creating a group must say AGX_ERR_NO_DEVICE (6), never fall back; argument checks come first"""
    import ctypes

    import pytest
    import torch

    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU")
    L = agx.lib()
    h = ctypes.c_void_p(None)
    devs = (ctypes.c_int * 2)(0, 0)
    q = np.array([agx.find_primes(60, 4096)[0]], dtype=np.uint64)
    p64 = ctypes.POINTER(ctypes.c_uint64)
    qp = q.ctypes.data_as(p64)
    assert L.agx_ntt_group_create_auto(ctypes.byref(h), devs, 2, 4096, 1, qp, None) == 6 and not h.value
    assert L.agx_ntt_group_create_auto(ctypes.byref(h), devs, 0, 4096, 1, qp, None) == 5      # no shards
    assert L.agx_ntt_group_create_auto(ctypes.byref(h), devs, 2, 4095, 1, qp, None) == 2      # bad size
    assert L.agx_ntt_group_create_auto(None, devs, 2, 4096, 1, qp, None) == 1
    assert L.agx_ntt_group_create_auto(ctypes.byref(h), None, 2, 4096, 1, qp, None) == 1
    assert L.agx_ntt_group_forward_host(None, qp, qp, qp, 1) == 1
    assert L.agx_ntt_group_synchronize(None) == 1
    assert L.agx_ntt_group_destroy(None) == 0
