import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import agilex_ntt_amd as agx
from oracle import oracle as orc
from gpu_util import DeviceHelper
orc.build()
dev = DeviceHelper(torch)
n = 32768
for bits, ids in ((60, (None, 114, 43)),):
    q = orc.find_prime(bits, n); psi = orc.min_root(q, n)
    itw = orc.make_inv_tables(q, psi, n)[0]
    for topmul in (4, 1):
        if bits == 62 and topmul == 4: continue
        top = topmul * q - 1
        x = np.concatenate([np.full(n, top, dtype=np.uint64), np.full(n, q - 1, dtype=np.uint64),
                            np.where(np.arange(n) % 2 == 0, np.uint64(top), np.uint64(0)),
                            np.where(np.arange(n) % 3 == 0, np.uint64(top), np.uint64(top - q + 1))])
        want = orc.inverse(x % np.uint64(q), q, itw, n)
        for cid in ids:
            plan = agx.Plan(n, [q], psi=[psi])
            if cid is not None: plan.set_variant(agx.VARIANT_REGBLOCK_BASE + cid)
            d = dev.to_device(x)
            plan.inverse(d.data_ptr(), d.data_ptr(), 4, dev.stream)
            got = dev.to_host(d)
            bad = np.nonzero(got != want)[0]
            print(bits, "top", topmul, "id", cid, "mismatches", bad.size, "first", bad[:8], "frames", sorted(set((bad // n).tolist())))
            plan.close()
