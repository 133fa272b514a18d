import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import agilex_ntt_amd as agx
from oracle import oracle as orc
from gpu_util import DeviceHelper, rand_coeffs
orc.build()
dev = DeviceHelper(torch)
for n, bits in ((32768, 17), (32768, 20), (32768, 30), (16384, 17), (4096, 17)):
    q = orc.find_prime(bits, n); psi = orc.min_root(q, n)
    tw, pre = orc.make_tables(q, psi, n)
    for batch in (3, 64):
        x = rand_coeffs(np.random.default_rng(1), batch * n, q, hi_mult=4)
        want = orc.forward(x, q, tw, pre, n)
        for variant in ("auto", "radix2"):
            plan = agx.Plan(n, [q], psi=[psi])
            if variant == "radix2": plan.set_variant(agx.VARIANT_LDS_RADIX2)
            for oop in (True, False):
                d = dev.to_device(x); o = dev.empty(x.size) if oop else d
                plan.forward(d.data_ptr(), o.data_ptr(), batch, dev.stream)
                got = dev.to_host(o)
                bad = np.nonzero(got != want)[0]
                print(n, bits, "q", q, "batch", batch, variant, "oop" if oop else "ip", "mismatches", bad.size, "first", bad[:4], "max", int(got.max()))
            plan.close()
