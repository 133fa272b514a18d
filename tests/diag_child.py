"""Child process of tests/test_gpu_diag.py: runs with AGX_NTT_LIB = lib/libagxntt_diag.so (the product library plus the trace twin
of csrc/reg_diag.hip and the A/B twins of the registry groups) and checks those kernels against the oracle.  Prints DIAG OK on success."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import agilex_ntt_amd as agx  # noqa: E402
from gpu_util import DeviceHelper, rand_coeffs, tables_for  # noqa: E402
from oracle import oracle as orc  # noqa: E402

assert agx.LIB_PATH.endswith("libagxntt_diag.so"), agx.LIB_PATH
orc.build()
dev = DeviceHelper(torch)
n = 4096


def plan_for(bits, primes):
    tabs = tables_for(orc, n, bits, primes)
    plan = agx.Plan(n, [t[0] for t in tabs], tables=(np.stack([t[2] for t in tabs]), np.stack([t[3] for t in tabs])))
    return plan, tabs


def oracle_forward(x, tabs, batch):
    return np.concatenate([orc.forward(x[p * batch * n:(p + 1) * batch * n], t[0], t[2], t[3], n) for p, t in enumerate(tabs)])


# every diagnostics entry against the oracle (30- and 60-bit moduli); 16q-lazy kernels must refuse a 61-bit modulus
for config in (70, 147, 161):
    for bits in (30, 60):
        plan, tabs = plan_for(bits, 2)
        plan.set_variant(agx.VARIANT_REGBLOCK_BASE + config)
        rng = np.random.default_rng(config * 10 + bits)
        x = np.concatenate([rand_coeffs(rng, 3 * n, t[0], hi_mult=4) for t in tabs])
        d = dev.to_device(x)
        plan.forward(d.data_ptr(), d.data_ptr(), 3, dev.stream)
        assert np.array_equal(dev.to_host(d), oracle_forward(x, tabs, 3)), (config, bits)
        plan.close()
    plan, tabs = plan_for(61, 1)
    try:
        plan.set_variant(agx.VARIANT_REGBLOCK_BASE + config)
        raise SystemExit(f"config {config} accepted a 61-bit modulus")
    except agx.AgxError as e:
        assert e.status == 2
    plan.close()

# the trace hook exists here (and only here) and the trace twin fills the buffer
plan, tabs = plan_for(60, 1)
plan.set_variant(agx.VARIANT_REGBLOCK_BASE + 70)
waves = 8 * 4
trace = torch.zeros(waves * 16, dtype=torch.int64, device="cuda")
agx.debug_set_trace_buffer(trace.data_ptr(), trace.numel() * 8)
d = dev.to_device(rand_coeffs(np.random.default_rng(1), 4 * n, tabs[0][0]))
plan.forward(d.data_ptr(), d.data_ptr(), 4, dev.stream)
torch.cuda.synchronize()
agx.debug_set_trace_buffer(0, 0)
t = trace.cpu().numpy().reshape(waves, 16)
assert (t[:, 0] > 0).all() and (np.diff(t[:, :12].astype(np.uint64), axis=1).astype(np.int64) >= 0).all()
plan.close()
print("DIAG OK")
