"""tools/data_dependence_probe.py -- is the headline kernel power-limited?  The SAME launch (n=4096, 4 primes, batch 4096, in place,
4 rotating slabs) on (a) uniformly random coefficients, (b) all-zero coefficients: identical instruction stream and memory traffic,
but no operand toggling in the multipliers.  If the board is at its power cap on (a), (b) must run faster at a higher clock and
lower power.  Prints ms per launch, socket power and shader clock (amdgpu sysfs, same sampler as bench.py) for ~1.5 s of each."""
import importlib.util
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import agilex_ntt_amd as agx  # noqa: E402

spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)

N, P, B, SLABS = 4096, 4, 4096, 4
plan = agx.Plan(N, agx.find_primes(60, N, P))
stream = torch.cuda.current_stream().cuda_stream
slabs = [torch.empty(P * B * N, dtype=torch.int64, device="cuda") for _ in range(SLABS)]
sampler = bench.PowerSampler(torch, 0)


def run(label, refill):
    for s in slabs:
        refill(s)
    torch.cuda.synchronize()
    t_end = time.perf_counter() + 0.4
    i = 0
    while time.perf_counter() < t_end:          # clock ramp / settle
        for _ in range(64):
            plan.forward(slabs[i % SLABS].data_ptr(), slabs[i % SLABS].data_ptr(), B, stream)
            i += 1
        torch.cuda.synchronize()
    launches = 4000
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for k in range(launches):
        if refill is zero and False:
            pass
        plan.forward(slabs[k % SLABS].data_ptr(), slabs[k % SLABS].data_ptr(), B, stream)
    e1.record()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    ms = e0.elapsed_time(e1) / launches
    w = sampler.window(t0 + 0.1, t1) or {}
    print(f"{label:<44} {ms:.4f} ms/launch  {P * B / ms / 1e3:6.2f} M NTT/s  {w.get('socket_power_w_median')} W  sclk {w.get('sclk_mhz_median')} MHz  ({w.get('samples')} samples)")


def rand(s, k=[0]):
    plan.fill_synthetic(s.data_ptr(), B, k[0] * B, 42, stream)
    k[0] += 1


def zero(s):
    s.zero_()


# NB: in place, repeated transforms of a slab stay uniformly distributed (random case) / stay zero (zero case)
run("uniformly random coefficients (the bench workload)", rand)
run("all-zero coefficients", zero)
run("uniformly random coefficients again", rand)
sampler.stop()
plan.close()
