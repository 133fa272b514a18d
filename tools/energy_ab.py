"""tools/energy_ab.py -- energy per transform of registered kernel configurations, one after the other in ONE process.

At the board's power cap throughput follows the energy of one transform, so an A/B of two kernel shapes has to
report micro-joules per NTT beside milliseconds.  Each configuration runs back-to-back launches for --seconds while amdgpu's sysfs
files are sampled every 20 ms (bench.py's PowerSampler); idle power is sampled first.  Ids follow tools/sweep.py
(AGX_VARIANT_REGBLOCK_BASE + id, -2 = the plan's tuned default); the timing-only ablation twins (67-72) need
`make -C agilex-ntt_amd diag EXTRA=-DAGX_TIMING_ABLATIONS` and AGX_NTT_LIB=agilex-ntt_amd/lib/libagxntt_diag.so.

Usage: python3 tools/energy_ab.py [--n N --primes P --batch B --bits 60 --op fwd|inv|mul --seconds 1.5 --rounds 2] ids..."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import agilex_ntt_amd as agx  # noqa: E402
from bench import PowerSampler  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("ids", nargs="*", type=int)
ap.add_argument("--n", type=int, default=4096)
ap.add_argument("--primes", type=int, default=4)
ap.add_argument("--batch", type=int, default=4096)
ap.add_argument("--bits", type=int, default=60)
ap.add_argument("--slabs", type=int, default=4)
ap.add_argument("--op", choices=["fwd", "inv", "mul"], default="fwd")
ap.add_argument("--seconds", type=float, default=1.5)
ap.add_argument("--rounds", type=int, default=2)
ap.add_argument("--zeros", action="store_true", help="all-zero coefficients (no operand toggling) instead of uniformly random ones")
args = ap.parse_args()
ids = args.ids or [-2]
N, P, B = args.n, args.primes, args.batch
plan = agx.Plan(N, agx.find_primes(args.bits, N, P))
stream = torch.cuda.current_stream().cuda_stream
slabs = [torch.zeros(P * B * N, dtype=torch.int64, device="cuda") for _ in range(args.slabs)]
if not args.zeros:
    for i, s in enumerate(slabs):
        plan.fill_synthetic(s.data_ptr(), B, i * B, 42, stream)
torch.cuda.synchronize()
sampler = PowerSampler(torch, 0)
t0 = time.perf_counter()
time.sleep(0.6)
idle = sampler.median_w(t0, time.perf_counter())


def run(i):
    a, b = slabs[i % args.slabs], slabs[(i + 1) % args.slabs]
    if args.op == "fwd":
        plan.forward(a.data_ptr(), a.data_ptr(), B, stream)
    elif args.op == "inv":
        plan.inverse(a.data_ptr(), a.data_ptr(), B, stream)
    else:
        plan.polymul(a.data_ptr(), b.data_ptr(), a.data_ptr(), 0, B, stream)


print(f"{args.op} n={N} primes={P} ({args.bits} bits) batch={B}{' ZERO data' if args.zeros else ''}; idle {idle} W; {args.seconds} s per line, first 0.3 s of each dropped")
print(f"{'id':>4} {'ms/launch':>10} {'M units/s':>10} {'%8TB/s':>7} {'W median':>9} {'sclk MHz':>9} {'uJ/unit':>9}")
for rnd in range(args.rounds):
    for k in ids:
        plan.set_variant(agx.VARIANT_AUTO if k == -2 else agx.VARIANT_LDS_RADIX2 if k < 0 else agx.VARIANT_REGBLOCK_BASE + k)
        for i in range(8):
            run(i)
        torch.cuda.synchronize()
        launches, t_start = 0, time.perf_counter()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        while time.perf_counter() - t_start < args.seconds:
            for i in range(64):
                run(launches + i)
            launches += 64
            torch.cuda.synchronize()
        e1.record()
        torch.cuda.synchronize()
        t_end = time.perf_counter()
        ms = e0.elapsed_time(e1) / launches
        w = sampler.window(t_start + 0.3, t_end)
        rate = P * B / (ms * 1e-3)
        watts = w["socket_power_w_median"] if w else None
        uj = (watts - idle) / rate * 1e6 if watts is not None and idle is not None else float("nan")
        bpu = (24 if args.op == "mul" else 16) * N
        print(f"{k:>4} {ms:10.4f} {rate / 1e6:10.2f} {rate * bpu / 8e12 * 100:7.2f} {watts!s:>9} {(w['sclk_mhz_median'] if w else None)!s:>9} {uj:9.2f}")
sampler.stop()
plan.close()
