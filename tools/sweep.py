"""tools/sweep.py -- A/B the registered kernel configurations on the roofline workload
(default n=4096, 4 primes, batch 4096; 4 rotating slabs; --n/--primes/--batch for other shapes), all in ONE process, interleaved rounds
(cdna_hip_programming.md rule 24).  Every configuration's output is compared bit for bit with
the radix-2 kernel's before it is timed.  Usage: python tools/sweep.py [--n N --primes P --batch B] [ids...]   (id -1 = the radix-2 kernel)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import agilex_ntt_amd as agx  # noqa: E402

import argparse  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("ids", nargs="*", type=int)
ap.add_argument("--n", type=int, default=4096)
ap.add_argument("--primes", type=int, default=4)
ap.add_argument("--batch", type=int, default=4096)
ap.add_argument("--slabs", type=int, default=4)
ap.add_argument("--bits", type=int, default=60, help="modulus size in bits")
ap.add_argument("--op", choices=["fwd", "inv", "mul"], default="fwd")
ap.add_argument("--launches", type=int, default=20, help="back-to-back launches per timing (20 = burst; 100+ shows the sustained clock)")
ap.add_argument("--oop", action="store_true", help="time out of place (slab i -> slab i+1) instead of in place")
args = ap.parse_args()
N, P, B, SLABS = args.n, args.primes, args.batch, args.slabs
ids = args.ids or [93]
qs = agx.find_primes(args.bits, N, P)
plan = agx.Plan(N, qs)
stream = torch.cuda.current_stream().cuda_stream
per = P * B * N
slabs = [torch.empty(per, dtype=torch.int64, device="cuda") for _ in range(SLABS)]
for i, s in enumerate(slabs):
    plan.fill_synthetic(s.data_ptr(), B, i * B, 42, stream)
ref_in = slabs[0].clone()
ref_in2 = slabs[1 % SLABS].clone()
scratch = torch.empty_like(ref_in)
ref = torch.empty_like(ref_in)


def run(src, dst, src2=None):
    if args.op == "fwd":
        plan.forward(src.data_ptr(), dst.data_ptr(), B, stream)
    elif args.op == "inv":
        plan.inverse(src.data_ptr(), dst.data_ptr(), B, stream)
    else:
        plan.polymul(src.data_ptr(), (src2 if src2 is not None else src).data_ptr(), dst.data_ptr(), scratch.data_ptr(), B, stream)


plan.set_variant(agx.VARIANT_LDS_RADIX2)
run(ref_in, ref, ref_in2)
torch.cuda.synchronize()
out = torch.empty_like(ref_in)
ok = {}
def select(k):
    plan.set_variant(agx.VARIANT_AUTO if k == -2 else agx.VARIANT_LDS_RADIX2 if k < 0 else agx.VARIANT_REGBLOCK_BASE + k)      # -2: the plan's tuned default


for k in ids:
    select(k)
    out.zero_()
    run(ref_in, out, ref_in2)
    torch.cuda.synchronize()
    ok[k] = bool(torch.equal(out, ref))
times = {k: [] for k in ids}
for rnd in range(5):
    for k in ids:
        select(k)
        for i in range(3):
            run(slabs[i % SLABS], slabs[(i + 1) % SLABS] if args.oop else slabs[i % SLABS], slabs[(i + 1) % SLABS])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(args.launches):
            run(slabs[i % SLABS], slabs[(i + 1) % SLABS] if args.oop else slabs[i % SLABS], slabs[(i + 1) % SLABS])
        e1.record()
        torch.cuda.synchronize()
        times[k].append(e0.elapsed_time(e1) / args.launches)
print(f"{'id':>3} {'bit-exact':>9} {'min ms':>8} {'med ms':>8} {'MNTT/s':>8} {'GB/s':>8} {'%8TB/s':>7}")
for k in ids:
    t = sorted(times[k])
    mn, md = t[0], t[len(t) // 2]
    rate = P * B / (md * 1e-3)
    bpu = (24 if args.op == "mul" else 16) * N     # algorithmic bytes per unit
    print(f"{k:>3} {str(ok[k]):>9} {mn:8.4f} {md:8.4f} {rate / 1e6:8.2f} {rate * bpu / 1e9:8.1f} {rate * bpu / 8e12 * 100:7.2f}")
