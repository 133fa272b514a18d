"""tools/run_op.py -- launch ONE operation of the engine a fixed number of times (for rocprofv3 --pmc / --kernel-trace
runs on paths bench.py's headline does not cover).
Usage: python3 tools/run_op.py --op {fwd,inv,mul} [--n N --primes P --batch B --bits 60 --oop --launches K --variant ID]
Under the profiler: rocprofv3 ... -- python3 tools/run_op.py ...   (the interpreter itself after `--`, never this file: an
`env` shebang hop after the profiler's preload has initialised the GPU is a forbidden exec on this pool)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import agilex_ntt_amd as agx  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--op", choices=["fwd", "inv", "mul"], default="inv")
ap.add_argument("--n", type=int, default=4096)
ap.add_argument("--primes", type=int, default=4)
ap.add_argument("--batch", type=int, default=4096)
ap.add_argument("--slabs", type=int, default=4)
ap.add_argument("--launches", type=int, default=20)
ap.add_argument("--bits", type=int, default=60, help="modulus size in bits")
ap.add_argument("--oop", action="store_true", help="forward / inverse out of place (slab i -> slab i+1)")
ap.add_argument("--variant", type=int, default=None, help="registry id (AGX_VARIANT_REGBLOCK_BASE + id)")
ap.add_argument("--ramp-seconds", type=float, default=0.5, help="run the operation this long before the warm-up launches so the GPU clock has ramped, as bench.py does (0 = cold)")
ap.add_argument("--mulsets", type=int, default=0, help="mul: K rotating (a, b) operand sets, c and scratch separate (bench.py's n = 32768 product line); 0 = c aliases a on the slabs")
ap.add_argument("--report", type=str, default=None, help="write {calls: ramp + warm-up + timed launches, ms: ...} here (tools/summarize_ops.py)")
args = ap.parse_args()
plan = agx.Plan(args.n, agx.find_primes(args.bits, args.n, args.primes))
if args.variant is not None:
    plan.set_variant(agx.VARIANT_REGBLOCK_BASE + args.variant)
stream = torch.cuda.current_stream().cuda_stream
per = args.primes * args.batch * args.n
slabs = [torch.empty(per, dtype=torch.int64, device="cuda") for _ in range(args.slabs)]
for i, s in enumerate(slabs):
    plan.fill_synthetic(s.data_ptr(), args.batch, i * args.batch, 42, stream)
scratch = torch.empty(per, dtype=torch.int64, device="cuda")
sets, cbuf = [], None
if args.op == "mul" and args.mulsets:
    sets = [[torch.empty(per, dtype=torch.int64, device="cuda") for _ in range(2)] for _ in range(args.mulsets)]
    for k, (a, b) in enumerate(sets):
        plan.fill_synthetic(a.data_ptr(), args.batch, 2 * k * args.batch, 42, stream)
        plan.fill_synthetic(b.data_ptr(), args.batch, (2 * k + 1) * args.batch, 42, stream)
    cbuf = torch.empty(per, dtype=torch.int64, device="cuda")


def run(i):
    if sets:
        a, b = sets[i % len(sets)]
        plan.polymul(a.data_ptr(), b.data_ptr(), cbuf.data_ptr(), scratch.data_ptr(), args.batch, stream)
        return
    a, b = slabs[i % args.slabs], slabs[(i + 1) % args.slabs]
    dst = b if args.oop else a
    if args.op == "fwd":
        plan.forward(a.data_ptr(), dst.data_ptr(), args.batch, stream)
    elif args.op == "inv":
        plan.inverse(a.data_ptr(), dst.data_ptr(), args.batch, stream)
    else:
        plan.polymul(a.data_ptr(), b.data_ptr(), a.data_ptr(), scratch.data_ptr(), args.batch, stream)


import time  # noqa: E402

calls = 0
t_end = time.perf_counter() + args.ramp_seconds
while time.perf_counter() < t_end:      # clock ramp (bench.py: the first ~100 ms of work after an idle period run ~10 % slower)
    for i in range(16):
        run(calls)
        calls += 1
    torch.cuda.synchronize()
for i in range(5):
    run(i)
calls += 5
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(args.launches):
    run(i)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / args.launches
calls += args.launches
if args.report:
    import json

    json.dump({"calls": calls, "timed": args.launches, "ms_per_launch": ms}, open(args.report, "w"))
print(f"{args.op} n={args.n} primes={args.primes} batch={args.batch}: {ms:.4f} ms per launch, {args.primes * args.batch / ms / 1e3:.2f} M units/s")
plan.close()
