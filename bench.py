#!/usr/bin/env python3
"""bench.py -- headline benchmark: batched n=4096 forward NTTs/s on N MI355X (BASELINE.json).

A "step" is one forward pass of the hot path over one batch resident in HBM: per GPU,
n=4096 x 4 RNS primes (the four largest 60-bit primes = 1 mod 8192) x batch 4096 polynomials
= 16,384 NTTs = 512 MiB read + 512 MiB written, in place, one kernel launch
(BASELINE.json configs[2]).  Steps rotate over 4 distinct slabs
(2 GiB) so the 256 MiB Infinity Cache cannot serve the reads.  Frames are
independent, so N GPUs run N shards with no collective on the data path (weak scaling:
fixed work per GPU); the only communication is the barrier / max-reduce of the timing.

Usage:  python bench.py [--gpus N] [--steps K] [--warmup W]
  N > 1 is launched by the driver as
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_COEFF = 4096
NUM_PRIMES = 4
PRIME_BITS = 60
BATCH_PER_GPU = 4096
NUM_SLABS = 4
RAMP_SECONDS = 0.5                        # set-up work that brings the GPU clock out of idle (see main)
HBM_PEAK_GBS = 8000.0                     # MI355X_MICROARCH.md: 8.0 TB/s spec
ALGO_BYTES_PER_NTT = 16 * N_COEFF         # read 8n + write 8n


def host_cores():
    """Cores this process may really use: the affinity mask capped by the cgroup CPU quota."""
    cores = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except Exception:
        pass
    return cores


def cpu_baseline(target_seconds=12.0):
    """The CPU restatement of the reference butterfly (oracle, kind "port") timed on this
    box's host cores on a bounded sample of the same workload shape (n=4096, 60-bit q):
    repeated passes over a fixed 1024-frames-per-thread buffer until ~target_seconds."""
    from oracle import oracle

    cores = int(os.environ.get("AGX_BENCH_CPU_THREADS", "0")) or host_cores()
    q = oracle.find_prime(PRIME_BITS, N_COEFF)
    psi = oracle.min_root(q, N_COEFF)
    tw, pre = oracle.make_tables(q, psi, N_COEFF)
    frames = 1024 * cores
    x = oracle.fill_splitmix(frames * N_COEFF, 42, q)
    t0 = time.perf_counter()
    oracle.forward_mt(x, q, tw, pre, N_COEFF, cores)      # also the warm-up pass
    first = time.perf_counter() - t0
    passes = max(1, min(512, int(target_seconds / max(first, 1e-3))))
    t0 = time.perf_counter()
    for _ in range(passes):
        oracle.forward_mt(x, q, tw, pre, N_COEFF, cores)
    elapsed = time.perf_counter() - t0
    one = x[: N_COEFF * 1024]
    t0 = time.perf_counter()
    oracle.forward_mt(one, q, tw, pre, N_COEFF, 1)
    single = 1024 / (time.perf_counter() - t0)
    return {
        "value": frames * passes / elapsed, "unit": "NTT/s", "cores": cores, "kind": "port",
        "single_core_value": single,
        "sample": f"{passes} passes over {frames} frames of n={N_COEFF}, one {PRIME_BITS}-bit modulus, "
                  f"oracle/ntt_oracle.c (restatement of the reference butterfly) on {cores} pthreads, {elapsed:.1f} s",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=BATCH_PER_GPU, help="polynomials per GPU per step")
    ap.add_argument("--ramp-seconds", type=float, default=RAMP_SECONDS,
                    help="set-up: run the step this long before the W warm-up steps so the GPU clock has ramped (0 = cold start)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch

    import agilex_ntt_amd as agx

    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if world_env != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world_env}: launch N>1 through torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    grp = agx.Group(backend="nccl", device=torch.device("cuda", local_rank))   # "nccl" is RCCL on ROCm
    rank, world = grp.rank, grp.world

    batch = args.batch
    qs = agx.find_primes(PRIME_BITS, N_COEFF, NUM_PRIMES)
    plan = agx.Plan(N_COEFF, qs)
    per_slab = NUM_PRIMES * batch * N_COEFF
    stream = torch.cuda.current_stream().cuda_stream
    # this rank's shard of the global batch: polynomials [rank*batch, (rank+1)*batch) of every slab
    slabs = [torch.empty(per_slab, dtype=torch.int64, device="cuda") for _ in range(NUM_SLABS)]
    for i, s in enumerate(slabs):
        plan.fill_synthetic(s.data_ptr(), batch, first_poly=(i * world + rank) * batch, seed=42, stream=stream)

    # correctness guard on real data before timing: INTT(NTT(x)) == x on slab 0 (product path only)
    check = slabs[0].clone()
    plan.forward(check.data_ptr(), check.data_ptr(), batch, stream)
    plan.inverse(check.data_ptr(), check.data_ptr(), batch, stream)
    torch.cuda.synchronize()
    if not torch.equal(check, slabs[0]):
        raise SystemExit("round trip failed: refusing to report a number")
    del check

    def step(i):
        s = slabs[i % NUM_SLABS]
        plan.forward(s.data_ptr(), s.data_ptr(), batch, stream)

    # Steady-state clocks.  After the host-side set-up (prime search, tables, RCCL bring-up) the GPU
    # is idle, and its clock takes ~100 ms of work to ramp: the first 100 launches average 0.33 ms,
    # every later one 0.30 ms (profiles/r01f_clock_ramp.txt).  A service runs in the ramped state, so
    # set-up ends with RAMP_SECONDS of the same step; the W warm-up steps and the K timed steps
    # follow unchanged.  The first barrier here also pays RCCL's communicator set-up outside the timing.
    grp.barrier()
    torch.cuda.synchronize()
    ramp_steps = 0
    t_end = time.perf_counter() + args.ramp_seconds
    while time.perf_counter() < t_end:
        for _ in range(64):
            step(ramp_steps)
            ramp_steps += 1
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    grp.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()            # same stream the kernels are launched on
    for i in range(args.steps):
        step(args.warmup + i)
    ev1.record()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0     # this rank's K steps; the slowest rank defines the job (MAX below)
    grp.barrier()
    torch.cuda.synchronize()
    kernel_ms = ev0.elapsed_time(ev1) / args.steps   # average launch duration over the timed region
    elapsed = grp.max_over_ranks(elapsed)

    ntts_per_step_per_gpu = NUM_PRIMES * batch
    value = agx.aggregate_throughput(ntts_per_step_per_gpu, args.steps, world, elapsed)
    achieved = ntts_per_step_per_gpu * ALGO_BYTES_PER_NTT / (kernel_ms * 1e-3) / 1e9
    traffic = None
    tfile = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(tfile):
        try:
            traffic = json.load(open(tfile)).get("bytes_per_launch")
        except Exception:
            traffic = None
    out = {
        "metric": "batched n=4096 forward NTTs/sec at 1/2/4/8 MI355X; %HBM roofline",
        "value": value, "unit": "NTT/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {
            "workload": f"n={N_COEFF}, {NUM_PRIMES}-prime RNS ({PRIME_BITS}-bit), batch={batch} polynomials per GPU, "
                        "forward negacyclic NTT in place (BASELINE.json configs[2])",
            "n": N_COEFF, "primes": NUM_PRIMES, "batch_per_gpu": batch, "ntts_per_step_per_gpu": ntts_per_step_per_gpu,
            "slabs_rotated": NUM_SLABS, "clock_ramp_steps_before_warmup": ramp_steps, "parallelism": f"batch-sharded x{world}, no collective",
            "polys_per_sec": value / NUM_PRIMES,
        },
        "roofline": {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic, "kernel_ms": kernel_ms,
            "algorithmic_bytes_per_launch": ntts_per_step_per_gpu * ALGO_BYTES_PER_NTT,
        },
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline()
    if rank == 0:
        print(json.dumps(out), flush=True)
    plan.close()
    grp.close()


if __name__ == "__main__":
    main()
