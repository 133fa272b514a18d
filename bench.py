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


class PowerSampler:
    """Socket power and shader clock of the bench GPU from amdgpu's sysfs files (hwmon power1_average/_input in microwatts,
    pp_dpm_sclk's starred level), sampled every 5 ms by a daemon thread while the steps run (the shortest secondary line is ~50 ms: >= 10 samples).  Evidence only: whether the
    part sits at its power cap on this instruction mix (profiles/r02a_clock_power.txt); never part of `value`."""

    def __init__(self, torch, index):
        import glob
        import threading

        self.samples = []
        self._stop = threading.Event()
        self._thread = None
        self.power_file = self.sclk_file = self.cap_file = None
        try:
            props = torch.cuda.get_device_properties(index)
            bdf = f"{props.pci_domain_id:04x}:{props.pci_bus_id:02x}:{props.pci_device_id:02x}"
            for dev in glob.glob("/sys/class/drm/card*/device"):
                if os.path.basename(os.path.realpath(dev)).startswith(bdf):
                    for name in ("power1_average", "power1_input"):
                        hit = glob.glob(os.path.join(dev, "hwmon", "hwmon*", name))
                        if hit and self.power_file is None:
                            self.power_file = hit[0]
                    cap = glob.glob(os.path.join(dev, "hwmon", "hwmon*", "power1_cap"))
                    self.cap_file = cap[0] if cap else None
                    self.sclk_file = os.path.join(dev, "pp_dpm_sclk")
        except Exception:
            pass
        if self.power_file or self.sclk_file:
            self._thread = threading.Thread(target=self._run, daemon=True)
            self._thread.start()

    def _read(self):
        w = mhz = None
        try:
            if self.power_file:
                w = int(open(self.power_file).read()) / 1e6
        except Exception:
            pass
        try:
            for line in open(self.sclk_file):
                if "*" in line:
                    mhz = int(line.split(":")[1].strip().split("Mhz")[0])
        except Exception:
            pass
        return w, mhz

    def _run(self):
        while not self._stop.is_set():
            self.samples.append((time.perf_counter(),) + self._read())
            self._stop.wait(0.005)

    def window(self, t0, t1):
        """samples taken in [t0, t1] (perf_counter times) -> summary dict or None"""
        pick = [s for s in self.samples if t0 <= s[0] <= t1]
        watts = sorted(s[1] for s in pick if s[1] is not None)
        mhz = sorted(s[2] for s in pick if s[2] is not None)
        if not watts and not mhz:
            return None
        cap = None
        try:
            cap = int(open(self.cap_file).read()) / 1e6 if self.cap_file else None
        except Exception:
            pass
        return {"samples": len(pick), "socket_power_w_median": watts[len(watts) // 2] if watts else None,
                "socket_power_w_max": watts[-1] if watts else None, "power_cap_w": cap,
                "sclk_mhz_median": mhz[len(mhz) // 2] if mhz else None, "sclk_mhz_max_level": 2400,
                "source": "amdgpu sysfs (hwmon power1_*, pp_dpm_sclk), 5 ms period, while the ramp + warm-up + timed steps ran"}

    def median_w(self, t0, t1):
        watts = sorted(s[1] for s in self.samples if t0 <= s[0] <= t1 and s[1] is not None)
        return watts[len(watts) // 2] if watts else None

    def stop(self):
        self._stop.set()
        if self._thread:
            self._thread.join(timeout=1.0)


def _timed(torch, fn, launches, warmup, min_window_ms=60.0):
    """average duration (ms) of `launches` back-to-back calls of fn(i) after `warmup` untimed ones, measured with events on the
    launch stream; a line whose timed region would be shorter than min_window_ms is timed again with proportionally more launches, so
    that the power sampler (5 ms period) sees at least ~10 samples of it"""
    for i in range(warmup):
        fn(i)
    torch.cuda.synchronize()
    while True:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        _timed.window = [time.perf_counter(), None]
        e0.record()
        for i in range(launches):
            fn(warmup + i)
        e1.record()
        torch.cuda.synchronize()
        _timed.window[1] = time.perf_counter()
        total = e0.elapsed_time(e1)
        _timed.launches = launches
        if total >= min_window_ms or launches >= 20000:
            return total / launches
        launches = int(launches * min_window_ms / max(total, 1e-3) * 1.1) + 1


PROFILE_TAG = "r04f"      # the committed per-operation counter summaries the secondary lines point at


def per_slab_elems(slabs):
    return slabs[0].numel()


def secondary_lines(torch, agx, plan4096, slabs, batch, stream, sampler=None, idle_w=None):
    """The other north_star paths on the same clock, same process, after the headline:
    n=4096 inverse, n=4096 fused poly-mul, BASELINE configs[3] per-GPU slice (n=16384, 8 primes, batch 8192,
    in place) and configs[4] per-GPU slice (n=32768 poly-mul, batch 1024).  Each entry: units/s, average
    launch time (HIP events), algorithmic bytes per launch and the fraction of the 8 TB/s HBM roofline."""
    out = []

    def entry(name, workload, units, bytes_per_unit, ms, unit, launches, note=None, profile=None):
        gbs = units * bytes_per_unit / (ms * 1e-3) / 1e9
        e = {"name": name, "workload": workload, "value": units / (ms * 1e-3), "unit": unit, "kernel_ms": ms,
             "launches_timed": getattr(_timed, "launches", launches), "algorithmic_bytes_per_launch": units * bytes_per_unit,
             "achieved_GBps": gbs, "frac": gbs / HBM_PEAK_GBS}
        if note:
            e["note"] = note
        if profile:      # rocprofv3 counters of this line's dominant kernel(s): tools/profile_ops.sh -> tools/summarize_ops.py
            e["profile"] = f"profiles/{PROFILE_TAG}_{profile}_summary.md"
        if sampler is not None:
            w = sampler.window(*_timed.window)
            if w:
                e["power"] = {k: w[k] for k in ("samples", "socket_power_w_median", "sclk_mhz_median")}
                if idle_w is not None and w["socket_power_w_median"] is not None:
                    e["power"]["energy_uj_per_unit"] = (w["socket_power_w_median"] - idle_w) / (units / (ms * 1e-3)) * 1e6
        out.append(e)

    ns = len(slabs)
    units = NUM_PRIMES * batch
    # (1) n=4096 inverse, same slabs, in place (values are whatever the forward steps left: any input below 4q is legal)
    ms = _timed(torch, lambda i: plan4096.inverse(slabs[i % ns].data_ptr(), slabs[i % ns].data_ptr(), batch, stream), 300, 8)
    entry("inverse_n4096", f"n={N_COEFF}, {NUM_PRIMES} primes, batch {batch}, inverse NTT in place", units, 16 * N_COEFF, ms, "NTT/s", 300, profile="inv4096")
    # (2) n=4096 fused polynomial product c = INTT(NTT(a) o NTT(b)), c aliasing a
    ms = _timed(torch, lambda i: plan4096.polymul(slabs[i % ns].data_ptr(), slabs[(i + 1) % ns].data_ptr(), slabs[i % ns].data_ptr(), 0, batch, stream), 120, 4)
    entry("polymul_n4096", f"n={N_COEFF}, {NUM_PRIMES} primes, batch {batch}, fused NTT x2 -> pointwise -> INTT in one launch (24n bytes per product)",
          units, 24 * N_COEFF, ms, "products/s", 120,
          note="three transforms per 24n bytes: bounded by VALU integer multiply issue, not HBM (DESIGN.md section 4)", profile="mul4096")

    # narrow moduli (the reference's own modulus class: src/main.cpp:55 is 65537, BASELINE configs[0] a 30-bit prime): 32-bit arithmetic
    # kernels, same uint64 data at the ABI, same 16n bytes per NTT -- the configuration where the engine is HBM-bound, not power-bound
    for nn, tag in ((1024, "fwd1024q30"), (4096, "fwd4096q30")):
        pn = agx.Plan(nn, agx.find_primes(30, nn, NUM_PRIMES))
        per = NUM_PRIMES * batch * nn
        views = [sl[:per] for sl in slabs]
        for k, v in enumerate(views):
            pn.fill_synthetic(v.data_ptr(), batch, k * batch, 42, stream)
        ms = _timed(torch, lambda i: pn.forward(views[i % ns].data_ptr(), views[i % ns].data_ptr(), batch, stream), 300, 8)
        entry(f"forward_n{nn}_30bit", f"n={nn}, {NUM_PRIMES} primes of 30 bits, batch {batch}, forward NTT in place, 32-bit arithmetic kernels (csrc/rb32_kernels.hpp)",
              units, 16 * nn, ms, "NTT/s", 300, profile=tag)
        if nn == 4096:
            ms = _timed(torch, lambda i: pn.inverse(views[i % ns].data_ptr(), views[i % ns].data_ptr(), batch, stream), 300, 8)
            entry("inverse_n4096_30bit", f"n={nn}, {NUM_PRIMES} primes of 30 bits, batch {batch}, inverse NTT in place, 32-bit arithmetic", units, 16 * nn, ms, "NTT/s", 300, profile="inv4096q30")
            ms = _timed(torch, lambda i: pn.polymul(views[i % ns].data_ptr(), views[(i + 1) % ns].data_ptr(), views[i % ns].data_ptr(), 0, batch, stream), 200, 4)
            entry("polymul_n4096_30bit", f"n={nn}, {NUM_PRIMES} primes of 30 bits, batch {batch}, fused product in one launch, 32-bit arithmetic", units, 24 * nn, ms, "products/s", 200, profile="mul4096q30")
        pn.close()
    # the small sizes (n = 32 is the smallest size of the reference's own table, include/kernel/ntt.h:11-12): wave-packed kernels
    # (csrc/wp_kernels.hpp), >= 1 GiB per launch, the same four slabs rotated; 60-bit primes, and 30-bit ones on the 32-bit kernels
    for nn, bits, tag in ((32, 60, "fwd32"), (256, 60, "fwd256"), (512, 60, "fwd512"), (32, 30, "fwd32q30"), (512, 30, "fwd512q30")):
        bsm = per_slab_elems(slabs) // (NUM_PRIMES * nn)
        pn = agx.Plan(nn, agx.find_primes(bits, nn, NUM_PRIMES))
        for k, v in enumerate(slabs):
            pn.fill_synthetic(v.data_ptr(), bsm, k * bsm, 42, stream)
        ms = _timed(torch, lambda i: pn.forward(slabs[i % ns].data_ptr(), slabs[i % ns].data_ptr(), bsm, stream), 200, 8)
        entry(f"forward_n{nn}" + ("" if bits == 60 else f"_{bits}bit"), f"n={nn}, {NUM_PRIMES} primes of {bits} bits, batch {bsm}, forward NTT in place, "
              f"{'32-bit' if bits < 32 else '64-bit'} wave-packed kernels (several frames per wave, no workgroup barrier)", NUM_PRIMES * bsm, 16 * nn, ms, "NTT/s", 200, profile=tag)
        if (nn, bits) in ((32, 60), (512, 60)):
            ms = _timed(torch, lambda i: pn.inverse(slabs[i % ns].data_ptr(), slabs[i % ns].data_ptr(), bsm, stream), 200, 8)
            entry(f"inverse_n{nn}", f"n={nn}, {NUM_PRIMES} primes of {bits} bits, batch {bsm}, inverse NTT in place", NUM_PRIMES * bsm, 16 * nn, ms, "NTT/s", 200, profile=tag.replace("fwd", "inv"))
            ms = _timed(torch, lambda i: pn.polymul(slabs[i % ns].data_ptr(), slabs[(i + 1) % ns].data_ptr(), slabs[i % ns].data_ptr(), 0, bsm, stream), 100, 4)
            entry(f"polymul_n{nn}", f"n={nn}, {NUM_PRIMES} primes of {bits} bits, batch {bsm}, fused product in one launch (both transforms in registers)", NUM_PRIMES * bsm, 24 * nn, ms, "products/s", 100, profile=tag.replace("fwd", "mul"))
        pn.close()
    for s in slabs:
        s.untyped_storage().resize_(0)      # give the 2 GiB back before the large slices
    torch.cuda.empty_cache()

    # (3) BASELINE configs[3] per-GPU slice: n=16384, 8 primes, batch 65536/8 = 8192 -> 65,536 NTTs, 8 GiB in place
    n3, p3, b3 = 16384, 8, 8192
    plan3 = agx.Plan(n3, agx.find_primes(PRIME_BITS, n3, p3))
    buf = torch.empty(p3 * b3 * n3, dtype=torch.int64, device="cuda")
    plan3.fill_synthetic(buf.data_ptr(), b3, 0, 42, stream)
    ms = _timed(torch, lambda i: plan3.forward(buf.data_ptr(), buf.data_ptr(), b3, stream), 16, 2)
    entry("forward_n16384_config4_slice", f"n={n3}, {p3} primes, batch {b3} per GPU (BASELINE.json configs[3] / 8 GPUs), forward in place, 8 GiB",
          p3 * b3, 16 * n3, ms, "NTT/s", 16, profile="fwd16384")
    ms = _timed(torch, lambda i: plan3.inverse(buf.data_ptr(), buf.data_ptr(), b3, stream), 16, 2)
    entry("inverse_n16384_config4_slice", f"n={n3}, {p3} primes, batch {b3} per GPU, inverse in place, 8 GiB", p3 * b3, 16 * n3, ms, "NTT/s", 16, profile="inv16384")
    plan3.close()
    del buf
    torch.cuda.empty_cache()

    # (4) BASELINE configs[4] per-GPU slice: n=32768 poly-mul, one 60-bit prime, batch 8192/8 = 1024;
    # three (a, b) operand sets rotate (3 x 512 MiB) so the 256 MiB Infinity Cache cannot hold them
    n4, b4, sets = 32768, 1024, 3
    plan4 = agx.Plan(n4, agx.find_primes(PRIME_BITS, n4, 1))
    ab = [[torch.empty(b4 * n4, dtype=torch.int64, device="cuda") for _ in range(2)] for _ in range(sets)]
    for k, (a, b) in enumerate(ab):
        plan4.fill_synthetic(a.data_ptr(), b4, 2 * k * b4, 42, stream)
        plan4.fill_synthetic(b.data_ptr(), b4, (2 * k + 1) * b4, 42, stream)
    c = torch.empty(b4 * n4, dtype=torch.int64, device="cuda")
    scratch = torch.empty(b4 * n4, dtype=torch.int64, device="cuda")
    ms = _timed(torch, lambda i: plan4.polymul(ab[i % sets][0].data_ptr(), ab[i % sets][1].data_ptr(), c.data_ptr(), scratch.data_ptr(), b4, stream), 120, 3)
    entry("polymul_n32768_config5_slice", f"n={n4}, one {PRIME_BITS}-bit prime, batch {b4} per GPU (BASELINE.json configs[4] / 8 GPUs), "
          "c = INTT(NTT(a) o NTT(b)) in one launch (one frame in registers, NTT(a) parked in c's frame), operands never modified", b4, 24 * n4, ms, "products/s", 120,
          note="VALU-bound like polymul_n4096", profile="mul32768")
    ms = _timed(torch, lambda i: plan4.forward(ab[i % sets][0].data_ptr(), c.data_ptr(), b4, stream), 300, 3)
    entry("forward_n32768", f"n={n4}, one prime, batch {b4}, forward out of place", b4, 16 * n4, ms, "NTT/s", 300, profile="fwd32768oop")
    ms = _timed(torch, lambda i: plan4.inverse(ab[i % sets][0].data_ptr(), c.data_ptr(), b4, stream), 300, 3)
    entry("inverse_n32768", f"n={n4}, one prime, batch {b4}, inverse out of place", b4, 16 * n4, ms, "NTT/s", 300, profile="inv32768")
    plan4.close()
    return out


def bound_note(power, value_per_gpu):
    """how far the headline kernel is from its own power bound, from this run's samples: at the board's cap throughput is
    (cap - idle) / (energy per NTT); VERDICT r03: the 64-bit butterfly sits at that bound, not at an instruction-scheduling one"""
    if not power or power.get("energy_uj_per_ntt") is None or power.get("power_cap_w") is None or power.get("idle_w") is None:
        return None
    e, cap, idle = power["energy_uj_per_ntt"], power["power_cap_w"], power["idle_w"]
    at_cap = (cap - idle) / (e * 1e-6)
    return {"bound": "board power cap", "energy_uj_per_ntt": e, "power_cap_w": cap, "idle_w": idle, "measured_socket_w_median": power.get("socket_power_w_median"),
            "value_at_cap": at_cap, "value_over_value_at_cap": value_per_gpu / at_cap,
            "roofline_target_value": 0.60 * HBM_PEAK_GBS * 1e9 / ALGO_BYTES_PER_NTT, "socket_w_the_target_would_draw": idle + 0.60 * HBM_PEAK_GBS * 1e9 / ALGO_BYTES_PER_NTT * e * 1e-6,
            "note": "value_at_cap = (power_cap_w - idle_w) / energy_uj_per_ntt: NTT/s per GPU this kernel reaches when the board draws its cap; "
                    "value_over_value_at_cap ~ 1 means the kernel is power-bound (DESIGN.md 3.5)"}


def main_group(args, torch, agx):
    """--single-process: the same step on N devices through agx_ntt_group_* (include/agx_ntt.h section 5).  One process, no
    torch.distributed: the group's host threads launch every shard on its own stream; the timed region is bracketed by
    agx_ntt_group_synchronize (the barrier) and per-device HIP events on the shards' streams give per_rank."""
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    devices = [int(d) for d in args.devices.split(",")] if args.devices else list(range(args.gpus))
    world = len(devices)
    batch = args.batch
    qs = agx.find_primes(PRIME_BITS, N_COEFF, NUM_PRIMES)
    grp = agx.DeviceGroup(devices, N_COEFF, qs)
    per_slab = NUM_PRIMES * batch * N_COEFF
    L = agx.lib()
    slabs, ext = [], []
    for i, d in enumerate(devices):
        torch.cuda.set_device(d)
        _, plan_h, stream_h = grp.shard(i)
        ext.append(torch.cuda.ExternalStream(stream_h, device=torch.device("cuda", d)))
        mine = [torch.empty(per_slab, dtype=torch.int64, device=f"cuda:{d}") for _ in range(NUM_SLABS)]
        for k, sl in enumerate(mine):
            rc = L.agx_ntt_fill_synthetic(plan_h, sl.data_ptr(), batch, (k * world + i) * batch, 42, stream_h)
            if rc:
                raise SystemExit(f"fill_synthetic failed on shard {i}: {rc}")
        slabs.append(mine)
    grp.synchronize()
    # correctness guard before timing: INTT(NTT(x)) == x on slab 0 of every shard, through the group calls
    checks = []
    for i, d in enumerate(devices):
        torch.cuda.set_device(d)
        checks.append(slabs[i][0].clone())
    torch.cuda.synchronize()
    grp.forward([c.data_ptr() for c in checks], [c.data_ptr() for c in checks], [batch] * world)
    grp.inverse([c.data_ptr() for c in checks], [c.data_ptr() for c in checks], [batch] * world)
    grp.synchronize()
    for i in range(world):
        if not torch.equal(checks[i], slabs[i][0]):
            raise SystemExit(f"round trip failed on shard {i}: refusing to report a number")
    del checks
    batches = [batch] * world

    def step(it):
        ptrs = [slabs[i][it % NUM_SLABS].data_ptr() for i in range(world)]
        grp.forward(ptrs, ptrs, batches)

    sampler = PowerSampler(torch, devices[0])
    t_idle0 = time.perf_counter()
    time.sleep(0.4)
    idle_w = sampler.median_w(t_idle0, time.perf_counter())
    t_busy0 = time.perf_counter() + 0.2
    ramp_steps = 0
    t_end = time.perf_counter() + args.ramp_seconds
    while time.perf_counter() < t_end:
        for _ in range(64):
            step(ramp_steps)
            ramp_steps += 1
        grp.synchronize()
    for i in range(args.warmup):
        step(i)
    grp.synchronize()      # the barrier: every shard idle
    ev = []
    for i, d in enumerate(devices):
        torch.cuda.set_device(d)
        ev.append((torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)))
    t0 = time.perf_counter()
    for i in range(world):
        ev[i][0].record(ext[i])
    for it in range(args.steps):
        step(args.warmup + it)
    for i in range(world):
        ev[i][1].record(ext[i])
    grp.synchronize()
    elapsed = time.perf_counter() - t0      # all shards: the slowest one defines it
    t_busy1 = time.perf_counter()
    power = sampler.window(t_busy0, t_busy1)
    kernel_ms = [ev[i][0].elapsed_time(ev[i][1]) / args.steps for i in range(world)]
    per_rank = [{"rank": i, "device": devices[i], "elapsed_s": kernel_ms[i] * args.steps * 1e-3, "kernel_ms": kernel_ms[i],
                 "socket_power_w_median": power["socket_power_w_median"] if (power and devices[i] == devices[0]) else None,
                 "sclk_mhz_median": power["sclk_mhz_median"] if (power and devices[i] == devices[0]) else None} for i in range(world)]
    ntts = NUM_PRIMES * batch
    value = agx.aggregate_throughput(ntts, args.steps, world, elapsed)
    achieved = value / world * ALGO_BYTES_PER_NTT / 1e9
    slow_ms = max(kernel_ms)
    achieved_events = ntts * ALGO_BYTES_PER_NTT / (slow_ms * 1e-3) / 1e9
    if power and idle_w is not None and power["socket_power_w_median"] is not None and len(set(devices)) == world:
        power["idle_w"] = idle_w
        power["energy_uj_per_ntt"] = (power["socket_power_w_median"] - idle_w) / (value / world) * 1e6
    out = {
        "metric": "batched n=4096 forward NTTs/sec at 1/2/4/8 MI355X; %HBM roofline",
        "value": value, "unit": "NTT/s", "n_gpus": len(set(devices)), "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {
            "workload": f"n={N_COEFF}, {NUM_PRIMES}-prime RNS ({PRIME_BITS}-bit), batch={batch} polynomials per shard, "
                        "forward negacyclic NTT in place (BASELINE.json configs[2])",
            "n": N_COEFF, "primes": NUM_PRIMES, "batch_per_gpu": batch, "ntts_per_step_per_gpu": ntts, "slabs_rotated": NUM_SLABS,
            "clock_ramp_steps_before_warmup": ramp_steps, "parallelism": f"batch-sharded x{world}, no collective",
            "launch": "single process: agx_ntt_group_* (one shard, host thread and stream per device; no torch.distributed)", "devices": devices,
            "polys_per_sec": value / NUM_PRIMES,
        },
        "roofline": {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": None, "kernel_ms": slow_ms, "achieved_from_kernel_events": achieved_events,
            "frac_from_kernel_events": achieved_events / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": ntts * ALGO_BYTES_PER_NTT,
        },
        "per_rank": per_rank,
    }
    note = bound_note(power, value / world)
    if note:
        out["roofline"]["bound_note"] = note
    if power:
        out["power"] = power
    sampler.stop()
    print(json.dumps(out), flush=True)
    grp.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=BATCH_PER_GPU, help="polynomials per GPU per step")
    ap.add_argument("--ramp-seconds", type=float, default=RAMP_SECONDS,
                    help="set-up: run the step this long before the W warm-up steps so the GPU clock has ramped (0 = cold start)")
    ap.add_argument("--single-process", action="store_true",
                    help="drive the N GPUs from ONE process through the library's own multi-GPU driver (agx_ntt_group_*: one shard, host thread and "
                         "stream per device, no torch.distributed); same JSON line, per_rank from per-device HIP events")
    ap.add_argument("--devices", type=str, default=None, help="with --single-process: explicit device list, e.g. 0,0 (two shards on one GPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary lines (inverse, poly-mul, n=16384, n=32768)")
    args = ap.parse_args()

    import torch

    import agilex_ntt_amd as agx

    if args.single_process:
        return main_group(args, torch, agx)
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if world_env != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world_env}: launch N>1 through torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # AGX_BENCH_REHEARSAL=1: every rank on GPU 0 and the timing collectives over gloo (RCCL refuses two ranks on one GPU) -- the N > 1 control flow
    # of this file (shard offsets, barriers, MAX over ranks, per_rank gather, one JSON line from rank 0) rehearsed on a one-GPU box.  The line says so
    # ("rehearsal": true); its `value` is two ranks SHARING one GPU and is not a multi-GPU figure.
    rehearsal = os.environ.get("AGX_BENCH_REHEARSAL") == "1"
    dev_index = 0 if rehearsal else local_rank
    torch.cuda.set_device(dev_index)
    grp = agx.Group(backend="gloo", device=None) if rehearsal else agx.Group(backend="nccl", device=torch.device("cuda", dev_index))   # "nccl" is RCCL on ROCm
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
    sampler = PowerSampler(torch, dev_index)       # every rank samples its own GPU (per_rank below); rank 0's goes into `power`
    # idle socket power: 0.4 s with nothing queued, before the ramp (the baseline of energy_uj_per_ntt)
    t_idle0 = time.perf_counter()
    time.sleep(0.4)
    idle_w = sampler.median_w(t_idle0, time.perf_counter())
    t_busy0 = time.perf_counter() + 0.2     # skip the first 200 ms (idle -> ramp)
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
    t_busy1 = time.perf_counter()
    power = sampler.window(t_busy0, t_busy1)
    grp.barrier()
    torch.cuda.synchronize()
    kernel_ms = ev0.elapsed_time(ev1) / args.steps   # average launch duration over the timed region
    own_elapsed = elapsed
    elapsed = grp.max_over_ranks(elapsed)
    # every rank's own figures in rank order, so a slow or throttled rank is visible beside the MAX that defines `value`
    rows = grp.gather_rows([own_elapsed, kernel_ms, power["socket_power_w_median"] if power else None, power["sclk_mhz_median"] if power else None])
    nan2none = lambda v: None if v != v else v
    per_rank = [{"rank": r, "elapsed_s": row[0], "kernel_ms": row[1], "socket_power_w_median": nan2none(row[2]), "sclk_mhz_median": nan2none(row[3])}
                for r, row in enumerate(rows)]

    ntts_per_step_per_gpu = NUM_PRIMES * batch
    value = agx.aggregate_throughput(ntts_per_step_per_gpu, args.steps, world, elapsed)
    # achieved = NTT/s x 16n per GPU -- follows from `value` (wall clock around the K steps, slowest rank); the HIP-event
    # figure of this rank's launches is kept beside it as kernel_ms / achieved_from_kernel_events
    achieved = value / world * ALGO_BYTES_PER_NTT / 1e9
    achieved_events = ntts_per_step_per_gpu * ALGO_BYTES_PER_NTT / (kernel_ms * 1e-3) / 1e9
    if power and idle_w is not None and power["socket_power_w_median"] is not None:
        power["idle_w"] = idle_w
        power["energy_uj_per_ntt"] = (power["socket_power_w_median"] - idle_w) / (value / world) * 1e6      # the line to optimise at the power cap
    # HBM bytes per launch from the PMC passes of tools/profile.sh (FETCH_SIZE x2 + WRITE_SIZE, separate runs,
    # MI355X_MICROARCH.md): a figure of the build it was profiled on, so it is only reported while the kernel
    # sources still hash to what that profile recorded; otherwise null (re-run tools/profile.sh)
    traffic, traffic_source = None, None
    tfile = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(tfile):
        try:
            tj = json.load(open(tfile))
            if tj.get("kernel_source_sha16") == agx.kernel_source_sha16():
                traffic = tj.get("bytes_per_launch")
                traffic_source = {"profile": tj.get("tag"), "kernel_source_sha16": tj.get("kernel_source_sha16")}
            else:
                traffic_source = {"profile": tj.get("tag"), "stale": True}
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
            **({"rehearsal": True, "rehearsal_note": f"{world} ranks on ONE GPU, timing collectives over gloo: exercises the N > 1 control flow, not a multi-GPU figure"} if rehearsal else {}),
        },
        "roofline": {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic, "traffic_source": traffic_source, "kernel_ms": kernel_ms,
            "achieved_from_kernel_events": achieved_events, "frac_from_kernel_events": achieved_events / HBM_PEAK_GBS,
            "algorithmic_bytes_per_launch": ntts_per_step_per_gpu * ALGO_BYTES_PER_NTT,
        },
        "per_rank": per_rank,
    }
    note = bound_note(power, value / world)
    if note:
        out["roofline"]["bound_note"] = note      # the three numbers the "power-bound" argument rests on, measured in this run
    if power and rank == 0:
        out["power"] = power
    if world == 1 and not args.no_secondary:
        out["secondary"] = secondary_lines(torch, agx, plan, slabs, batch, stream, sampler, idle_w)
    sampler.stop()
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline()
    if rank == 0:
        print(json.dumps(out), flush=True)
    plan.close()
    grp.close()


if __name__ == "__main__":
    main()
