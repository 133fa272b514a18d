"""tools/timeline.py -- where a wave of the default n=4096 kernel spends its life.

Runs the registry's trace twin of the default kernel (id 70 = the default id 90 + s_memtime stamps at 12 phase
boundaries, written per wave through agx_ntt_debug_set_trace_buffer) on the roofline workload and prints,
per phase, the mean / median / p90 duration and its share of the wave's life, plus the average number of
waves resident per SIMD.  The stamps cost a few scalar instructions and one forced wait for the stores, so
the traced launch is ~7 % slower than id 90; the shares are what matter.
The trace twin and the hook live in lib/libagxntt_diag.so (`make -C agilex-ntt_amd diag`):
Usage: AGX_NTT_LIB=agilex-ntt_amd/lib/libagxntt_diag.so python tools/timeline.py [--batch B] [--out FILE.json]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=4096)
ap.add_argument("--id", type=int, default=70)
ap.add_argument("--op", choices=["fwd", "inv"], default="fwd", help="which transform of the trace twin to stamp")
ap.add_argument("--ref", type=int, default=None, help="registry id of the untraced kernel timed beside the twin (default 93)")
ap.add_argument("--out", default=None)
ap.add_argument("--raw", default=None, help="also save the raw [wave][16] stamp array and the launch timings (.npz)")
ap.add_argument("--load", default=None, help="analyse a saved .npz instead of running on the GPU")
args = ap.parse_args()
if args.load:
    z = np.load(args.load)
    t, ms_default, ms_traced, ms_last = z["stamps"], float(z["ms_default"]), float(z["ms_traced"]), float(z["ms_last"])
    waves = t.shape[0]
else:
    import torch

    import agilex_ntt_amd as agx
    N, P, B, SLABS = 4096, 4, args.batch, 4
    qs = agx.find_primes(60, N, P)
    plan = agx.Plan(N, qs)
    stream = torch.cuda.current_stream().cuda_stream
    slabs = [torch.empty(P * B * N, dtype=torch.int64, device="cuda") for _ in range(SLABS)]
    for i, s in enumerate(slabs):
        plan.fill_synthetic(s.data_ptr(), B, i * B, 42, stream)
    waves = P * B * 8
    trace = torch.zeros(waves * 16, dtype=torch.int64, device="cuda")
    agx.debug_set_trace_buffer(trace.data_ptr(), trace.numel() * 8)

    def run(slab):
        (plan.inverse if args.op == "inv" else plan.forward)(slab.data_ptr(), slab.data_ptr(), B, stream)

    ref = args.ref if args.ref is not None else 93

    def launches(k, count):
        plan.set_variant(agx.VARIANT_REGBLOCK_BASE + k)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(count):
            run(slabs[i % SLABS])
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / count

    launches(ref, 600)                     # clocks up
    ms_default = launches(ref, 100)
    ms_traced = launches(args.id, 100)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    run(slabs[0])   # the launch whose stamps are kept
    e1.record()
    torch.cuda.synchronize()
    ms_last = e0.elapsed_time(e1)
    agx.debug_set_trace_buffer(0, 0)
    t = trace.cpu().numpy().astype(np.uint64).reshape(waves, 16)
    plan.close()
if args.raw:
    np.savez_compressed(args.raw, stamps=t, ms_default=ms_default, ms_traced=ms_traced, ms_last=ms_last)
ts = t[:, :12].astype(np.int64)
xcc = t[:, 13].astype(np.int64) & 0xF
# s_memtime counters are not synchronised across the chip: calibrate on the median per-CU span
# (a CU is busy from the launch's first workgroups to its last)
hw0 = t[:, 12].astype(np.int64)
cu_key = (xcc << 16) | ((hw0 >> 8) & 0xFF)
spans = np.array([int(ts[cu_key == c, 11].max() - ts[cu_key == c, 0].min()) for c in np.unique(cu_key)])
span = int(np.median(spans))
if os.environ.get("AGX_TIMELINE_DEBUG"):
    print("per-CU spans: CUs", len(spans), "min", spans.min(), "median", span, "max", spans.max())
if os.environ.get("AGX_TIMELINE_DEBUG"):
    for c in np.unique(t[:, 13]):
        m = t[:, 13] == c
        print("xcc reg", hex(int(c)), "waves", int(m.sum()), "span", int(ts[m, 11].max() - ts[m, 0].min()), "hw_id sample", hex(int(t[m, 12][0])))
    life_t = ts[:, 11] - ts[:, 0]
    print("life ticks: min", int(life_t.min()), "median", int(np.median(life_t)), "max", int(life_t.max()))
ns_per_tick = ms_last * 1e6 / span      # launch duration / stamp span of the launch whose stamps were kept
names = ["entry -> frame loaded (HBM read)", "pass 0 butterflies (stages 1-3)", "exchange 0 (LDS write+read)",
         "pass 1 butterflies", "exchange 1 (LDS + the s_barrier)", "pass 2 butterflies", "exchange 2",
         "pass 3 butterflies + final reduce", "stage-out exchange", "issue stores", "stores retire (forced wait)"]
if args.op == "inv":
    names = ["entry -> frame loaded (HBM read)", "staging through the image (LDS write+read)", "pass 3 butterflies (per-lane twiddles)",
             "exchange 3->2 (wave-local)", "pass 2 butterflies (per-lane twiddles)", "exchange 2->1 (wave-local)", "pass 1 butterflies",
             "exchange 1->0 (LDS + the s_barrier)", "pass 0 butterflies + n^-1 + final reduce", "issue stores", "stores retire (forced wait)"]
d = np.diff(ts, axis=1).astype(np.float64) * ns_per_tick
life = (ts[:, 11] - ts[:, 0]).astype(np.float64) * ns_per_tick
print(f"default kernel {ms_default:.4f} ms/launch, traced twin {ms_traced:.4f} ms/launch (single traced launch {ms_last:.4f} ms)")
print(f"stamp span {span} ticks -> {ns_per_tick:.3f} ns/tick; wave life mean {life.mean() / 1e3:.2f} us, median {np.median(life) / 1e3:.2f} us")
rows = []
print(f"{'phase':<40} {'mean us':>8} {'med us':>8} {'p90 us':>8} {'share':>7}")
for i, nm in enumerate(names):
    col = d[:, i]
    rows.append({"phase": nm, "mean_us": col.mean() / 1e3, "median_us": float(np.median(col)) / 1e3,
                 "p90_us": float(np.percentile(col, 90)) / 1e3, "share": col.mean() / life.mean()})
    print(f"{nm:<40} {rows[-1]['mean_us']:8.3f} {rows[-1]['median_us']:8.3f} {rows[-1]['p90_us']:8.3f} {rows[-1]['share'] * 100:6.1f}%")
compute = sum(r["share"] for r in rows if "butterflies" in r["phase"])
print(f"butterfly phases {compute * 100:.1f} % of a wave's life; everything else {100 - compute * 100:.1f} %")
hw = t[:, 12].astype(np.int64)
simd_key = (xcc << 16) | (((hw >> 13) & 7) << 12) | (((hw >> 12) & 1) << 11) | (((hw >> 8) & 0xF) << 4) | ((hw >> 4) & 3)
n_simd = len(np.unique(simd_key))
resident = life.sum() / (span * ns_per_tick) / n_simd
print(f"{n_simd} SIMDs seen; average resident waves per SIMD over the launch: {resident:.2f} of 8")
# time-weighted distribution of resident waves per CU (32 slots = 4 workgroups), middle 80 % of the launch
hist = np.zeros(40)
for c in np.unique(cu_key):
    m = cu_key == c
    t_start, t_end = ts[m, 0], ts[m, 11]
    lo, hi = t_start.min(), t_end.max()
    a, b = lo + (hi - lo) // 10, hi - (hi - lo) // 10
    ev = np.concatenate([np.stack([t_start, np.ones_like(t_start)], 1), np.stack([t_end, -np.ones_like(t_end)], 1)])
    ev = ev[np.argsort(ev[:, 0], kind="stable")]
    level = np.cumsum(ev[:, 1])
    tt = np.clip(ev[:, 0], a, b)
    dt = np.diff(tt)
    np.add.at(hist, np.clip(level[:-1], 0, 39), dt)
hist /= hist.sum()
print("resident waves per CU (time share, middle 80 % of the launch): " +
      ", ".join(f"{i}:{hist[i] * 100:.0f}%" for i in range(40) if hist[i] >= 0.005))
print(f"mean resident waves per CU {float((hist * np.arange(40)).sum()):.2f} of 32")
# start-time skew of the 8 waves of a workgroup at the end (slots held until the last wave leaves)
end = ts[:, 11].reshape(-1, 8)
tail = (end.max(axis=1, keepdims=True) - end).astype(np.float64) * ns_per_tick
print(f"wave slots idle while the rest of the workgroup finishes: mean {tail.mean() / 1e3:.3f} us per wave ({tail.mean() / life.mean() * 100:.1f} % of a life)")
if args.out:
    json.dump({"ms_default": ms_default, "ms_traced": ms_traced, "ns_per_tick": ns_per_tick, "wave_life_us": life.mean() / 1e3,
               "phases": rows, "simds": n_simd, "resident_waves_per_simd": resident,
               "wg_tail_idle_us": tail.mean() / 1e3}, open(args.out, "w"), indent=1)
