"""tools/summarize_ops.py TAG [op ...] -- condense gpurun_out/prof_TAG/<op>/ (written by tools/profile_ops.sh on the GPU
box) into profiles/TAG_<op>_summary.md, one file per operation, plus profiles/TAG_ops_table.md with one row per operation.

Per kernel of the operation (every agx kernel but the synthetic fill): launch count and average duration from
--kernel-trace --stats (the last 100 dispatches = run_op.py's timed launches), VGPR / scratch / LDS of the dispatch, the SQ
counters as means per dispatch, VALU instructions per wave and per butterfly, wave-cycle shares, and HBM bytes per launch
(FETCH_SIZE x 2 on gfx950 for wide coalesced reads + WRITE_SIZE, separate --pmc passes, MI355X_MICROARCH.md) against
the algorithmic bytes of the launch."""
import collections
import csv
import glob
import math
import os
import re
import sys

import json

argv = sys.argv[1:]
bench = None
if "--bench" in argv:      # a bench.py JSON line of the same box: its secondary kernel_ms beside the profiler's
    i = argv.index("--bench")
    try:
        bench = json.load(open(argv[i + 1]))
    except (OSError, ValueError):
        bench = None
    del argv[i:i + 2]
table_name = None
if "--table" in argv:      # name of the one-row-per-operation table (default TAG_ops_table.md): several box sessions of one TAG keep separate tables
    i = argv.index("--table")
    table_name = argv[i + 1]
    del argv[i:i + 2]
tag = argv[0]
root = os.path.join("gpurun_out", f"prof_{tag}")
ops = argv[1:] or sorted(d for d in os.listdir(root) if os.path.isdir(os.path.join(root, d)))
# profile op -> name of the bench line that times the same launch shape
BENCH_LINE = {"fwd4096": None, "inv4096": "inverse_n4096", "mul4096": "polymul_n4096", "fwd16384": "forward_n16384_config4_slice", "inv16384": "inverse_n16384_config4_slice",
              "fwd32768oop": "forward_n32768", "inv32768": "inverse_n32768", "mul32768": "polymul_n32768_config5_slice", "fwd1024q30": "forward_n1024_30bit",
              "fwd4096q30": "forward_n4096_30bit", "inv4096q30": "inverse_n4096_30bit", "mul4096q30": "polymul_n4096_30bit", "fwd32": "forward_n32", "inv32": "inverse_n32",
              "mul32": "polymul_n32", "fwd256": "forward_n256", "fwd512": "forward_n512", "inv512": "inverse_n512", "mul512": "polymul_n512", "fwd32q30": "forward_n32_30bit",
              "fwd512q30": "forward_n512_30bit"}


def bench_ms(op):
    if not bench:
        return None
    if op == "fwd4096":
        return bench.get("roofline", {}).get("kernel_ms")
    for e in bench.get("secondary", []):
        if e.get("name") == BENCH_LINE.get(op):
            return e.get("kernel_ms")
    return None
SKIP = ("fill_kernel", "__amd_rocclr", "copyBuffer", "fillBuffer")


def short(name):
    name = name.replace("void ", "")
    return name.split("(")[0]


def parse_args(s):
    a = s.split()
    g = lambda k, d: a[a.index(k) + 1] if k in a else d
    return {"op": g("--op", "inv"), "n": int(g("--n", 4096)), "primes": int(g("--primes", 4)), "batch": int(g("--batch", 4096)),
            "bits": int(g("--bits", 60)), "oop": "--oop" in a}


table = ["| op | kernel(s) | avg µs per launch (sum) | bench.py kernel µs (same box) | profiler ÷ bench | frac of 8 TB/s | HBM bytes ÷ algorithmic | VALU / butterfly | WAIT_ANY share | WAIT_INST_ANY share | VGPR | scratch B | profile |", "|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
sha = ""
try:
    sha = open(os.path.join(root, "kernel_source_sha16.txt")).read().strip()
except OSError:
    pass

for op in ops:
    d = os.path.join(root, op)
    try:
        spec = parse_args(open(os.path.join(d, "args.txt")).read())
    except OSError:
        continue
    n, units = spec["n"], spec["primes"] * spec["batch"]
    logn = int(math.log2(n))
    transforms = 3 if spec["op"] == "mul" else 1
    bfly_per_launch = units * transforms * (n // 2) * logn
    algo = units * (24 if spec["op"] == "mul" else 16) * n
    lines = [f"# rocprofv3 summary `{tag}` / `{op}` on one MI355X", "",
             f"Command: `python3 tools/run_op.py {open(os.path.join(d, 'args.txt')).read().strip()}` (100 timed launches under `--kernel-trace --stats`, 20 under each `--pmc` group; "
             f"kernel sources sha16 `{sha}`).", "",
             f"Workload: n={n}, {spec['primes']} prime(s) of {spec['bits']} bits, batch {spec['batch']} -> {units} units per launch, "
             f"{'out of place' if spec['oop'] else 'in place'}; algorithmic bytes per launch {algo} ({algo / 2**20:.0f} MiB); "
             f"butterflies per launch {bfly_per_launch} ({transforms} transform(s) x n/2 x log2 n).", ""]
    trace = glob.glob(os.path.join(d, "trace", "*", "*kernel_trace.csv"))
    kern = collections.OrderedDict()
    if trace:
        rows = [r for r in csv.DictReader(open(trace[0])) if "agx::" in r["Kernel_Name"] and not any(s in r["Kernel_Name"] for s in SKIP)]
        rows.sort(key=lambda r: int(r["Start_Timestamp"]))
        by = collections.defaultdict(list)
        for r in rows:
            by[r["Kernel_Name"]].append(r)
        # launches per timed step of each kernel = its dispatch count / (clock ramp + 5 warm-up + 100 timed launches of the operation)
        calls = 105
        try:
            calls = json.load(open(os.path.join(d, "trace_report.json")))["calls"]
        except (OSError, ValueError, KeyError):
            pass
        for name, rs in by.items():
            per_step = max(1, round(len(rs) / calls))
            last = rs[-100 * per_step:]
            dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in last]
            kern[name] = {"calls": len(rs), "per_step": per_step, "avg_ns": sum(dur) / len(dur), "min_ns": min(dur), "max_ns": max(dur),
                          "vgpr": rs[-1]["VGPR_Count"], "agpr": rs[-1]["Accum_VGPR_Count"], "sgpr": rs[-1]["SGPR_Count"], "scratch": rs[-1]["Scratch_Size"],
                          "lds": rs[-1]["LDS_Block_Size"], "wg": rs[-1]["Workgroup_Size_X"], "grid": rs[-1]["Grid_Size_X"]}
    if not kern:
        lines.append("(no kernel trace found)")
        open(os.path.join("profiles", f"{tag}_{op}_summary.md"), "w").write("\n".join(lines) + "\n")
        continue
    step_ns = sum(k["avg_ns"] * k["per_step"] for k in kern.values())
    lines += ["## kernel-trace (timed launches only)", "", "| kernel | dispatches | per step | avg ns | min | max | VGPR | AGPR | SGPR | scratch B/lane | LDS B | workgroup | grid |", "|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
    for name, k in kern.items():
        lines.append(f"| `{short(name)}` | {k['calls']} | {k['per_step']} | {k['avg_ns']:.0f} | {k['min_ns']} | {k['max_ns']} | {k['vgpr']} | {k['agpr']} | {k['sgpr']} | {k['scratch']} | {k['lds']} | {k['wg']} | {k['grid']} |")
    frac = algo / (step_ns * 1e-9) / 8e12
    lines += ["", f"One step = {step_ns / 1e3:.1f} µs of kernel time -> {units / (step_ns * 1e-9) / 1e6:.2f} M units/s, "
              f"{algo / (step_ns * 1e-9) / 1e9:.0f} GB/s algorithmic = **{frac:.3f} of 8 TB/s**.", ""]

    counters = {name: collections.OrderedDict() for name in kern}
    for grp in ["pmc_sq1", "pmc_sq2", "pmc_fetch", "pmc_write", "pmc_l2"]:
        f = glob.glob(os.path.join(d, grp, "*", "*counter_collection.csv"))
        if not f:
            continue
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f[0])):
            if r["Kernel_Name"] in kern:
                acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for name, cs in acc.items():
            for c, v in cs.items():
                v = v[len(v) // 5:]          # drop the warm-up launches (5 of 25)
                counters[name][c] = sum(v) / len(v)
    tot_rd = tot_wr = 0.0
    have_traffic = True
    valu_total = 0.0
    wait_any = wait_inst = wave_cyc = 0.0
    for name, c in counters.items():
        k = kern[name]
        lines += [f"## PMC, mean per dispatch: `{short(name)}`", "", "| counter | mean |", "|---|---|"]
        for cn, v in c.items():
            lines.append(f"| {cn} | {v:.6g} |")
        lines.append("")
        if "SQ_WAVE_CYCLES" in c:
            wc = c["SQ_WAVE_CYCLES"]
            for cn in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_VALU"):
                if cn in c:
                    lines.append(f"- {cn} / SQ_WAVE_CYCLES = {c[cn] / wc:.3f}")
            if "SQ_INSTS_VALU" in c and "SQ_WAVES" in c:
                lines.append(f"- VALU instructions per wave = {c['SQ_INSTS_VALU'] / c['SQ_WAVES']:.0f}; wave lifetime = {wc * 4 / c['SQ_WAVES']:.0f} cycles (SQ_WAVE_CYCLES x 4 / SQ_WAVES)")
                valu_total += c["SQ_INSTS_VALU"] * 64 * k["per_step"]
            wait_any += c.get("SQ_WAIT_ANY", 0) * k["per_step"]
            wait_inst += c.get("SQ_WAIT_INST_ANY", 0) * k["per_step"]
            wave_cyc += wc * k["per_step"]
        if "SQ_LDS_BANK_CONFLICT" in c and c.get("SQ_LDS_IDX_ACTIVE"):
            lines.append(f"- LDS bank-conflict cycles / LDS active cycles = {c['SQ_LDS_BANK_CONFLICT'] / c['SQ_LDS_IDX_ACTIVE']:.3f}")
        if "GRBM_GUI_ACTIVE" in c:
            lines.append(f"- effective clock = GRBM_GUI_ACTIVE / 8 / avg kernel time = {c['GRBM_GUI_ACTIVE'] / 8 / k['avg_ns']:.2f} GHz")
        if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c and c["TCC_HIT_sum"] + c["TCC_MISS_sum"] > 0:
            lines.append(f"- L2 hit rate = {c['TCC_HIT_sum'] / (c['TCC_HIT_sum'] + c['TCC_MISS_sum']):.3f}")
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            rd, wr = c["FETCH_SIZE"] * 1024 * 2, c["WRITE_SIZE"] * 1024
            tot_rd += rd * k["per_step"]
            tot_wr += wr * k["per_step"]
            lines.append(f"- HBM traffic per dispatch: read {rd / 2**20:.1f} MiB (FETCH_SIZE {c['FETCH_SIZE']:.0f} KiB x 2, gfx950 correction), write {wr / 2**20:.1f} MiB")
        else:
            have_traffic = False
        lines.append("")
    ratio = (tot_rd + tot_wr) / algo if have_traffic else None
    vpb = valu_total / bfly_per_launch if valu_total else None
    lines += ["## per step", ""]
    if ratio is not None:
        lines.append(f"- HBM bytes per step (all kernels) = {(tot_rd + tot_wr) / 2**20:.1f} MiB = **{ratio:.3f} x algorithmic** ({algo / 2**20:.0f} MiB)")
    if vpb is not None:
        lines.append(f"- VALU lane-instructions per butterfly (everything included) = **{vpb:.1f}**")
    if wave_cyc:
        lines.append(f"- SQ_WAIT_ANY / SQ_WAVE_CYCLES = {wait_any / wave_cyc:.3f}; SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES = {wait_inst / wave_cyc:.3f}")
    open(os.path.join("profiles", f"{tag}_{op}_summary.md"), "w").write("\n".join(lines) + "\n")
    first = next(iter(kern.values()))
    bms = bench_ms(op)
    if bms:
        lines += ["", f"bench.py on the same box (`secondary.kernel_ms` of `{BENCH_LINE.get(op) or 'the headline'}`): {bms * 1e3:.1f} µs per launch; profiler ÷ bench = {step_ns / 1e3 / (bms * 1e3):.3f}."]
        open(os.path.join("profiles", f"{tag}_{op}_summary.md"), "a").write("\n".join(lines[-2:]) + "\n")
    table.append(f"| {op} | {', '.join('`' + short(nm).split('::')[-1].split('<')[0] + '`' for nm in kern)} | {step_ns / 1e3:.1f} | {'%.1f' % (bms * 1e3) if bms else 'n/a'} | {'%.3f' % (step_ns / 1e3 / (bms * 1e3)) if bms else 'n/a'} | {frac:.3f} | "
                 f"{'%.3f' % ratio if ratio is not None else 'n/a'} | {'%.1f' % vpb if vpb else 'n/a'} | "
                 f"{'%.2f' % (wait_any / wave_cyc) if wave_cyc else 'n/a'} | {'%.2f' % (wait_inst / wave_cyc) if wave_cyc else 'n/a'} | "
                 f"{'/'.join(k['vgpr'] for k in kern.values())} | {'/'.join(k['scratch'] for k in kern.values())} | `profiles/{tag}_{op}_summary.md` |")
    print(f"{op}: {step_ns / 1e3:.1f} us/step frac {frac:.3f} traffic ratio {ratio} valu/bfly {vpb}")

open(os.path.join("profiles", table_name or f"{tag}_ops_table.md"), "w").write(
    f"# `{tag}`: counters behind every secondary line (tools/profile_ops.sh, tools/summarize_ops.py; kernel sources sha16 `{sha}`)\n\n" + "\n".join(table) + "\n")
