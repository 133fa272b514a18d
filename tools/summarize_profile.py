"""tools/summarize_profile.py TAG [kernel-substring] -- condense gpurun_out/prof_TAG (written by
tools/profile.sh on the GPU box) into profiles/TAG_summary.md and profiles/hbm_traffic.json.

HBM traffic follows MI355X_MICROARCH.md (HBM / rocprofv3): FETCH_SIZE and WRITE_SIZE come from
separate --pmc passes, are reported in KiB, and on gfx950 FETCH_SIZE tallies 128-byte requests at
64 bytes for wide coalesced reads, so the read side is doubled; WRITE_SIZE is exact for
streaming stores."""
import collections
import csv
import glob
import json
import os
import sys

tag = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else "fwd_r"
root = os.path.join("gpurun_out", f"prof_{tag}")
lines = [f"# rocprofv3 summary `{tag}` on one MI355X -- stats pass: `python3 bench.py --no-cpu-baseline` (default 100 steps + 10 warm-up);",
         "# PMC passes: `python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline`, one counter group per run", ""]

stats = glob.glob(os.path.join(root, "trace", "*", "*kernel_stats.csv"))
avg_ns = None
if stats:
    lines += ["## kernel-trace --stats (top kernels)", "", "| kernel | calls | avg ns | total % |", "|---|---|---|---|"]
    for r in list(csv.DictReader(open(stats[0])))[:6]:
        name = r["Name"].split("(")[0][:70]
        lines.append(f"| `{name}` | {r['Calls']} | {float(r['AverageNs']):.0f} | {float(r['Percentage']):.2f} |")
        if want in r["Name"] and avg_ns is None:
            avg_ns = float(r["AverageNs"])
    lines.append("")

# the timed region = the last 100 dispatches of the kernel (bench.py's default --steps); the earlier ones are
# the clock-ramp and warm-up launches, which run on a colder clock
trace = glob.glob(os.path.join(root, "trace", "*", "*kernel_trace.csv"))
if trace:
    rows = [r for r in csv.DictReader(open(trace[0])) if want in r.get("Kernel_Name", "")]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
    if len(dur) >= 100:
        last = dur[-100:]
        avg_ns = sum(last) / len(last)
        first = dur[:100]
        lines += [f"`{want}*` dispatches: {len(dur)}; average of the LAST 100 (bench.py's timed steps) = **{avg_ns:.0f} ns**, "
                  f"min {min(last)} / max {max(last)}; average of the first 100 after idle = {sum(first) / len(first):.0f} ns (clock ramp)", ""]

counters = collections.OrderedDict()
for d in ["pmc_sq1", "pmc_sq2", "pmc_fetch", "pmc_write", "pmc_l2"]:
    f = glob.glob(os.path.join(root, d, "*", "*counter_collection.csv"))
    if not f:
        continue
    acc = collections.defaultdict(list)
    meta = {}
    for r in csv.DictReader(open(f[0])):
        if want in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta = {k: r[k] for k in ("VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Workgroup_Size", "Grid_Size")}
    for k, v in acc.items():
        counters[k] = sum(v) / len(v)
    counters["_meta"] = meta
meta = counters.pop("_meta", {})
lines += [f"## PMC passes, mean per dispatch of `{want}*` (each group its own run)", "", f"dispatch: {meta}", "", "| counter | mean |", "|---|---|"]
for k, v in counters.items():
    lines.append(f"| {k} | {v:.6g} |")
lines.append("")

out = {}
if "FETCH_SIZE" in counters and "WRITE_SIZE" in counters:
    rd = counters["FETCH_SIZE"] * 1024 * 2      # gfx950: wide coalesced reads counted at half
    wr = counters["WRITE_SIZE"] * 1024
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import agilex_ntt_amd as _agx
    # the hash recorded ON THE GPU BOX when the counters were taken (tools/profile.sh); a tree that has moved on since is flagged
    try:
        measured_sha = open(os.path.join(root, "kernel_source_sha16.txt")).read().strip()
    except OSError:
        measured_sha = None
    if measured_sha is None:
        sys.exit("no kernel_source_sha16.txt in " + root + ": re-run tools/profile.sh (it stamps the sources it measured)")
    if measured_sha != _agx.kernel_source_sha16():
        print(f"WARNING: profile {tag} was measured on sources {measured_sha}, the tree is at {_agx.kernel_source_sha16()}: bench.py will report traffic as stale")
    out = {"tag": tag, "kernel_source_sha16": measured_sha, "kernel": want, "fetch_size_kib_raw": counters["FETCH_SIZE"], "write_size_kib_raw": counters["WRITE_SIZE"],
           "read_bytes_corrected": rd, "write_bytes": wr, "bytes_per_launch": rd + wr,
           "note": "FETCH_SIZE x2 per MI355X_MICROARCH.md (gfx950 counts 128-B requests at 64 B); separate --pmc passes"}
    lines += ["## HBM traffic per launch", "", f"read {rd/2**20:.1f} MiB (FETCH_SIZE {counters['FETCH_SIZE']:.0f} KiB x2 gfx950 correction), "
              f"write {wr/2**20:.1f} MiB; algorithmic 512 + 512 MiB.", ""]
    json.dump(out, open(os.path.join("profiles", "hbm_traffic.json"), "w"), indent=1)
if "SQ_WAVE_CYCLES" in counters:
    wc = counters["SQ_WAVE_CYCLES"]
    lines += ["## wave-cycle shares (SQ_* count quad-cycles)", ""]
    for k in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_VALU"):
        if k in counters:
            lines.append(f"- {k} / SQ_WAVE_CYCLES = {counters[k]/wc:.3f}")
    if "SQ_INSTS_VALU" in counters and "SQ_WAVES" in counters:
        lines.append(f"- VALU instructions per wave = {counters['SQ_INSTS_VALU']/counters['SQ_WAVES']:.0f}")
    if "GRBM_GUI_ACTIVE" in counters and avg_ns:
        lines.append(f"- effective clock = GRBM_GUI_ACTIVE/8/avg kernel time = {counters['GRBM_GUI_ACTIVE']/8/avg_ns:.2f} GHz")
    lines.append("")
open(os.path.join("profiles", f"{tag}_summary.md"), "w").write("\n".join(lines))
print("\n".join(lines))
