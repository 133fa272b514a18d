"""tools/isa_histogram.py -- opcode histogram of a shipped kernel's gfx950 code ("commit an opcode
histogram of the default kernel's disassembly").  Compiles one registry translation unit to assembly with the product
flags (hipcc --cuda-device-only -S; no GPU needed), finds the kernel whose mangled name matches PATTERN, and prints
  * every opcode with its static count (the n=4096 kernels are straight-line code: static = executed per wave, except
    for the listed wave-uniform branches),
  * the counts grouped by what the instruction is for, per wave and per butterfly,
  * an issue-cost estimate from the box calibration (profiles/r01g_alu_issue_calibration.txt, 8 waves/SIMD,
    cost relative to v_add_u32 = one 2-cycle issue slot).
Usage: python tools/isa_histogram.py [--tu reg_n4096.hip] [--kernel 'fwd_rb2ILi12ELi3ELi1ELi1850727'] [--butterflies 48] [--out FILE.md]"""
import argparse
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--tu", default="reg_n4096.hip")
ap.add_argument("--kernel", default="fwd_rb2ILi12ELi3ELi1ELi1850727")
ap.add_argument("--butterflies", type=int, default=48, help="butterflies per thread (n=4096, R=3: 12 stages x 4)")
ap.add_argument("--asm", default=None, help="reuse an existing .s file instead of compiling")
ap.add_argument("--out", default=None)
args = ap.parse_args()

asm = args.asm
if not asm:
    asm = os.path.join(tempfile.gettempdir(), "agx_" + args.tu.replace(".hip", ".s"))
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                           "--cuda-device-only", "-S", "-o", asm, os.path.join(ROOT, "agilex-ntt_amd", "csrc", args.tu)],
                          stderr=subprocess.DEVNULL)
lines = open(asm).read().split("\n")
start = next((i for i, l in enumerate(lines) if re.match(r"^_Z\w*:", l) and args.kernel in l), None)
if start is None:
    sys.exit(f"no kernel matching {args.kernel} in {asm}")
name = lines[start].split(":")[0]
end = start
while ".end_amdhsa_kernel" not in lines[end] and not lines[end].strip().startswith(".Lfunc_end"):
    end += 1
body = lines[start:end]
meta = {}
for l in lines[end:end + 80]:
    m = re.match(r";\s*(NumVgprs|TotalNumSgprs|ScratchSize|Occupancy|codeLenInByte):\s*(\d+)", l.strip())
    if m:
        meta.setdefault(m.group(1), int(m.group(2)))

ops = collections.Counter()
for l in body:
    m = re.match(r"^\s+([a-z_0-9]+)\s", l + " ")
    if m and not l.strip().startswith((".", ";")):
        ops[m.group(1)] += 1

# relative issue cost at 8 waves/SIMD, v_add_u32 = 1 (profiles/r01g_alu_issue_calibration.txt, wall-clock column)
COST = {"v_mad_u64_u32": 1.86, "v_mul_hi_u32": 1.55, "v_mul_lo_u32": 1.68, "v_lshl_add_u64": 1.81, "v_add3_u32": 1.66,
        "v_lshl_add_u32": 1.61, "v_and_or_b32": 1.6}
for o in ("v_add_co_u32", "v_addc_co_u32", "v_sub_co_u32", "v_subb_co_u32"):
    COST[o + "_e32"] = COST[o + "_e64"] = 1.60
for o in ("v_cmp_gt_i64", "v_cmp_le_u64", "v_cmp_lt_u64", "v_cmp_ge_u64", "v_cmp_gt_u64", "v_cmp_lt_i64"):
    COST[o + "_e32"] = COST[o + "_e64"] = 1.60


def group(op):
    if op.startswith(("v_mad_u64", "v_mul_hi", "v_mul_lo", "v_mul_u32")):
        return "32x32 multiplies (v_mad_u64_u32 / v_mul_hi_u32 / v_mul_lo_u32)"
    if op.startswith(("v_lshl_add_u64", "v_add_co", "v_addc_co", "v_sub_co", "v_subb_co", "v_lshlrev_b64")):
        return "64-bit add / subtract (v_lshl_add_u64, carry pairs)"
    if op.startswith(("v_cmp", "v_cndmask")):
        return "compare + select (conditional subtracts)"
    if op.startswith(("v_cvt", "v_mul_f32")):
        return "quotient estimate of the final reduction (v_cvt / v_mul_f32)"
    if op.startswith("v_mov"):
        return "register moves"
    if op.startswith("v_"):
        return "other VALU (32-bit adds, address arithmetic, lane ids)"
    if op.startswith("ds_"):
        return "LDS exchange"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vector memory (frame loads / stores, per-lane twiddles)"
    if op.startswith(("s_load", "s_buffer_load")):
        return "scalar loads (kernel arguments, constants, wave-uniform twiddles)"
    if op in ("s_nop", "s_waitcnt", "s_barrier", "s_setprio", "s_endpgm"):
        return "waits / nops / barrier"
    return "scalar ALU and branches"


groups = collections.OrderedDict()
for op, c in ops.most_common():
    g = groups.setdefault(group(op), {"count": 0, "cost": 0.0, "ops": []})
    g["count"] += c
    if op.startswith("v_"):
        g["cost"] += c * COST.get(op, 1.0)
    g["ops"].append(f"{op} {c}")

valu = sum(c for o, c in ops.items() if o.startswith("v_"))
cost = sum(c * COST.get(o, 1.0) for o, c in ops.items() if o.startswith("v_"))
nb = args.butterflies
out = [f"# opcode histogram of `{name}`", "",
       f"source: `agilex-ntt_amd/csrc/{args.tu}`, hipcc -O3 --offload-arch=gfx950 (tools/isa_histogram.py); "
       f"{sum(ops.values())} instructions, {valu} VALU; {meta}", "",
       f"Per thread: {nb} butterflies.  VALU per butterfly (everything included): {valu / nb:.2f}; "
       f"estimated issue slots per wave (v_add_u32 = 1 slot = 2 cycles at 8 waves/SIMD): {cost:.0f} "
       f"= {cost / nb:.1f} per butterfly = {2 * cost / nb:.0f} cycles per wave-butterfly.", "",
       "| purpose | instructions | per butterfly | est. issue slots | share of VALU slots |", "|---|---|---|---|---|"]
for g, d in groups.items():
    share = f"{100 * d['cost'] / cost:.1f} %" if d["cost"] else ""
    out.append(f"| {g} | {d['count']} | {d['count'] / nb:.2f} | {d['cost']:.0f} | {share} |")
out += ["", "## every opcode", "", "| opcode | count |", "|---|---|"]
out += [f"| {o} | {c} |" for o, c in ops.most_common()]
branches = [l.strip() for l in body if re.match(r"^\s+s_cbranch", l)]
out += ["", f"wave-uniform branches: {len(branches)} (`{'`, `'.join(sorted(set(b.split()[0] for b in branches)))}`)"]
text = "\n".join(out) + "\n"
if args.out:
    open(args.out, "w").write(text)
print(text)
