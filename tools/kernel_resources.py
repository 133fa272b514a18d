"""tools/kernel_resources.py -- VGPRs / SGPRs / scratch / occupancy / LDS of every kernel one registry translation unit
instantiates (hipcc -Rpass-analysis=kernel-resource-usage, device code only; no GPU needed).
Usage: python tools/kernel_resources.py reg_wp.hip [-DAGX_DIAG] [--grep PATTERN]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
pat = None
if "--grep" in args:
    i = args.index("--grep")
    pat = args[i + 1]
    del args[i:i + 2]
tu = args[0]
extra = args[1:]
obj = os.path.join(tempfile.gettempdir(), "agx_res_" + os.path.basename(tu) + ".o")
r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-ffp-contract=off",
                    "--cuda-device-only", "-c", "-o", obj, os.path.join(ROOT, "agilex-ntt_amd", "csrc", tu),
                    "-Rpass-analysis=kernel-resource-usage"] + extra, capture_output=True, text=True)
if r.returncode:
    sys.exit(r.stderr[-4000:])
cur = None
rows = []
for line in r.stderr.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()}
        rows.append(cur)
        continue
    m = re.search(r"remark:\s+(VGPRs|AGPRs|SGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).split(" ")[0]] = int(m.group(2))
print(f"{'VGPR':>5} {'SGPR':>5} {'scr':>4} {'occ':>3} {'LDS':>6}  kernel")
for k in rows:
    short = re.sub(r"\(.*", "", k["name"]).replace("void agx::", "")
    if pat and not re.search(pat, short):
        continue
    print(f"{k.get('VGPRs', 0):5d} {k.get('SGPRs', 0):5d} {k.get('ScratchSize', 0):4d} {k.get('Occupancy', 0):3d} {k.get('LDS', 0):6d}  {short}")
