#!/bin/bash
# tools/all_sizes.sh TAG -- the tuned defaults of every size n = 32 ... 32768, forward / inverse / product, 60- and 30-bit moduli, at 1 GiB of
# algorithmic traffic per launch (4 primes x batch, in place, 4 slabs rotated), 20 launches per timing: one table for DESIGN.md section 4.
TAG=${1:-all_sizes}
OUT=gpurun_out/${TAG}.txt
: > $OUT
for n in 2 4 8 16 32 64 128 256 512 1024 2048 4096 8192 16384 32768; do
  batch=$(( 1073741824 / 4 / 16 / n ))
  [ $batch -lt 256 ] && batch=256
  for bits in 60 30; do
    for op in fwd inv mul; do
      echo "== n=$n bits=$bits op=$op batch=$batch (4 primes)" >> $OUT
      python3 tools/sweep.py --n $n --primes 4 --batch $batch --bits $bits --op $op --launches 20 -2 2>&1 | grep -v amdgpu.ids >> $OUT || exit 1
    done
  done
done
python3 - "$OUT" <<'PY'
import re, sys
rows = {}
hdr = None
for line in open(sys.argv[1]):
    m = re.match(r"== n=(\d+) bits=(\d+) op=(\w+)", line)
    if m:
        hdr = (int(m.group(1)), int(m.group(2)), m.group(3))
        continue
    f = line.split()
    if hdr and len(f) == 7 and f[0] == "-2":
        rows[hdr] = (f[1], float(f[4]), float(f[6]))
print("| n | fwd 60-bit | inv 60-bit | product 60-bit | fwd 30-bit | inv 30-bit | product 30-bit |")
print("|---|---|---|---|---|---|---|")
for n in sorted({k[0] for k in rows}):
    cells = []
    for bits in (60, 30):
        for op in ("fwd", "inv", "mul"):
            r = rows.get((n, bits, op))
            cells.append("n/a" if not r else f"{r[2]:.1f} % ({r[1]:.1f} M/s){'' if r[0] == 'True' else ' MISMATCH'}")
    print(f"| {n} | " + " | ".join(cells) + " |")
PY
