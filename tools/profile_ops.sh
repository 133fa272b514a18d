#!/bin/bash
# tools/profile_ops.sh TAG [op ...]  -- run on the GPU box: counters behind every `secondary` line of bench.py.
# For each operation: rocprofv3 --kernel-trace --stats, then the two SQ groups, FETCH_SIZE, WRITE_SIZE and the L2 / clock
# group, each in its own run (no sys/hip/hsa trace next to --pmc, as the pool requires; the program after `--` is
# python3 itself).  Output: gpurun_out/prof_TAG/<op>/{trace,pmc_sq1,pmc_sq2,pmc_fetch,pmc_write,pmc_l2};
# tools/summarize_ops.py TAG condenses them into profiles/TAG_<op>_summary.md.
set -u
TAG=${1:-r03}
shift || true
OUT=gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
python3 -c "import agilex_ntt_amd as a; print(a.kernel_source_sha16())" > "$OUT/kernel_source_sha16.txt" 2>/dev/null

declare -A OPS
OPS[fwd4096]="--op fwd --n 4096 --primes 4 --batch 4096"
OPS[inv4096]="--op inv --n 4096 --primes 4 --batch 4096"
OPS[mul4096]="--op mul --n 4096 --primes 4 --batch 4096"
# shapes = the ones bench.py's secondary lines time: config 4's slice is 8 primes x batch 8192 = 8 GiB in place
OPS[fwd16384]="--op fwd --n 16384 --primes 8 --batch 8192 --slabs 1"
OPS[inv16384]="--op inv --n 16384 --primes 8 --batch 8192 --slabs 1"
OPS[fwd32768oop]="--op fwd --n 32768 --primes 1 --batch 1024 --slabs 3 --oop"
OPS[fwd32768ip]="--op fwd --n 32768 --primes 1 --batch 1024 --slabs 3"
OPS[inv32768]="--op inv --n 32768 --primes 1 --batch 1024 --slabs 3 --oop"
OPS[mul32768]="--op mul --n 32768 --primes 1 --batch 1024 --slabs 1 --mulsets 3"
OPS[fwd1024q30]="--op fwd --n 1024 --primes 4 --batch 4096 --bits 30"
OPS[fwd4096q30]="--op fwd --n 4096 --primes 4 --batch 4096 --bits 30"
OPS[inv4096q30]="--op inv --n 4096 --primes 4 --batch 4096 --bits 30"
OPS[mul4096q30]="--op mul --n 4096 --primes 4 --batch 4096 --bits 30"
# the small sizes (wave-packed kernels): the bench's four 512 MiB slabs, i.e. 16 Mi coefficients per prime per launch
OPS[fwd32]="--op fwd --n 32 --primes 4 --batch 524288"
OPS[inv32]="--op inv --n 32 --primes 4 --batch 524288"
OPS[mul32]="--op mul --n 32 --primes 4 --batch 524288"
OPS[fwd256]="--op fwd --n 256 --primes 4 --batch 65536"
OPS[fwd512]="--op fwd --n 512 --primes 4 --batch 32768"
OPS[inv512]="--op inv --n 512 --primes 4 --batch 32768"
OPS[mul512]="--op mul --n 512 --primes 4 --batch 32768"
OPS[fwd32q30]="--op fwd --n 32 --primes 4 --batch 524288 --bits 30"
OPS[fwd512q30]="--op fwd --n 512 --primes 4 --batch 32768 --bits 30"

LIST=("$@")
if [ ${#LIST[@]} -eq 0 ]; then LIST=(fwd4096 inv4096 mul4096 fwd16384 inv16384 fwd32768oop inv32768 mul32768 fwd1024q30 fwd4096q30 inv4096q30 mul4096q30 fwd32 inv32 mul32 fwd256 fwd512 inv512 mul512 fwd32q30 fwd512q30); fi

for op in "${LIST[@]}"; do
  ARGS=${OPS[$op]:-}
  if [ -z "$ARGS" ]; then echo "unknown op $op" >> "$OUT/errors.txt"; continue; fi
  D="$OUT/$op"
  mkdir -p "$D"
  echo "$ARGS" > "$D/args.txt"
  RUN="python3 tools/run_op.py $ARGS --launches 20 --ramp-seconds 0"      # counters do not depend on the clock; 5 warm-up + 20 counted launches
  rocprofv3 --kernel-trace --stats -f csv -d "$D/trace" -- python3 tools/run_op.py $ARGS --launches 100 --report "$D/trace_report.json" > "$D/trace.log" 2>&1 || echo "$op trace failed" >> "$OUT/errors.txt"
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -f csv -d "$D/pmc_sq1" -- $RUN > "$D/pmc_sq1.log" 2>&1 || echo "$op pmc_sq1 failed" >> "$OUT/errors.txt"
  rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS -f csv -d "$D/pmc_sq2" -- $RUN > "$D/pmc_sq2.log" 2>&1 || echo "$op pmc_sq2 failed" >> "$OUT/errors.txt"
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d "$D/pmc_fetch" -- $RUN > "$D/pmc_fetch.log" 2>&1 || echo "$op pmc_fetch failed" >> "$OUT/errors.txt"
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d "$D/pmc_write" -- $RUN > "$D/pmc_write.log" 2>&1 || echo "$op pmc_write failed" >> "$OUT/errors.txt"
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum -f csv -d "$D/pmc_l2" -- $RUN > "$D/pmc_l2.log" 2>&1 || echo "$op pmc_l2 failed" >> "$OUT/errors.txt"
  # keep what the summary needs, drop the bulky per-dispatch agent/marker files
  find "$D" -name "*agent_info.csv" -delete
  echo "profiled $op"
done
echo "profile_ops $TAG done"
