#!/bin/bash
# tools/profile.sh TAG  -- run on the GPU box: rocprofv3 kernel-trace stats + separate PMC passes
# for the bench workload, everything under gpurun_out/prof_TAG/.  Summaries are then condensed by
# tools/summarize_profile.py into profiles/ (tracked).
# Counter passes are separate runs with --kernel-trace only (no sys/hip/hsa trace), as the box requires.
set -u
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
# the device sources these counters are measured on (summarize_profile.py copies this into profiles/hbm_traffic.json; bench.py
# reports roofline.traffic only while the tree still hashes to it)
python3 -c "import agilex_ntt_amd as a; print(a.kernel_source_sha16())" > "$OUT/kernel_source_sha16.txt"
BENCH="python3 bench.py --steps 20 --warmup 3 --ramp-seconds 0.05 --no-cpu-baseline --no-secondary"
# the stats pass runs bench.py's default step counts so its per-kernel average is the bench's own
rocprofv3 --kernel-trace --stats -f csv -d "$OUT/trace" -- python3 bench.py --no-cpu-baseline --no-secondary > "$OUT/trace.log" 2>&1 || echo "trace pass failed" >> "$OUT/errors.txt"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -f csv -d "$OUT/pmc_sq1" -- $BENCH > "$OUT/pmc_sq1.log" 2>&1 || echo "pmc_sq1 failed" >> "$OUT/errors.txt"
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS -f csv -d "$OUT/pmc_sq2" -- $BENCH > "$OUT/pmc_sq2.log" 2>&1 || echo "pmc_sq2 failed" >> "$OUT/errors.txt"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d "$OUT/pmc_fetch" -- $BENCH > "$OUT/pmc_fetch.log" 2>&1 || echo "pmc_fetch failed" >> "$OUT/errors.txt"
rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d "$OUT/pmc_write" -- $BENCH > "$OUT/pmc_write.log" 2>&1 || echo "pmc_write failed" >> "$OUT/errors.txt"
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum -f csv -d "$OUT/pmc_l2" -- $BENCH > "$OUT/pmc_l2.log" 2>&1 || echo "pmc_l2 failed" >> "$OUT/errors.txt"
find "$OUT" -name "*.csv" | head -50 > "$OUT/files.txt"
echo "profile $TAG done"
