#!/bin/bash
# tools/clock_power.sh TAG -- run on the GPU box: sclk / power samples (rocm-smi, every ~0.25 s) beside a sustained
# 3,000-step bench run, so that it is on record whether the part holds its clock down on this instruction mix
# (VERDICT r01 #1 iii).  Output: gpurun_out/clock_TAG/{smi.txt,bench.json}
TAG=${1:-r02}
OUT=gpurun_out/clock_$TAG
mkdir -p "$OUT"
( for i in $(seq 1 40); do
    echo "t=$(date +%s.%N)"
    rocm-smi --showclocks --showpower --showtemp 2>&1 | grep -E "sclk|mclk|fclk|Power|Temperature \(Sensor (edge|junction)" | head -8
    sleep 0.25
  done ) > "$OUT/smi.txt" 2>&1 &
SMI=$!
python3 bench.py --steps 3000 --warmup 10 --no-cpu-baseline --no-secondary > "$OUT/bench.json" 2> "$OUT/bench.err"
wait $SMI
echo "clock/power log $TAG done"
