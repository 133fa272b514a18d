#!/bin/bash
# tools/small_sweep.sh TAG -- forward / inverse / product sweeps of the small-size kernels (n = 32 ... 512) at >= 1 GiB per launch,
# 60-bit and 30-bit moduli, the radix-2 kernel (-1) beside the tuned default (-2) and any extra registry ids given per size in IDS_<n>
# (60-bit) / IDS30_<n> (30-bit); A/B ids need AGX_NTT_LIB=agilex-ntt_amd/lib/libagxntt_diag.so.
TAG=${1:-small}
OUT=gpurun_out/${TAG}.txt
: > $OUT
for n in 32 64 128 256 512; do
  batch=$(( 1073741824 / 4 / 8 / n ))
  ids_var=IDS_$n
  for bits in 60 30; do
    extra="${!ids_var}"
    ids30_var=IDS30_$n
    [ "$bits" = 30 ] && extra="${!ids30_var}"
    for op in fwd inv mul; do
      echo "== n=$n bits=$bits op=$op batch=$batch (4 primes)" >> $OUT
      python3 tools/sweep.py --n $n --primes 4 --batch $batch --bits $bits --op $op --launches 20 -1 -2 $extra >> $OUT 2>&1 || exit 1
    done
  done
done
tail -n 200 $OUT
