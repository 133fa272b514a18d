set -x
python3 tools/sweep.py --bits 30 --n 1024 --primes 4 --batch 16384 130 136 63
python3 tools/sweep.py --bits 30 --n 2048 --primes 4 --batch 8192 131 137 59
python3 tools/sweep.py --bits 30 --n 4096 --primes 4 --batch 4096 132 138 93
python3 tools/sweep.py --bits 30 --n 8192 --primes 4 --batch 2048 133 139 64
python3 tools/sweep.py --bits 30 --n 16384 --primes 4 --batch 1024 134 140 117
python3 tools/sweep.py --bits 30 --n 32768 --primes 2 --batch 1024 --slabs 2 135 141 119
python3 tools/sweep.py --bits 30 --n 4096 --primes 4 --batch 4096 --op inv 132 138 93
python3 tools/sweep.py --bits 30 --n 4096 --primes 4 --batch 4096 --op mul 132 138 93
python3 tools/sweep.py --bits 31 --n 4096 --primes 4 --batch 4096 138 93
python3 tools/sweep.py --bits 17 --n 4096 --primes 1 --batch 4096 132 138 93
