set -x
python3 tools/sweep.py --n 32768 --primes 1 --batch 1024 --launches 50 114 116
python3 tools/sweep.py --n 32768 --primes 1 --batch 1024 --launches 50 --op inv 114 116
python3 tools/sweep.py --n 32768 --primes 1 --batch 8192 --slabs 2 --launches 20 114 116
python3 tools/sweep.py --n 32768 --primes 1 --batch 8192 --slabs 2 --launches 20 --op inv 114 116
python3 tools/sweep.py --n 16384 --primes 8 --batch 2048 --launches 20 115 118 43
python3 tools/sweep.py --n 16384 --primes 8 --batch 2048 --launches 20 --op inv 115 117 43
