python3 tools/sweep.py --n 1024 --primes 4 --batch 16384 --launches 30 --op mul -- -2
python3 tools/sweep.py --n 1024 --primes 4 --batch 16384 --launches 50 -- -2
python3 tools/sweep.py --n 8192 --primes 4 --batch 2048 --launches 50 -- -2
