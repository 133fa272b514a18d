for lib in tools/ab_old/libagxntt_oldpad.so agilex-ntt_amd/lib/libagxntt.so; do
echo "=== $lib"
export AGX_NTT_LIB=$lib
python3 tools/sweep.py --launches 100 -- -2 2>&1 | tail -1
python3 tools/sweep.py --n 16384 --primes 8 --batch 2048 --launches 20 -- -2 2>&1 | tail -1
python3 tools/sweep.py --n 16384 --primes 8 --batch 2048 --launches 20 --op inv -- -2 2>&1 | tail -1
python3 tools/sweep.py --n 32768 --primes 1 --batch 1024 --launches 50 -- -2 2>&1 | tail -1
python3 tools/sweep.py --n 32768 --primes 1 --batch 1024 --launches 50 --op inv -- -2 2>&1 | tail -1
python3 tools/sweep.py --n 32768 --primes 1 --batch 1024 --launches 30 --op mul -- -2 2>&1 | tail -1
python3 tools/sweep.py --n 8192 --primes 4 --batch 2048 --launches 50 -- -2 2>&1 | tail -1
python3 tools/sweep.py --n 2048 --primes 4 --batch 8192 --launches 50 -- -2 2>&1 | tail -1
python3 tools/sweep.py --n 1024 --primes 4 --batch 16384 --launches 50 -- -2 2>&1 | tail -1
python3 tools/sweep.py --bits 30 --launches 200 -- -2 2>&1 | tail -1
python3 tools/sweep.py --bits 30 --n 16384 --batch 1024 --launches 200 -- -2 2>&1 | tail -1
python3 tools/sweep.py --bits 30 --n 32768 --primes 2 --batch 1024 --slabs 2 --launches 100 -- -2 2>&1 | tail -1
python3 tools/sweep.py --bits 30 --op mul --launches 100 -- -2 2>&1 | tail -1
done
