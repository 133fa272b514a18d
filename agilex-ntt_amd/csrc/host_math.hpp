// host_math.hpp -- host-side number theory for plan creation: NTT-friendly primes,
// primitive roots and the twiddle / precomputed-quotient tables the reference expects its
// caller to provide (include/kernel/ntt.h:35-41; src/main.cpp:49-55 ships placeholders only).
#pragma once
#include <cstdint>
#include <vector>

namespace agx {

uint64_t mul_mod(uint64_t a, uint64_t b, uint64_t q);
uint64_t pow_mod(uint64_t a, uint64_t e, uint64_t q);
uint64_t inv_mod(uint64_t a, uint64_t q);  // q prime
bool is_prime_u64(uint64_t q);
inline bool is_pow2(uint32_t n) { return n && !(n & (n - 1)); }
inline int log2u(uint32_t n) { int l = 0; while ((1u << l) < n) ++l; return l; }
inline uint32_t bit_reverse(uint32_t x, int bits) {
    uint32_t r = 0;
    for (int i = 0; i < bits; ++i) { r = (r << 1) | (x & 1u); x >>= 1; }
    return r;
}

// descending list of primes below 2^bits congruent to 1 mod 2n
std::vector<uint64_t> find_ntt_primes(uint32_t bits, uint32_t n, uint32_t count);
// least primitive 2n-th root of unity mod q, 0 if q-1 is not divisible by 2n
uint64_t min_primitive_root_2n(uint64_t q, uint32_t n);
bool is_primitive_root_2n(uint64_t psi, uint64_t q, uint32_t n);

// floor(w * 2^64 / q), w < q
uint64_t shoup_quotient(uint64_t w, uint64_t q);
// tw[j] = base^bitrev(j) mod q, pre[j] = shoup_quotient(tw[j])
void power_tables_bitrev(uint64_t q, uint64_t base, uint32_t n, uint64_t* tw, uint64_t* pre);

}  // namespace agx
