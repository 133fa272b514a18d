#include "host_math.hpp"

namespace agx {

typedef unsigned __int128 u128;

uint64_t mul_mod(uint64_t a, uint64_t b, uint64_t q) { return (uint64_t)((u128)a * b % q); }

uint64_t pow_mod(uint64_t a, uint64_t e, uint64_t q) {
    uint64_t r = 1 % q, base = a % q;
    for (; e; e >>= 1) {
        if (e & 1) r = mul_mod(r, base, q);
        base = mul_mod(base, base, q);
    }
    return r;
}

uint64_t inv_mod(uint64_t a, uint64_t q) { return pow_mod(a, q - 2, q); }

// Miller-Rabin with the first twelve primes as witnesses: deterministic below 3.3e24
bool is_prime_u64(uint64_t q) {
    if (q < 2) return false;
    const uint64_t wit[] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};
    for (uint64_t p : wit)
        if (q % p == 0) return q == p;
    uint64_t odd = q - 1;
    int twos = 0;
    while ((odd & 1) == 0) { odd >>= 1; ++twos; }
    for (uint64_t a : wit) {
        uint64_t y = pow_mod(a, odd, q);
        if (y == 1 || y == q - 1) continue;
        bool witness = true;
        for (int k = 1; k < twos && witness; ++k) {
            y = mul_mod(y, y, q);
            if (y == q - 1) witness = false;
        }
        if (witness) return false;
    }
    return true;
}

std::vector<uint64_t> find_ntt_primes(uint32_t bits, uint32_t n, uint32_t count) {
    std::vector<uint64_t> found;
    if (bits < 2 || bits > 62 || !is_pow2(n)) return found;
    const uint64_t two_n = 2ull * n;
    const uint64_t limit = (1ull << bits) - 1;
    if (limit <= two_n) return found;
    uint64_t cand = limit - ((limit - 1) % two_n);  // largest value <= limit that is 1 mod 2n
    while (found.size() < count && cand > two_n) {
        if (is_prime_u64(cand)) found.push_back(cand);
        cand -= two_n;
    }
    return found;
}

bool is_primitive_root_2n(uint64_t psi, uint64_t q, uint32_t n) {
    // the multiplicative order divides 2n (a power of two) and is exactly 2n iff psi^n = -1
    return psi > 1 && psi < q && pow_mod(psi, n, q) == q - 1;
}

uint64_t min_primitive_root_2n(uint64_t q, uint32_t n) {
    const uint64_t two_n = 2ull * n;
    if (q < 3 || (q - 1) % two_n) return 0;
    const uint64_t cofactor = (q - 1) / two_n;
    uint64_t some_root = 0;
    for (uint64_t g = 2; g < q && g < 100000; ++g) {
        uint64_t r = pow_mod(g, cofactor, q);
        if (is_primitive_root_2n(r, q, n)) { some_root = r; break; }
    }
    if (!some_root) return 0;
    // the primitive 2n-th roots are exactly the odd powers of any one of them
    const uint64_t sq = mul_mod(some_root, some_root, q);
    uint64_t least = some_root, walk = some_root;
    for (uint32_t k = 1; k < n; ++k) {
        walk = mul_mod(walk, sq, q);
        if (walk < least) least = walk;
    }
    return least;
}

uint64_t shoup_quotient(uint64_t w, uint64_t q) { return (uint64_t)(((u128)w << 64) / q); }

void power_tables_bitrev(uint64_t q, uint64_t base, uint32_t n, uint64_t* tw, uint64_t* pre) {
    const int lg = log2u(n);
    std::vector<uint64_t> pw(n);
    pw[0] = 1 % q;
    for (uint32_t i = 1; i < n; ++i) pw[i] = mul_mod(pw[i - 1], base, q);
    for (uint32_t j = 0; j < n; ++j) {
        tw[j] = pw[bit_reverse(j, lg)];
        pre[j] = shoup_quotient(tw[j], q);
    }
}

}  // namespace agx
