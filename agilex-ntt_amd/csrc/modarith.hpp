// modarith.hpp -- 64-bit lazy modular arithmetic for gfx950 (device side).
//
// Value ranges follow the reference butterfly (src/kernel/ntt.cpp:302-369): coefficients live
// in [0,4q) between stages, q < 2^62 so nothing overflows 64 bits, and only the last stage
// reduces to [0,q) (src/kernel/ntt.cpp:377-394).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace agx {

// {w, w'} with w' = floor(w * 2^64 / q): one 16-byte load fetches both
typedef ulonglong2 twpair;

// w*y - floor(w'*y / 2^64) * q  (mod 2^64)  in [0,2q) for any 64-bit y   (ntt.cpp:344-363)
__device__ __forceinline__ uint64_t mul_shoup_lazy(uint64_t y, uint64_t w, uint64_t wp, uint64_t q) {
    const uint64_t c = __umul64hi(y, wp);
    return w * y - c * q;
}

__device__ __forceinline__ uint64_t csub(uint64_t v, uint64_t m) { return v >= m ? v - m : v; }

// [0,4q) -> [0,q)   (ntt.cpp:377-394)
__device__ __forceinline__ uint64_t reduce_4q(uint64_t v, uint64_t q, uint64_t q2) {
    return csub(csub(v, q2), q);
}

// Cooley-Tukey (decimation in time) Harvey butterfly: x,y in [0,4q) -> [0,4q)
__device__ __forceinline__ void ct_butterfly(uint64_t& x, uint64_t& y, uint64_t w, uint64_t wp,
                                             uint64_t q, uint64_t q2) {
    const uint64_t tx = csub(x, q2);                  // ntt.cpp:331-332
    const uint64_t qv = mul_shoup_lazy(y, w, wp, q);  // ntt.cpp:344-363
    x = tx + qv;                                      // ntt.cpp:368
    y = tx + q2 - qv;                                 // ntt.cpp:369
}

// Gentleman-Sande (decimation in frequency) butterfly for the inverse: x,y in [0,2q) -> [0,2q)
__device__ __forceinline__ void gs_butterfly(uint64_t& x, uint64_t& y, uint64_t w, uint64_t wp,
                                             uint64_t q, uint64_t q2) {
    const uint64_t s = csub(x + y, q2);
    const uint64_t d = x + q2 - y;  // (0,4q)
    x = s;
    y = mul_shoup_lazy(d, w, wp, q);
}

// per-prime constants for a*b mod q with both operands variable
struct barrett128 {
    uint64_t q;
    uint64_t mu_hi, mu_lo;  // floor(2^128 / q)
};

// a, b < q < 2^62 -> a*b mod q in [0,q).  Quotient estimate is low by at most 3.
__device__ __forceinline__ uint64_t mul_mod_barrett(uint64_t a, uint64_t b, const barrett128& k) {
    const uint64_t lo = a * b, hi = __umul64hi(a, b);
    const uint64_t est = hi * k.mu_hi + __umul64hi(hi, k.mu_lo) + __umul64hi(lo, k.mu_hi);
    uint64_t r = lo - est * k.q;  // [0,4q)
    r = csub(r, k.q << 1);
    return csub(r, k.q);
}

}  // namespace agx
