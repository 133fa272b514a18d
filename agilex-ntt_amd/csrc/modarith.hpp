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

// a*b mod q in [0,q) for a, b < q < 2^62, and also for a, b < 4q when q <= 2^60 (then a*b < 2^124 and
// the quotient < 16q <= 2^64 still fit).  Quotient estimate is low by at most 3.
__device__ __forceinline__ uint64_t mul_mod_barrett(uint64_t a, uint64_t b, const barrett128& k) {
    const uint64_t lo = a * b, hi = __umul64hi(a, b);
    const uint64_t est = hi * k.mu_hi + __umul64hi(hi, k.mu_lo) + __umul64hi(lo, k.mu_hi);
    uint64_t r = lo - est * k.q;  // [0,4q)
    r = csub(r, k.q << 1);
    return csub(r, k.q);
}


// ---------------------------------------------------------------------------------------
// Hand-selected instruction forms for the throughput kernels.
//
// Box calibration (profiles/r01_microbench_box_calibration.txt): on gfx950 every VOP3 integer
// op (v_mad_u64_u32, v_mul_lo/hi_u32, v_add3, v_lshl_add_u64, carry adds) issues at about half
// the rate of a plain VOP2 add, so a butterfly's cost is its instruction COUNT.  hipcc lowers
// the portable butterfly above to ~32 VALU instructions (v_mul_lo + v_add3 cross terms,
// v_mov pairs to zero-extend 32-bit partial products, 64-bit compare + 2 selects); the forms
// below need about 20 (fast) / 27 (exact).  They stay in C++ (two opaque values steer hipcc's
// instruction selection) because hipcc pads every inline-asm instruction with an s_nop.
// ---------------------------------------------------------------------------------------

// a*b + c with a,b 32-bit: hipcc selects v_mad_u64_u32 for this shape
__device__ __forceinline__ uint64_t mad64(uint32_t a, uint32_t b, uint64_t c) { return (uint64_t)a * b + c; }
__device__ __forceinline__ uint64_t mul64(uint32_t a, uint32_t b) { return (uint64_t)a * b; }

// The value 1 in a form the optimiser cannot see through: c + zext(a) is then one
// v_mad_u64_u32 (a * 1 + c) instead of a v_mov (to build the {a,0} pair) plus a 64-bit add.
template <int TAG>   // identical asm statements would be merged: give each value its own text
__device__ __forceinline__ uint32_t opaque_one() {
    uint32_t one;
    if constexpr (TAG == 0) asm("s_mov_b32 %0, 1" : "=s"(one));
    else asm("s_movk_i32 %0, 0x1" : "=s"(one));
    return one;
}
// Two such adds in one sum must use DIFFERENT opaque ones, or hipcc factors them back into
// (a + b) * one and rebuilds the register pairs this is meant to avoid.
__device__ __forceinline__ uint64_t add64_32(uint64_t c, uint32_t a, uint32_t one) { return (uint64_t)a * one + c; }

// hide how a wave-uniform 64-bit constant was derived (x + opaque(-m) stays ONE v_lshl_add_u64
// instead of becoming a v_sub_co/v_subb pair)
__device__ __forceinline__ uint64_t opaque_sgpr64(uint64_t v) {
    asm("" : "+s"(v));
    return v;
}

// Keeps all 64 bits of v live, so a multiply-add chain of which only the low word is used is
// not demoted to v_mul_lo_u32 + v_add3_u32 pairs (two instructions per term instead of one).
__device__ __forceinline__ uint64_t keep64(uint64_t v) {
    asm("" : "+v"(v));
    return v;
}

// per-prime constants of the throughput kernels (wave-uniform: live in SGPRs)
struct bf_consts {
    uint64_t q;
    uint64_t nq;     // 2^64 - q
    uint64_t m;      // lazy-range step: 2q (exact arithmetic) or 4q (fast arithmetic)
    uint64_t nm;     // 2^64 - m
    uint32_t one_a, one_b;  // two separate opaque_one() values (see add64_32)
    float est_inv;          // slightly below 2^32 / q, or 0 when q < 2^58 (reduce_final_est)
};

// x in [0,2m) -> x - (x >= m ? m : 0) through the sign of x - m; needs 2m <= 2^63... see callers
__device__ __forceinline__ uint64_t csub_sign(uint64_t x, const bf_consts& k) {
    const uint64_t d = x + k.nm;
    const uint32_t neg = (uint32_t)((int32_t)(uint32_t)(d >> 32) >> 31);  // all ones when x < m
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    u32x2 add;
    add.x = neg & (uint32_t)k.m;
    add.y = neg & (uint32_t)(k.m >> 32);
    return d + __builtin_bit_cast(uint64_t, add);
}

// x' = tx + w*y - c*q (mod 2^64), the whole right-hand side in 6 multiply-adds + 1 add:
// the low words accumulate straight onto tx, the four cross products only matter mod 2^32
// VAR (A/B in tools/microbench pwr, energy per butterfly): 0 = the four cross products as one v_mad_u64_u32 chain (shipped);
// 1 = as 32-bit products (v_mul_lo_u32 + adds: more instructions, narrower results)
template <int VAR = 0>
__device__ __forceinline__ uint64_t fold_product(uint64_t tx, uint64_t y, uint64_t w, uint64_t c, const bf_consts& k) {
    const uint32_t y0 = (uint32_t)y, y1 = (uint32_t)(y >> 32), w0 = (uint32_t)w, w1 = (uint32_t)(w >> 32);
    const uint32_t c0 = (uint32_t)c, c1 = (uint32_t)(c >> 32), nq0 = (uint32_t)k.nq, nq1 = (uint32_t)(k.nq >> 32);
    uint64_t acc = mad64(w0, y0, tx);
    acc = mad64(nq0, c0, acc);
    if constexpr (VAR == 1) {
        const uint32_t z32 = w0 * y1 + w1 * y0 + nq1 * c0 + nq0 * c1;
        typedef uint32_t u32x2v __attribute__((ext_vector_type(2)));
        u32x2v av = __builtin_bit_cast(u32x2v, acc);
        av.y += z32;
        return __builtin_bit_cast(uint64_t, av);
    }
    uint64_t z = mul64(w0, y1);
    z = mad64(w1, y0, z);
    z = mad64(nq1, c0, z);
    z = keep64(mad64(nq0, c1, z));
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    u32x2 a = __builtin_bit_cast(u32x2, acc);
    a.y += (uint32_t)z;
    return __builtin_bit_cast(uint64_t, a);
}

// EXACT arithmetic: the reference's butterfly (src/kernel/ntt.cpp:331-369) value for value --
// c = floor(y*w'/2^64) exactly, coefficients in [0,4q), any q < 2^62.
__device__ __forceinline__ void ct_butterfly_exact(uint64_t& x, uint64_t& y, uint64_t w, uint64_t wp, const bf_consts& k) {
    const uint32_t y0 = (uint32_t)y, y1 = (uint32_t)(y >> 32), p0 = (uint32_t)wp, p1 = (uint32_t)(wp >> 32);
    const uint64_t tx = csub(x, k.m);                                       // m = 2q
    const uint32_t t = __umulhi(y0, p0);
    const uint64_t m1 = add64_32(mul64(p1, y0), t, k.one_a);                 // y0*p1 + hi(y0*p0)
    const uint64_t m2 = add64_32(mul64(p0, y1), (uint32_t)m1, k.one_b);      // y1*p0 + lo(m1)
    uint64_t c = add64_32(mul64(p1, y1), (uint32_t)(m1 >> 32), k.one_a);
    c = add64_32(c, (uint32_t)(m2 >> 32), k.one_b);
    const uint64_t xn = fold_product(tx, y, w, c, k);
    y = (tx << 1) + k.m - xn;                                               // tx + 2q - Q
    x = xn;
}

// FAST arithmetic for q <= 2^61: quotient estimate c~ = y1*p1 + hi(y0*p1) + hi(y1*p0) is low by
// at most 2, so Q~ = w*y - c~*q lies in [0,4q); coefficients are kept in [0,8q) with m = 4q:
//   tx = x - (x >= 4q ? 4q : 0) < 4q,  x' = tx + Q~ < 8q,  y' = tx + 4q - Q~ in (0,8q).
// Same residues as the exact form at every stage, so the fully reduced outputs are identical.
// x in [0,2m) -> x - (x >= m ? m : 0) by selecting on the sign of x - m (compare + two selects)
__device__ __forceinline__ uint64_t csub_select(uint64_t x, const bf_consts& k) {
    const uint64_t d = x + k.nm;
    return (int64_t)d < 0 ? x : d;
}

template <bool SEL = false>
__device__ __forceinline__ void ct_butterfly_fast(uint64_t& x, uint64_t& y, uint64_t w, uint64_t wp, const bf_consts& k) {
    const uint32_t y0 = (uint32_t)y, y1 = (uint32_t)(y >> 32), p0 = (uint32_t)wp, p1 = (uint32_t)(wp >> 32);
    const uint64_t tx = SEL ? csub_select(x, k) : csub_sign(x, k);          // m = 4q <= 2^63
    uint64_t c = mad64(p1, y1, (uint64_t)__umulhi(y0, p1));                  // one v_mov builds the {h,0} pair
    c = add64_32(c, __umulhi(y1, p0), k.one_b);
    const uint64_t xn = fold_product(tx, y, w, c, k);
    y = (tx << 1) + k.m - xn;                                               // tx + 4q - Q~
    x = xn;
}

// [0,2m) -> [0,q) at the last stage (m = 2q: two steps; m = 4q: three).  The fast form may use
// the sign trick for every step (all values < 2^63 after the first one); the exact form must
// compare (q may exceed 2^61).
__device__ __forceinline__ uint64_t csub_sign_c(uint64_t x, uint64_t m, uint64_t nm) {
    const uint64_t d = x + nm;
    const uint32_t neg = (uint32_t)((int32_t)(uint32_t)(d >> 32) >> 31);
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    u32x2 add;
    add.x = neg & (uint32_t)m;
    add.y = neg & (uint32_t)(m >> 32);
    return d + __builtin_bit_cast(uint64_t, add);
}

struct final_consts {   // fast forms only
    uint64_t q2, nq2, q1, nq1;
    uint64_t q8, nq8;   // 8q and its negation: the conditional-subtract step of the 16q-lazy form
};

template <bool FAST, bool SEL = false>
__device__ __forceinline__ uint64_t reduce_final(uint64_t v, const bf_consts& k, const final_consts& f, bool lazy_out = false) {
    if constexpr (FAST && SEL) {
        const uint64_t d4 = v + k.nm;            // [0,8q) -> [0,4q)
        v = (int64_t)d4 < 0 ? v : d4;
        if (lazy_out) return v;
        const uint64_t d2 = v + f.nq2;
        v = (int64_t)d2 < 0 ? v : d2;
        const uint64_t d1 = v + f.nq1;
        return (int64_t)d1 < 0 ? v : d1;
    } else if constexpr (FAST) {
        v = csub_sign(v, k);                 // [0,8q) -> [0,4q)
        if (lazy_out) return v;
        v = csub_sign_c(v, f.q2, f.nq2);     // -> [0,2q)
        return csub_sign_c(v, f.q1, f.nq1);  // -> [0,q)
    } else {
        if (lazy_out) return v;              // already in [0,4q)
        v = csub(v, k.q << 1);
        return csub(v, k.q);
    }
}

// 16q-LAZY arithmetic for q <= 2^60 (16q <= 2^64): with twice the headroom of the fast form the
// conditional subtract is only needed when the running bound would pass 16q, which is on 5 of 12
// stages at n = 4096 (lazy16_schedule below) instead of on every stage.  A skipped stage maps a
// bound B to B + 4q (x' = x + Q~, y' = x + 4q - Q~, Q~ < 4q); a subtracting stage uses the step 8q:
// tx = x - [x >= 8q] 8q < max(8q, B - 8q).  Same residues as the other forms at every stage.
struct lazy16_schedule {
    // bound (in units of q) on the coefficients entering stage s, and whether stage s subtracts;
    // inputs are < 4q (the reference's input range, src/kernel/ntt.cpp:331-332)
    static constexpr int bound_in(int s) {
        int b = 4;
        for (int i = 0; i < s; ++i) b = next(b);
        return b;
    }
    static constexpr bool subtracts(int s) { return bound_in(s) + 4 > 16; }
    static constexpr int next(int b) { return (b + 4 > 16 ? (b - 8 > 8 ? b - 8 : 8) : b) + 4; }
};

// TAIL-FREE variant of the schedule for a transform of `total` stages: the same number of subtracting stages, but
// placed on the stages whose parity matches `total` (from stage 2 on), so that the LAST stage never subtracts and its
// outputs are simply below 16q; reduce_final_est then brings them home with one quotient estimate instead of
// 2 + 2x3 conditional subtracts per butterfly.
struct lazy16_tailfree {
    static constexpr bool subtracts(int s, int total) { return s >= 2 && (((s ^ total) & 1) == 0); }
    static constexpr int bound_in(int s, int total) {
        int b = 4;
        for (int i = 0; i < s; ++i) {
            if (subtracts(i, total)) b = b - 8 > 8 ? b - 8 : 8;
            b += 4;
        }
        return b;
    }
    // every stage's outputs stay within 16q, and the last stage does not subtract
    static constexpr bool valid(int total) {
        for (int s = 0; s < total; ++s) {
            const int in = bound_in(s, total);
            if (in > 16) return false;
            if (!subtracts(s, total) && in + 4 > 16) return false;
        }
        return total >= 3 && !subtracts(total - 1, total);
    }
};

// select-based conditional subtract with explicit constants
__device__ __forceinline__ uint64_t csub_select_c(uint64_t x, uint64_t nm) {
    const uint64_t d = x + nm;
    return (int64_t)d < 0 ? x : d;
}

// LAST: the transform's final stage.  There tx is brought below 4q (one more subtract), so the
// outputs are < 8q and the final reduction needs three steps per coefficient instead of four.
template <bool SEL, bool DO_CSUB, bool LAST = false, int VAR = 0>
__device__ __forceinline__ void ct_butterfly_lazy16(uint64_t& x, uint64_t& y, uint64_t w, uint64_t wp, const bf_consts& k,
                                                    const final_consts& f) {
    const uint32_t y0 = (uint32_t)y, y1 = (uint32_t)(y >> 32), p0 = (uint32_t)wp, p1 = (uint32_t)(wp >> 32);
    uint64_t tx = x;
    if constexpr (DO_CSUB || LAST) {
        if constexpr (SEL) tx = csub_select_c(tx, f.nq8);
        else tx = csub_sign_c(tx, f.q8, f.nq8);
    }
    if constexpr (LAST) {
        if constexpr (SEL) tx = csub_select_c(tx, k.nm);
        else tx = csub_sign(tx, k);
    }
    uint64_t c = mad64(p1, y1, (uint64_t)__umulhi(y0, p1));                  // one v_mov builds the {h,0} pair
    c = add64_32(c, __umulhi(y1, p0), k.one_b);
    const uint64_t xn = fold_product<VAR>(tx, y, w, c, k);
    y = (tx << 1) + k.m - xn;                                               // tx + 4q - Q~
    x = xn;
}

// after a LAST butterfly: [0,8q) -> [0,q), or only to [0,4q) when the caller asked for lazy outputs
template <bool SEL>
__device__ __forceinline__ uint64_t reduce_final_lazy16(uint64_t v, const bf_consts& k, const final_consts& f, bool lazy_out) {
    if constexpr (SEL) {
        v = csub_select_c(v, k.nm);          // 4q
        if (lazy_out) return v;
        v = csub_select_c(v, f.nq2);
        return csub_select_c(v, f.nq1);
    } else {
        v = csub_sign(v, k);
        if (lazy_out) return v;
        v = csub_sign_c(v, f.q2, f.nq2);
        return csub_sign_c(v, f.q1, f.nq1);
    }
}

// [0,16q) -> [0,q) (or only [0,2q) for lazy outputs) after a tail-free 16q-lazy transform.  For q >= 2^58 the quotient
// floor(v/q) <= 15 is estimated from the top word in single precision -- est_inv is a float slightly BELOW 2^32/q, so
// the estimate k' is floor(v/q) or one less (never above: the remainder stays non-negative) -- then v - k' q in [0,2q)
// and one conditional subtract finishes: 3 conversions/multiplies + 2 integer multiplies + 1 subtract step instead of
// four subtract steps.  Smaller moduli (top word too short for the estimate) take the four steps.
// Contract with the host (agx_ntt.cpp build_plan): est_inv != 0 only for q in [2^58, 2^60], so v < 16q <= 2^64 holds wherever the
// estimate path runs; the kernels that call this are the 16q-lazy ones (rb2_frame::EST requires LAZY16, legal for q <= 2^60 only).
// USE_EST: 1 = the caller has checked k.est_inv != 0 (quotient estimate), 0 = it has checked it is 0 (four steps), -1 = decide here
template <bool SEL, int USE_EST = -1>
__device__ __forceinline__ uint64_t reduce_final_est(uint64_t v, const bf_consts& k, const final_consts& f, bool lazy_out) {
    static_assert(SEL, "select-based conditional subtract only");
    if (USE_EST == 1 || (USE_EST == -1 && k.est_inv != 0.0f)) {   // wave-uniform
        const uint32_t kq = (uint32_t)((float)(uint32_t)(v >> 32) * k.est_inv);
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        u32x2 r = __builtin_bit_cast(u32x2, mad64(kq, (uint32_t)k.nq, v));      // v + k' * (2^64 - q), low word of -q
        r.y += kq * (uint32_t)(k.nq >> 32);                                      // ... and its high word
        const uint64_t r64 = __builtin_bit_cast(uint64_t, r);
        return lazy_out ? r64 : csub_select_c(r64, f.nq1);
    }
    v = csub_select_c(v, f.nq8);
    v = csub_select_c(v, k.nm);          // 4q
    if (lazy_out) return v;
    v = csub_select_c(v, f.nq2);
    return csub_select_c(v, f.nq1);
}

// w*d - c*q (mod 2^64) with the quotient estimate of the chosen arithmetic:
// exact -> [0,2q), fast -> [0,4q); d may be any 64-bit value
template <bool FAST>
__device__ __forceinline__ uint64_t mul_shoup_form(uint64_t d, uint64_t w, uint64_t wp, const bf_consts& k) {
    const uint32_t d0 = (uint32_t)d, d1 = (uint32_t)(d >> 32), p0 = (uint32_t)wp, p1 = (uint32_t)(wp >> 32);
    uint64_t c;
    if constexpr (FAST) {
        c = mad64(p1, d1, (uint64_t)__umulhi(d0, p1));
        c = add64_32(c, __umulhi(d1, p0), k.one_b);
    } else {
        const uint32_t t = __umulhi(d0, p0);
        const uint64_t m1 = add64_32(mul64(p1, d0), t, k.one_a);
        const uint64_t m2 = add64_32(mul64(p0, d1), (uint32_t)m1, k.one_b);
        c = add64_32(mul64(p1, d1), (uint32_t)(m1 >> 32), k.one_a);
        c = add64_32(c, (uint32_t)(m2 >> 32), k.one_b);
    }
    return fold_product(0, d, w, c, k);
}

// Gentleman-Sande butterfly of the inverse transform, coefficients in [0,m)
// (m = 2q exact / 4q fast): x' = x + y - [x+y >= m] m,  y' = w (x - y + m) lazily reduced.
template <bool FAST, bool SEL = false>
__device__ __forceinline__ void gs_butterfly_form(uint64_t& x, uint64_t& y, uint64_t w, uint64_t wp, const bf_consts& k) {
    const uint64_t s = x + y;
    const uint64_t d = x + k.m - y;
    if constexpr (FAST) x = SEL ? csub_select(s, k) : csub_sign(s, k);
    else x = csub(s, k.m);
    y = mul_shoup_form<FAST>(d, w, wp, k);
}

// last inverse stage with n^-1 folded in: x' = (x + y) n^-1, y' = (x - y) w n^-1
template <bool FAST>
__device__ __forceinline__ void gs_last_form(uint64_t& x, uint64_t& y, uint64_t ninv, uint64_t ninv_p,
                                             uint64_t w1n, uint64_t w1n_p, const bf_consts& k) {
    const uint64_t s = x + y;
    const uint64_t d = x + k.m - y;
    x = mul_shoup_form<FAST>(s, ninv, ninv_p, k);
    y = mul_shoup_form<FAST>(d, w1n, w1n_p, k);
}

// 16q-lazy Gentleman-Sande butterflies (q <= 2^60, so 16q <= 2^64).  y' = w d is below 4q whatever d is, so
// only the sums grow: B is the compile-time bound, in units of q, that both inputs share (4, 8 or 16 -- a
// butterfly's two inputs always have the same history).  Inputs at 16q are first brought below 8q; then
// x' = x + y < 2b q <= 16q and d = x - y + b q in (0, 16q).  The caller tracks the bounds (gs_bound).
template <bool SEL>
__device__ __forceinline__ uint64_t csub_8q(uint64_t v, const final_consts& f) {
    return SEL ? csub_select_c(v, f.nq8) : csub_sign_c(v, f.q8, f.nq8);
}
// bound of register r entering stage s of a pass whose registers all start below B0*q: the X output of a
// stage doubles its (reduced) input bound, the Y output is below 4q
constexpr int gs_bound(int B0, int s, int r) {
    if (s == 0) return B0;
    if ((r >> (s - 1)) & 1) return 4;
    const int b = gs_bound(B0, s - 1, r);
    return 2 * (b == 16 ? 8 : b);
}
template <int B, bool SEL>
__device__ __forceinline__ void gs_butterfly_lazy16(uint64_t& x, uint64_t& y, uint64_t w, uint64_t wp, const bf_consts& k, const final_consts& f) {
    static_assert(B == 4 || B == 8 || B == 16, "bounds are 4q, 8q or 16q");
    if constexpr (B == 16) {
        x = csub_8q<SEL>(x, f);
        y = csub_8q<SEL>(y, f);
    }
    const uint64_t s = x + y;
    const uint64_t d = x + (B == 4 ? k.m : f.q8) - y;
    y = mul_shoup_form<true>(d, w, wp, k);
    x = s;
}
template <int B, bool SEL>
__device__ __forceinline__ void gs_last_lazy16(uint64_t& x, uint64_t& y, uint64_t ninv, uint64_t ninv_p, uint64_t w1n, uint64_t w1n_p,
                                               const bf_consts& k, const final_consts& f) {
    if constexpr (B == 16) {
        x = csub_8q<SEL>(x, f);
        y = csub_8q<SEL>(y, f);
    }
    const uint64_t s = x + y;
    const uint64_t d = x + (B == 4 ? k.m : f.q8) - y;
    x = mul_shoup_form<true>(s, ninv, ninv_p, k);
    y = mul_shoup_form<true>(d, w1n, w1n_p, k);
}

// [0,m) -> [0,q) after the inverse transform
template <bool FAST, bool SEL = false>
__device__ __forceinline__ uint64_t reduce_final_inv(uint64_t v, const bf_consts& k, const final_consts& f) {
    if constexpr (FAST && SEL) {
        v = csub_select_c(v, f.nq2);
        return csub_select_c(v, f.nq1);
    } else if constexpr (FAST) {
        v = csub_sign_c(v, f.q2, f.nq2);
        return csub_sign_c(v, f.q1, f.nq1);
    } else {
        return csub(v, k.q);
    }
}

}  // namespace agx
