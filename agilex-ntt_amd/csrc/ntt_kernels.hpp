// ntt_kernels.hpp -- internal launch interface between the C ABI (agx_ntt.cpp) and the
// gfx950 kernels (ntt_kernels.hip).  Not installed; the public surface is include/agx_ntt.h.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <vector>

namespace agx {

struct prime_consts {           // one per prime, device array
    uint64_t q;
    uint64_t mu_hi, mu_lo;      // floor(2^128 / q)           (pointwise multiply)
    uint64_t n_inv, n_inv_p;    // n^-1 mod q and its precomputed quotient      (inverse, last stage)
    uint64_t w1n, w1n_p;        // inv_twiddle[1] * n^-1 mod q and its quotient  (inverse, last stage)
    uint64_t est;               // low word: float slightly below 2^32 / q for reduce_final_est, 0 when q < 2^58
};

// geometry of one register-blocked configuration (compile-time L, R mirrored at run time)
struct regblock_layout {
    int log_n = 0;       // whole transform = the frame one workgroup (or one wave's group of lanes) holds
    int r = 0;           // log2 coefficients per thread
    int config_id = -1;  // entry of the kernel registry in ntt_kernels.hip
    uint32_t pairs_per_prime = 0;  // table length per prime, in {w,w'} pairs
    uint32_t min_frames = 0;       // forward companions: only launches of at least this many frames take this layout
    bool valid() const { return r > 0; }
};

// device-side view of a plan
struct plan_view {
    uint32_t n = 0, log_n = 0, num_primes = 0;
    const prime_consts* consts = nullptr;  // [P]
    const ulonglong2* tw = nullptr;        // [P][n] {w,w'} natural index   (forward)
    const ulonglong2* itw = nullptr;       // [P][n] {w,w'} natural index   (inverse) or null
    regblock_layout rb;                    // forward register-blocked layout
    const ulonglong2* tw_rb = nullptr;     // [P][rb.pairs_per_prime]
    const ulonglong2* itw_rb = nullptr;    // same layout from the inverse tables, or null
    // Kernels that hand out frames through a counter ask for a {next frame, retired workgroups} pair of the plan HERE, at launch time and
    // only if they need one: the pair is keyed by the stream (launches on one stream serialise, so they may share a pair; the last
    // workgroup out zeroes it).  nullptr = no pair can be proven free (too many distinct streams): take the stateless fixed-stride form.
    uint32_t* (*ticket_for)(void* ctx, hipStream_t s) = nullptr;
    void (*ticket_launched)(void* ctx, hipStream_t s, uint32_t* pair) = nullptr;      // right behind the launch that uses `pair`: marks when the pair will be idle again
    void* ticket_ctx = nullptr;
    uint32_t* ticket(hipStream_t s) const { return ticket_for ? ticket_for(ticket_ctx, s) : nullptr; }
    void ticket_done(hipStream_t s, uint32_t* pair) const { if (ticket_launched) ticket_launched(ticket_ctx, s, pair); }
};

struct frame_layout {
    uint64_t batch;
    int64_t prime_stride, poly_stride;  // in elements
    bool lazy_out = false;              // forward: results may stay in [0,4q) (kernels are free to reduce fully)
};

// host-side construction of the register-blocked forward table for one prime from its
// natural-index tables; appends rb.pairs_per_prime pairs to `out`
// config_id -1: tuned default for n; arith_level: 0 exact only, 1 every modulus <= 2^61 (fast form legal),
// 2 every modulus <= 2^60 (16q-lazy form legal)
// narrow_level: 0 some modulus >= 2^31; 1 every modulus < 2^31; 2 every modulus < 2^30 (the 32-bit kernels of rb32_kernels.hpp; they
// also need arith_level >= 1, i.e. tables that honour the precon contract)
regblock_layout regblock_choose(uint32_t n, int config_id, int arith_level, int narrow_level = 0);
// forward-only layout that serves the forward calls of a plan whose tuned default is `main` (a second kernel shape that is faster for
// the forward transform only), or an invalid layout
regblock_layout regblock_forward_companion(const regblock_layout& main, uint32_t n, int arith_level, int narrow_level);
void regblock_build_table(const regblock_layout& rb, const uint64_t* tw, const uint64_t* pre, std::vector<ulonglong2>& out);

hipError_t kernels_init();  // one-time function attributes (large dynamic LDS)

hipError_t launch_forward_radix2(const plan_view& pv, const uint64_t* in, uint64_t* out, const frame_layout& fl, hipStream_t s);
hipError_t launch_inverse_radix2(const plan_view& pv, const uint64_t* in, uint64_t* out, const frame_layout& fl, hipStream_t s);
hipError_t launch_forward_regblock(const plan_view& pv, const uint64_t* in, uint64_t* out, const frame_layout& fl, hipStream_t s);
bool regblock_has_inverse(const regblock_layout& rb);
bool regblock_has_polymul(const regblock_layout& rb);   // fused NTT -> pointwise -> INTT in one kernel
// in2 != null: transforms the coefficient-wise product in * in2 (the pointwise step fused into the load)
hipError_t launch_inverse_regblock(const plan_view& pv, const uint64_t* in, const uint64_t* in2, uint64_t* out, const frame_layout& fl, hipStream_t s);
hipError_t launch_polymul_regblock(const plan_view& pv, const uint64_t* a, const uint64_t* b, uint64_t* c, const frame_layout& fl, hipStream_t s);
hipError_t launch_pointwise(const plan_view& pv, const uint64_t* a, const uint64_t* b, uint64_t* c, uint64_t batch, hipStream_t s);
hipError_t launch_fill(const plan_view& pv, uint64_t* out, uint64_t batch, uint64_t first_poly, uint64_t seed, hipStream_t s);

}  // namespace agx
