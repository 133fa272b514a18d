// wp_kernels.hpp -- "wave-packed" kernels for the small transform sizes, n = 2 ... 512 (n = 32 is the smallest size of the
// reference's own table, include/kernel/ntt.h:11-12, src/kernel/ntt.cpp:70-71; below it one LANE holds a whole frame).
//
// A frame of n = 2^L coefficients is held by T = 2^(L-R) lanes with 2^R coefficients each, T < 64: one WAVE carries 64 / T whole
// frames (n = 32 with R = 5: 64 frames, one per lane) and nothing in a transform ever crosses a wave, so these kernels contain
// no workgroup barrier at all -- every exchange between passes is an in-order LDS round trip of one wave (wave_lds_sync).  The
// butterfly passes themselves are rb2_frame's (64-bit forms) / rb32_frame's (32-bit forms, every modulus below 2^31): same
// arithmetic, same tables, same lazy ranges as the large sizes; pass 0's twiddles are wave-uniform (scalar loads into SGPRs),
// later passes read theirs per lane from a table of at most 16n bytes per prime (L1 / L2 resident).
//
// What is new here is the global side.  The frames of a wave are adjacent in memory (dense layout), so the wave moves its
// 64 * 2^R coefficients as ONE contiguous run, 512 bytes per load / store instruction ("lane layout": element lane + 64 k), and
// converts between that and the layouts the passes want through its private part of the LDS image:
//     lane layout  E = lane + 64 k                       (global accesses)
//     pass-0       E = fl n + tid + T r                  (fl = lane / T the lane's frame, tid = lane % T; forward in, inverse out)
//     last pass    E = lane 2^R + r                      (forward out, inverse in)
// All three are additive over disjoint bit fields, so with the padded image index E + (E >> PADS) every LDS access is
// (lane base) + (compile-time constant).  With T >= 16 the pass-0 side is accessed directly (128-byte runs per frame), which
// saves one LDS round trip.  Strided callers (poly_stride != n) take the same kernels with per-frame addresses.
//
// Included by reg_wp*.hip behind rb_kernels.hpp / rb32_kernels.hpp.
#pragma once

namespace agx {
namespace AGX_TU {

enum wp_layout { WP_LANE = 0, WP_PASS0 = 1, WP_LAST = 2 };

template <int L, int R>
struct wp_geom {
    static constexpr int C = 1 << R, T = 1 << (L - R), FPW = 64 / T, N = 1 << L;
    static_assert(L >= 1 && L <= 9 && R <= L && T < 64 && T >= 1, "wave-packed kernels: 2 <= n <= 512, fewer than 64 lanes per frame");
    static constexpr bool direct_pass0 = T >= 16;      // pass-0 layout accessed in global memory directly (runs of 8 T bytes per frame)
    // element (within the wave's 64 C coefficients) that `lane` holds in register r under layout LAY: lane part + register part
    template <int LAY>
    static __device__ __forceinline__ constexpr uint32_t lane_part(uint32_t lane) {
        if constexpr (LAY == WP_LANE) return lane;
        else if constexpr (LAY == WP_PASS0) return ((lane / T) << L) | (lane % T);
        else return lane << R;
    }
    template <int LAY>
    static __device__ __forceinline__ constexpr uint32_t reg_part(uint32_t r) {
        if constexpr (LAY == WP_LANE) return 64u * r;
        else if constexpr (LAY == WP_PASS0) return r << (L - R);
        else return r;
    }
};

// one wave's coefficients from layout FROM to layout TO through its part of the LDS image (64-bit words; with SPLIT the low and the
// high 32-bit words in turn through an image of half the size; wp_relayout32 below: the 32-bit kernels' image).
// The leading fence orders this wave's earlier LDS reads of the same words (an exchange, the previous operand's staging).
template <int L, int R, int FROM, int TO, int PADS, bool SPLIT>
__device__ __forceinline__ void wp_relayout(uint64_t (&x)[1 << R], void* wimg, uint32_t lane) {
    using G = wp_geom<L, R>;
    constexpr int C = G::C;
    auto img = [](uint32_t e) constexpr { return e + (e >> PADS); };
    const uint32_t a = img(G::template lane_part<FROM>(lane)), b = img(G::template lane_part<TO>(lane));
    wave_lds_sync();
    if constexpr (SPLIT) {
        uint32_t* w = static_cast<uint32_t*>(wimg);
        static_for<0, C>([&](auto Rr) { constexpr int r = Rr; w[a + img(G::template reg_part<FROM>(r))] = (uint32_t)x[r]; });
        wave_lds_sync();
        uint32_t lo[C];
        static_for<0, C>([&](auto Rr) { constexpr int r = Rr; lo[r] = w[b + img(G::template reg_part<TO>(r))]; });
        wave_lds_sync();
        static_for<0, C>([&](auto Rr) { constexpr int r = Rr; w[a + img(G::template reg_part<FROM>(r))] = (uint32_t)(x[r] >> 32); });
        wave_lds_sync();
        static_for<0, C>([&](auto Rr) { constexpr int r = Rr; x[r] = (uint64_t)lo[r] | ((uint64_t)w[b + img(G::template reg_part<TO>(r))] << 32); });
    } else {
        uint64_t* w = static_cast<uint64_t*>(wimg);
        static_for<0, C>([&](auto Rr) { constexpr int r = Rr; w[a + img(G::template reg_part<FROM>(r))] = x[r]; });
        wave_lds_sync();
        static_for<0, C>([&](auto Rr) { constexpr int r = Rr; x[r] = w[b + img(G::template reg_part<TO>(r))]; });
    }
}
template <int L, int R, int FROM, int TO>
__device__ __forceinline__ void wp_relayout32(uint32_t (&x)[1 << R], uint32_t* w, uint32_t lane) {
    using G = wp_geom<L, R>;
    constexpr int C = G::C;
    auto img = [](uint32_t e) constexpr { return e + (e >> 5); };
    const uint32_t a = img(G::template lane_part<FROM>(lane)), b = img(G::template lane_part<TO>(lane));
    wave_lds_sync();
    static_for<0, C>([&](auto Rr) { constexpr int r = Rr; w[a + img(G::template reg_part<FROM>(r))] = x[r]; });
    wave_lds_sync();
    static_for<0, C>([&](auto Rr) { constexpr int r = Rr; x[r] = w[b + img(G::template reg_part<TO>(r))]; });
}

// where a wave's frames live in global memory
struct wp_span {
    uint64_t f0;           // first frame of this wave
    uint32_t live;         // frames of this wave that exist (1 .. FPW); the others mirror frame f0 + live - 1 and are never stored
    int64_t pbase;         // prime offset (elements)
    int64_t poly_stride;
    bool fast;             // dense layout (poly_stride == n) and every frame of the wave exists: one contiguous run, no per-element checks
};

// element offset of register k of `lane` in layout LAY for the general case (strided frames, or a wave whose last frames do not
// exist: those are folded back onto the last one for loads and masked for stores)
template <int L, int R, int LAY>
__device__ __forceinline__ int64_t wp_offset(const wp_span& sp, uint32_t lane, uint32_t k, bool& alive) {
    using G = wp_geom<L, R>;
    const uint32_t E = G::template lane_part<LAY>(lane) + G::template reg_part<LAY>(k);
    uint32_t fl = E >> L;
    const uint32_t e = E & (G::N - 1);
    alive = fl < sp.live;
    if (!alive) fl = sp.live - 1;
    return sp.pbase + (int64_t)(sp.f0 + fl) * sp.poly_stride + (int64_t)e;
}

// W = uint64_t: whole coefficients; W = uint32_t: their low words only (the 32-bit tier-2 kernels: inputs are below 4q <= 2^32)
template <int L, int R, int LAY, bool NT, typename W>
__device__ __forceinline__ void wp_load(W (&v)[1 << R], const uint64_t* __restrict__ in, const wp_span& sp, uint32_t lane) {
    using G = wp_geom<L, R>;
    if (sp.fast) {      // wave-uniform: uniform base + lane part + compile-time register part
        const uint64_t* run = in + sp.pbase + (int64_t)(sp.f0 << L);
        const uint32_t lp = G::template lane_part<LAY>(lane);
        static_for<0, (1 << R)>([&](auto K) {
            constexpr uint32_t rp = G::template reg_part<LAY>((uint32_t)K);
            const W* p = reinterpret_cast<const W*>(run + lp + rp);
            v[K] = NT ? __builtin_nontemporal_load(p) : *p;
        });
        return;
    }
    static_for<0, (1 << R)>([&](auto K) {
        bool alive;
        const W* p = reinterpret_cast<const W*>(in + wp_offset<L, R, LAY>(sp, lane, (uint32_t)K, alive));
        v[K] = NT ? __builtin_nontemporal_load(p) : *p;
    });
}
template <int L, int R, int LAY, bool NT>
__device__ __forceinline__ void wp_store(const uint64_t (&v)[1 << R], uint64_t* __restrict__ out, const wp_span& sp, uint32_t lane) {
    using G = wp_geom<L, R>;
    if (sp.fast) {
        uint64_t* run = out + sp.pbase + (int64_t)(sp.f0 << L);
        const uint32_t lp = G::template lane_part<LAY>(lane);
        static_for<0, (1 << R)>([&](auto K) {
            constexpr uint32_t rp = G::template reg_part<LAY>((uint32_t)K);
            if constexpr (NT) __builtin_nontemporal_store(v[K], run + lp + rp);
            else run[lp + rp] = v[K];
        });
        return;
    }
    static_for<0, (1 << R)>([&](auto K) {
        bool alive;
        const int64_t o = wp_offset<L, R, LAY>(sp, lane, (uint32_t)K, alive);
        if (alive) {
            if constexpr (NT) __builtin_nontemporal_store(v[K], &out[o]);
            else out[o] = v[K];
        }
    });
}

#define AGX_WP_SPAN(frames)                                                                        \
    using G = wp_geom<L, R>;                                                                       \
    constexpr int C = G::C, T = G::T, FPW = G::FPW;                                                \
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;                              \
    wp_span sp;                                                                                    \
    sp.f0 = ((uint64_t)blockIdx.x * WPB + wave) * FPW;                                             \
    if (sp.f0 >= (frames)) return;        /* wave-uniform; the kernels have no workgroup barrier */ \
    sp.live = (uint32_t)((frames) - sp.f0 < (uint64_t)FPW ? (frames) - sp.f0 : (uint64_t)FPW);     \
    const uint32_t prime = blockIdx.y;                                                             \
    sp.pbase = (int64_t)prime * prime_stride;                                                      \
    sp.poly_stride = poly_stride;                                                                  \
    sp.fast = poly_stride == (int64_t)G::N && sp.live == (uint32_t)FPW

// ---- 64-bit arithmetic ------------------------------------------------------------------------------------------------------------
#define AGX_WP_FRAME                                                                               \
    using F = rb2_frame<L, R, (ARITH & 1) == 1, (ARITH >> 1)>;                                     \
    constexpr bool NTL = ((ARITH >> 1) & kOptNtLoad) != 0, NTS = ((ARITH >> 1) & kOptNtStore) != 0;  \
    [[maybe_unused]] constexpr bool FASTA = (ARITH & 1) == 1;                                                       \
    constexpr int PADS = F::PADS;                                                                  \
    unsigned char* wimg = agx_dyn_lds + (size_t)wave * wp_wave_image_bytes<L, R, ARITH>();         \
    F f;                                                                                           \
    f.tid = lane % T;                                                                              \
    f.slab = reinterpret_cast<uint64_t*>(wimg + (size_t)(lane / T) * F::image_bytes);              \
    const prime_consts pc = consts[prime];                                                         \
    f.init_consts(pc.q, pc.est)

// one wave's image: its 64 * 2^R coefficients under the padded index E + (E >> PADS) (= 64 / T frames of rb2_frame::image_bytes each once a
// frame has at least 2^PADS coefficients)
template <int L, int R, int ARITH>
constexpr size_t wp_wave_image_bytes() {
    using F = rb2_frame<L, R, (ARITH & 1) == 1, (ARITH >> 1)>;
    constexpr size_t words = ((size_t)64 << R) + (((size_t)64 << R) >> F::PADS);
    return words * (F::SPLIT ? 4 : 8);
}
template <int L, int R, int ARITH>
constexpr size_t wp_lds_bytes(int wpb) { return (size_t)wpb * wp_wave_image_bytes<L, R, ARITH>(); }

template <int L, int R, int WPB, int ARITH, int MINW>
__global__ void __launch_bounds__(64 * WPB, MINW)
fwd_wp(const uint64_t* __restrict__ in, uint64_t* __restrict__ out, const prime_consts* __restrict__ consts,
       const twpair* __restrict__ tw_rb, uint32_t pairs_per_prime, uint64_t frames, int64_t prime_stride, int64_t poly_stride, uint32_t lazy_out) {
    AGX_WP_SPAN(frames);
    AGX_WP_FRAME;
    f.lazy_out = lazy_out != 0;
    uint64_t x[C];
    if constexpr (G::direct_pass0) {
        wp_load<L, R, WP_PASS0, NTL>(x, in, sp, lane);
    } else {
        wp_load<L, R, WP_LANE, NTL>(x, in, sp, lane);
        wp_relayout<L, R, WP_LANE, WP_PASS0, PADS, F::SPLIT>(x, wimg, lane);
    }
    f.forward(x, tw_rb + (size_t)prime * pairs_per_prime);
    wp_relayout<L, R, WP_LAST, WP_LANE, PADS, F::SPLIT>(x, wimg, lane);
    wp_store<L, R, WP_LANE, NTS>(x, out, sp, lane);
}

// in2 != null: the coefficient-wise product in * in2 is taken while loading (the three-launch product's tail)
template <int L, int R, int WPB, int ARITH, int MINW>
__global__ void __launch_bounds__(64 * WPB, MINW)
inv_wp(const uint64_t* __restrict__ in, const uint64_t* __restrict__ in2, uint64_t* __restrict__ out, const prime_consts* __restrict__ consts,
       const twpair* __restrict__ itw_rb, uint32_t pairs_per_prime, uint64_t frames, int64_t prime_stride, int64_t poly_stride) {
    AGX_WP_SPAN(frames);
    AGX_WP_FRAME;
    const barrett128 bk{pc.q, pc.mu_hi, pc.mu_lo};
    uint64_t x[C];
    wp_load<L, R, WP_LANE, NTL>(x, in, sp, lane);
    if (in2) {      // wave-uniform
        uint64_t y[C];
        wp_load<L, R, WP_LANE, NTL>(y, in2, sp, lane);
#pragma unroll
        for (int r = 0; r < C; ++r) x[r] = mul_mod_barrett(reduce_4q(x[r], pc.q, pc.q << 1), reduce_4q(y[r], pc.q, pc.q << 1), bk);
    }
    if constexpr (!FASTA) {
#pragma unroll
        for (int r = 0; r < C; ++r) x[r] = csub(x[r], f.k.m);      // the exact form wants [0,2q); inputs may be below 4q
    }
    wp_relayout<L, R, WP_LANE, WP_LAST, PADS, F::SPLIT>(x, wimg, lane);
    f.inverse(x, itw_rb + (size_t)prime * pairs_per_prime, pc);
    if constexpr (G::direct_pass0) {
        wp_store<L, R, WP_PASS0, NTS>(x, out, sp, lane);
    } else {
        wp_relayout<L, R, WP_PASS0, WP_LANE, PADS, F::SPLIT>(x, wimg, lane);
        wp_store<L, R, WP_LANE, NTS>(x, out, sp, lane);
    }
}

// c = INTT(NTT(a) o NTT(b)) for the wave's frames in one launch: both forward results stay in registers (2^(R+1) 64-bit values
// per lane) in the last pass's layout, the product is taken there and the inverse starts from it.  A wave reads its a and b frames
// completely before it writes c, so c may alias either or both (squaring included).
template <int L, int R, int WPB, int ARITH, int MINW>
__global__ void __launch_bounds__(64 * WPB, MINW)
polymul_wp(const uint64_t* __restrict__ a, const uint64_t* __restrict__ b, uint64_t* __restrict__ c, const prime_consts* __restrict__ consts,
           const twpair* __restrict__ tw_rb, const twpair* __restrict__ itw_rb, uint32_t pairs_per_prime, uint64_t frames,
           int64_t prime_stride, int64_t poly_stride) {
    AGX_WP_SPAN(frames);
    AGX_WP_FRAME;
    const barrett128 bk{pc.q, pc.mu_hi, pc.mu_lo};
    f.lazy_out = F::LAZY16;      // the Barrett product takes operands in [0,4q) when q <= 2^60
    uint64_t xa[C], xb[C];
    auto load_pass0 = [&](uint64_t (&x)[C], const uint64_t* __restrict__ src) {
        if constexpr (G::direct_pass0) {
            wp_load<L, R, WP_PASS0, NTL>(x, src, sp, lane);
        } else {
            wp_load<L, R, WP_LANE, NTL>(x, src, sp, lane);
            wp_relayout<L, R, WP_LANE, WP_PASS0, PADS, F::SPLIT>(x, wimg, lane);
        }
    };
    load_pass0(xa, a);
    f.forward(xa, tw_rb + (size_t)prime * pairs_per_prime);
    asm volatile("" ::: "memory");      // b's loads stay behind NTT(a): holding them across it would cost 2^R more register pairs
    load_pass0(xb, b);
    {
        const twpair* tbl2 = tw_rb + (size_t)prime * pairs_per_prime;
        asm volatile("" : "+s"(tbl2));      // opaque: or NTT(a)'s table entries are kept (spilled) for reuse instead of re-read from L1
        wave_lds_sync();
        f.forward(xb, tbl2);
    }
#pragma unroll
    for (int r = 0; r < C; ++r) xa[r] = mul_mod_barrett(xa[r], xb[r], bk);
    wave_lds_sync();
    f.inverse(xa, itw_rb + (size_t)prime * pairs_per_prime, pc);
    if constexpr (G::direct_pass0) {
        wp_store<L, R, WP_PASS0, NTS>(xa, c, sp, lane);
    } else {
        wp_relayout<L, R, WP_PASS0, WP_LANE, PADS, F::SPLIT>(xa, wimg, lane);
        wp_store<L, R, WP_LANE, NTS>(xa, c, sp, lane);
    }
}

template <int L, int R, int WPB, int ARITH, int MINW>
hipError_t launch_wp_t(const plan_view& pv, const uint64_t* in, uint64_t* out, const frame_layout& fl, hipStream_t s) {
    constexpr int FPB = wp_geom<L, R>::FPW * WPB;
    dim3 grid((unsigned)((fl.batch + FPB - 1) / FPB), pv.num_primes);
    const size_t lds = wp_lds_bytes<L, R, ARITH>(WPB);
    hipLaunchKernelGGL((fwd_wp<L, R, WPB, ARITH, MINW>), grid, dim3(64 * WPB), lds, s, in, out, pv.consts, pv.tw_rb,
                       pv.rb.pairs_per_prime, fl.batch, fl.prime_stride, fl.poly_stride, (uint32_t)(fl.lazy_out ? 1 : 0));
    return hipGetLastError();
}
template <int L, int R, int WPB, int ARITH, int MINW>
hipError_t launch_inv_wp_t(const plan_view& pv, const uint64_t* in, const uint64_t* in2, uint64_t* out, const frame_layout& fl, hipStream_t s) {
    constexpr int FPB = wp_geom<L, R>::FPW * WPB;
    dim3 grid((unsigned)((fl.batch + FPB - 1) / FPB), pv.num_primes);
    const size_t lds = wp_lds_bytes<L, R, ARITH>(WPB);
    hipLaunchKernelGGL((inv_wp<L, R, WPB, ARITH, MINW>), grid, dim3(64 * WPB), lds, s, in, in2, out, pv.consts, pv.itw_rb,
                       pv.rb.pairs_per_prime, fl.batch, fl.prime_stride, fl.poly_stride);
    return hipGetLastError();
}
// the product keeps two frames in registers: 16 coefficients per lane and more get 4 waves per SIMD (128 VGPRs)
template <int R, int MINW>
constexpr int wp_mul_waves() { return R >= 4 ? 4 : (MINW > AGX_POLYMUL_MAXW ? AGX_POLYMUL_MAXW : MINW); }

template <int L, int R, int WPB, int ARITH, int MINW>
hipError_t launch_mul_wp_t(const plan_view& pv, const uint64_t* a, const uint64_t* b, uint64_t* c, const frame_layout& fl, hipStream_t s) {
    constexpr int FPB = wp_geom<L, R>::FPW * WPB;
    dim3 grid((unsigned)((fl.batch + FPB - 1) / FPB), pv.num_primes);
    const size_t lds = wp_lds_bytes<L, R, ARITH>(WPB);
    hipLaunchKernelGGL((polymul_wp<L, R, WPB, ARITH, wp_mul_waves<R, MINW>()>), grid, dim3(64 * WPB), lds, s, a, b, c,
                       pv.consts, pv.tw_rb, pv.itw_rb, pv.rb.pairs_per_prime, fl.batch, fl.prime_stride, fl.poly_stride);
    return hipGetLastError();
}
template <int L, int R, int WPB, int ARITH, int MINW>
hipError_t init_wp_t() {
    const int bytes = (int)wp_lds_bytes<L, R, ARITH>(WPB);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fwd_wp<L, R, WPB, ARITH, MINW>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&inv_wp<L, R, WPB, ARITH, MINW>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&polymul_wp<L, R, WPB, ARITH, wp_mul_waves<R, MINW>()>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    return e;
}

// registry entry: forward, inverse and the one-launch product of one (L, R) shape; WPB waves per workgroup (no barrier: the group
// only shares an LDS allocation), MINW waves per SIMD
template <int L, int R, int WPB, int ARITH, int MINW>
constexpr rb_entry make_entry_wp(int id) {
    rb_entry e{id, L, R, wp_geom<L, R>::FPW * WPB, MINW, (uint32_t)rb_geom<L, R>::table_pairs, wp_lds_bytes<L, R, ARITH>(WPB),
               &build_table_t<L, R>, &launch_wp_t<L, R, WPB, ARITH, MINW>, &init_wp_t<L, R, WPB, ARITH, MINW>, rb2_arith_level<ARITH>(),
               &launch_inv_wp_t<L, R, WPB, ARITH, MINW>, &launch_mul_wp_t<L, R, WPB, ARITH, MINW>};
    return e;
}


// ---- 32-bit arithmetic (every modulus of the plan below 2^31; rb32_kernels.hpp must have been included) ---------------------------
#ifdef AGX_WP_Q32
template <int R>
constexpr size_t wp32_wave_image_words() { return ((size_t)64 << R) + (((size_t)64 << R) >> 5); }      // one pad word per 32, as rb32_frame::img

#define AGX_WP_FRAME32                                                                             \
    using F = rb32_frame<L, R, TIER>;                                                              \
    uint32_t* wimg = reinterpret_cast<uint32_t*>(agx_dyn_lds) + (size_t)wave * wp32_wave_image_words<R>();  \
    F f;                                                                                           \
    f.tid = lane % T;                                                                              \
    f.slab = wimg + (size_t)(lane / T) * F::slab_words;                                            \
    const prime_consts pc = consts[prime];                                                         \
    f.a.init(pc.q);                                                                                \
    const tw32* tbl = reinterpret_cast<const tw32*>(tw_rb) + (size_t)prime * pairs_per_prime * 2

template <int L, int R>
constexpr size_t wp32_lds_bytes(int wpb) { return (size_t)wpb * wp32_wave_image_words<R>() * 4; }

// the wave's coefficients in layout LAY as 32-bit values in the transform's entry range (rb32_kernels.hpp: q32_arith::enter)
template <int L, int R, int LAY, int TIER, bool INVERSE>
__device__ __forceinline__ void wp_load32(uint32_t (&x)[1 << R], const uint64_t* __restrict__ in, const wp_span& sp, uint32_t lane, const q32_arith<TIER>& a) {
    if constexpr (TIER == 2) {      // inputs are below 4q <= 2^32: only the low words are needed (ntt.cpp:331-332 accepts [0,4q))
        wp_load<L, R, LAY, true>(x, in, sp, lane);
        if constexpr (INVERSE) static_for<0, (1 << R)>([&](auto K) { x[K] = a.template enter<true>((uint64_t)x[K]); });
    } else {
        uint64_t v[1 << R];
        wp_load<L, R, LAY, true>(v, in, sp, lane);
        static_for<0, (1 << R)>([&](auto K) { x[K] = a.template enter<INVERSE>(v[K]); });
    }
}
template <int L, int R, int LAY>
__device__ __forceinline__ void wp_store32(const uint32_t (&x)[1 << R], uint64_t* __restrict__ out, const wp_span& sp, uint32_t lane) {
    uint64_t v[1 << R];
    static_for<0, (1 << R)>([&](auto K) { v[K] = (uint64_t)x[K]; });
    wp_store<L, R, LAY, true>(v, out, sp, lane);
}

template <int L, int R, int WPB, int TIER, int MINW>
__global__ void __launch_bounds__(64 * WPB, MINW)
fwd_wp32(const uint64_t* __restrict__ in, uint64_t* __restrict__ out, const prime_consts* __restrict__ consts,
         const twpair* __restrict__ tw_rb, uint32_t pairs_per_prime, uint64_t frames, int64_t prime_stride, int64_t poly_stride, uint32_t lazy_out) {
    AGX_WP_SPAN(frames);
    AGX_WP_FRAME32;
    uint32_t x[C];
    if constexpr (G::direct_pass0) {
        wp_load32<L, R, WP_PASS0, TIER, false>(x, in, sp, lane, f.a);
    } else {
        wp_load32<L, R, WP_LANE, TIER, false>(x, in, sp, lane, f.a);
        wp_relayout32<L, R, WP_LANE, WP_PASS0>(x, wimg, lane);
    }
    f.forward(x, tbl, lazy_out != 0);
    wp_relayout32<L, R, WP_LAST, WP_LANE>(x, wimg, lane);
    wp_store32<L, R, WP_LANE>(x, out, sp, lane);
}

template <int L, int R, int WPB, int TIER, int MINW>
__global__ void __launch_bounds__(64 * WPB, MINW)
inv_wp32(const uint64_t* __restrict__ in, const uint64_t* __restrict__ in2, uint64_t* __restrict__ out, const prime_consts* __restrict__ consts,
         const twpair* __restrict__ tw_rb, uint32_t pairs_per_prime, uint64_t frames, int64_t prime_stride, int64_t poly_stride) {
    AGX_WP_SPAN(frames);
    AGX_WP_FRAME32;      // tw_rb = the plan's inverse table here
    uint32_t x[C];
    if (in2) {      // wave-uniform: the product fused into the load; its remainder is in the inverse's entry range already
        uint32_t y[C];
        wp_load32<L, R, WP_LANE, TIER, false>(x, in, sp, lane, f.a);
        wp_load32<L, R, WP_LANE, TIER, false>(y, in2, sp, lane, f.a);
        static_for<0, C>([&](auto K) { x[K] = f.a.mulmod(x[K], y[K], pc.mu_hi); });
    } else {
        wp_load32<L, R, WP_LANE, TIER, true>(x, in, sp, lane, f.a);
    }
    wp_relayout32<L, R, WP_LANE, WP_LAST>(x, wimg, lane);
    const tw32 ninv = make_uint2((uint32_t)pc.n_inv, (uint32_t)(pc.n_inv_p >> 32)), w1n = make_uint2((uint32_t)pc.w1n, (uint32_t)(pc.w1n_p >> 32));
    f.inverse(x, tbl, ninv, w1n);
    if constexpr (G::direct_pass0) {
        wp_store32<L, R, WP_PASS0>(x, out, sp, lane);
    } else {
        wp_relayout32<L, R, WP_PASS0, WP_LANE>(x, wimg, lane);
        wp_store32<L, R, WP_LANE>(x, out, sp, lane);
    }
}

template <int L, int R, int WPB, int TIER, int MINW>
__global__ void __launch_bounds__(64 * WPB, MINW)
polymul_wp32(const uint64_t* __restrict__ pa, const uint64_t* __restrict__ pb, uint64_t* __restrict__ pcout, const prime_consts* __restrict__ consts,
             const twpair* __restrict__ tw_rb, const twpair* __restrict__ itw_rb, uint32_t pairs_per_prime, uint64_t frames,
             int64_t prime_stride, int64_t poly_stride) {
    AGX_WP_SPAN(frames);
    AGX_WP_FRAME32;
    const tw32* itbl = reinterpret_cast<const tw32*>(itw_rb) + (size_t)prime * pairs_per_prime * 2;
    uint32_t xa[C], xb[C];
    auto load_pass0 = [&](uint32_t (&x)[C], const uint64_t* __restrict__ src) {
        if constexpr (G::direct_pass0) {
            wp_load32<L, R, WP_PASS0, TIER, false>(x, src, sp, lane, f.a);
        } else {
            wp_load32<L, R, WP_LANE, TIER, false>(x, src, sp, lane, f.a);
            wp_relayout32<L, R, WP_LANE, WP_PASS0>(x, wimg, lane);
        }
    };
    load_pass0(xa, pa);
    f.forward(xa, tbl, true);
    asm volatile("" ::: "memory");      // b's loads stay behind NTT(a)
    load_pass0(xb, pb);
    {
        const tw32* tbl2 = tbl;
        asm volatile("" : "+s"(tbl2));
        wave_lds_sync();
        f.forward(xb, tbl2, true);
    }
    static_for<0, C>([&](auto K) { xa[K] = f.a.mulmod(xa[K], xb[K], pc.mu_hi); });
    wave_lds_sync();
    const tw32 ninv = make_uint2((uint32_t)pc.n_inv, (uint32_t)(pc.n_inv_p >> 32)), w1n = make_uint2((uint32_t)pc.w1n, (uint32_t)(pc.w1n_p >> 32));
    f.inverse(xa, itbl, ninv, w1n);
    if constexpr (G::direct_pass0) {
        wp_store32<L, R, WP_PASS0>(xa, pcout, sp, lane);
    } else {
        wp_relayout32<L, R, WP_PASS0, WP_LANE>(xa, wimg, lane);
        wp_store32<L, R, WP_LANE>(xa, pcout, sp, lane);
    }
}

// the product keeps two frames in registers (2^(R+1) VGPRs): fewer waves per SIMD from 16 coefficients per lane on
template <int R, int MINW>
constexpr int wp32_mul_waves() { return R >= 5 ? (MINW > 4 ? 4 : MINW) : R == 4 ? (MINW > 5 ? 5 : MINW) : MINW; }

template <int L, int R, int WPB, int TIER, int MINW>
hipError_t launch_wp32_t(const plan_view& pv, const uint64_t* in, uint64_t* out, const frame_layout& fl, hipStream_t s) {
    constexpr int FPB = wp_geom<L, R>::FPW * WPB;
    dim3 grid((unsigned)((fl.batch + FPB - 1) / FPB), pv.num_primes);
    const size_t lds = wp32_lds_bytes<L, R>(WPB);
    hipLaunchKernelGGL((fwd_wp32<L, R, WPB, TIER, MINW>), grid, dim3(64 * WPB), lds, s, in, out, pv.consts, pv.tw_rb,
                       pv.rb.pairs_per_prime, fl.batch, fl.prime_stride, fl.poly_stride, (uint32_t)(fl.lazy_out ? 1 : 0));
    return hipGetLastError();
}
template <int L, int R, int WPB, int TIER, int MINW>
hipError_t launch_inv_wp32_t(const plan_view& pv, const uint64_t* in, const uint64_t* in2, uint64_t* out, const frame_layout& fl, hipStream_t s) {
    constexpr int FPB = wp_geom<L, R>::FPW * WPB;
    dim3 grid((unsigned)((fl.batch + FPB - 1) / FPB), pv.num_primes);
    const size_t lds = wp32_lds_bytes<L, R>(WPB);
    hipLaunchKernelGGL((inv_wp32<L, R, WPB, TIER, MINW>), grid, dim3(64 * WPB), lds, s, in, in2, out, pv.consts, pv.itw_rb,
                       pv.rb.pairs_per_prime, fl.batch, fl.prime_stride, fl.poly_stride);
    return hipGetLastError();
}
template <int L, int R, int WPB, int TIER, int MINW>
hipError_t launch_mul_wp32_t(const plan_view& pv, const uint64_t* a, const uint64_t* b, uint64_t* c, const frame_layout& fl, hipStream_t s) {
    constexpr int FPB = wp_geom<L, R>::FPW * WPB;
    dim3 grid((unsigned)((fl.batch + FPB - 1) / FPB), pv.num_primes);
    const size_t lds = wp32_lds_bytes<L, R>(WPB);
    hipLaunchKernelGGL((polymul_wp32<L, R, WPB, TIER, wp32_mul_waves<R, MINW>()>), grid, dim3(64 * WPB), lds, s, a, b, c, pv.consts, pv.tw_rb, pv.itw_rb,
                       pv.rb.pairs_per_prime, fl.batch, fl.prime_stride, fl.poly_stride);
    return hipGetLastError();
}
template <int L, int R, int WPB, int TIER, int MINW>
hipError_t init_wp32_t() {
    const int bytes = (int)wp32_lds_bytes<L, R>(WPB);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fwd_wp32<L, R, WPB, TIER, MINW>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&inv_wp32<L, R, WPB, TIER, MINW>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&polymul_wp32<L, R, WPB, TIER, wp32_mul_waves<R, MINW>()>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    return e;
}
template <int L, int R, int WPB, int TIER, int MINW>
constexpr rb_entry make_entry_wp32(int id) {
    rb_entry e{id, L, R, wp_geom<L, R>::FPW * WPB, MINW, (uint32_t)rb_geom<L, R>::table_pairs / 2, wp32_lds_bytes<L, R>(WPB),
               &build_table32_t<L, R>, &launch_wp32_t<L, R, WPB, TIER, MINW>, &init_wp32_t<L, R, WPB, TIER, MINW>, 1,
               &launch_inv_wp32_t<L, R, WPB, TIER, MINW>, &launch_mul_wp32_t<L, R, WPB, TIER, MINW>};
    e.narrow = TIER;
    return e;
}
#endif  // AGX_WP_Q32

}  // namespace AGX_TU
}  // namespace agx
