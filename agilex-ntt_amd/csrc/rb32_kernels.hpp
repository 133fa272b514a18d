// rb32_kernels.hpp -- narrow-modulus kernels: every modulus of the plan below 2^31 (the reference's own example modulus is
// 17 bits, src/main.cpp:55; BASELINE configs[0] is a 30-bit prime; SURVEY 7 H1 names this path as the first answer to the
// VALU / power wall of the 64-bit butterfly).  Same transform, same tables, same uint64_t data at the ABI -- only the
// arithmetic inside a workgroup is 32 bits wide:
//   w' = floor(w 2^32 / q) = precon >> 32 exactly (precon = floor(w 2^64 / q)),
//   Q  = w y - floor(w' y / 2^32) q  (mod 2^32)  in [0, 2q) for ANY 32-bit y          (src/kernel/ntt.cpp:344-363 in 32 bits)
// i.e. THREE multiplies per butterfly (v_mul_hi_u32 + 2 v_mul_lo_u32) instead of ten, a conditional subtract is
// v_sub + v_min (min(a, a - m) is a - m exactly when a >= m), and a coefficient is one VGPR: 16 per thread at 8 waves/SIMD.
// Fully reduced outputs are unique, so the results are bit-identical to the 64-bit kernels'.
//   TIER 2 (every q < 2^30): the reference's value ranges, coefficients in [0,4q) between stages (ntt.cpp:331-369), 4q <= 2^32.
//   TIER 1 (every q < 2^31): coefficients fully reduced after every butterfly (2q <= 2^32 is all the headroom there is).
// Included by reg_q32a.hip / reg_q32b.hip behind rb_kernels.hpp (geometry, static_for, wave_lds_sync, the dynamic LDS symbol).
#pragma once

namespace agx {
namespace AGX_TU {

typedef uint2 tw32;   // {w, w'}
typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));      // two adjacent coefficients: one 16-byte access

__device__ __forceinline__ tw32 load_uniform32(const tw32* p) {
    typedef const uint64_t __attribute__((address_space(4))) * const_ptr;
    const uint64_t v = *(const_ptr)(uintptr_t)p;      // scalar load: the entry lives in SGPRs
    return make_uint2((uint32_t)v, (uint32_t)(v >> 32));
}

// a in [0, 2^32), m > 0:  a - m if a >= m, else a   (the wrapped difference of a < m exceeds a)
__device__ __forceinline__ uint32_t csub32(uint32_t a, uint32_t m) {
    const uint32_t d = a - m;
    return d < a ? d : a;
}

template <int TIER>
struct q32_arith {
    static_assert(TIER == 1 || TIER == 2, "1: q < 2^31, 2: q < 2^30");
    uint32_t q, q2;
    __device__ __forceinline__ void init(uint64_t q64) {
        q = (uint32_t)q64;
        q2 = q << 1;
    }
    __device__ __forceinline__ uint32_t shoup(uint32_t y, tw32 w) const { return w.x * y - __umulhi(y, w.y) * q; }      // [0, 2q)
    // forward (Cooley-Tukey) butterfly, ntt.cpp:331-369
    __device__ __forceinline__ void ct(uint32_t& x, uint32_t& y, tw32 w) const {
        if constexpr (TIER == 2) {
            const uint32_t tx = csub32(x, q2);
            const uint32_t Q = shoup(y, w);
            x = tx + Q;
            y = tx + q2 - Q;
        } else {
            const uint32_t Q = csub32(shoup(y, w), q);
            const uint32_t s = x + Q, d = x + q - Q;
            x = csub32(s, q);
            y = csub32(d, q);
        }
    }
    // ntt.cpp:377-394; lazy outputs stay below 4q (tier 1 is reduced already)
    __device__ __forceinline__ uint32_t final_fwd(uint32_t v, bool lazy) const {
        if constexpr (TIER == 2) return lazy ? v : csub32(csub32(v, q2), q);
        else return v;
    }
    // inverse (Gentleman-Sande) butterfly: tier 2 keeps [0,2q), tier 1 [0,q)
    __device__ __forceinline__ void gs(uint32_t& x, uint32_t& y, tw32 w) const {
        if constexpr (TIER == 2) {
            const uint32_t s = x + y, d = x + q2 - y;
            x = csub32(s, q2);
            y = shoup(d, w);
        } else {
            const uint32_t s = x + y, d = x + q - y;
            x = csub32(s, q);
            y = csub32(shoup(d, w), q);
        }
    }
    // top inverse stage with n^-1 folded in: outputs fully reduced
    __device__ __forceinline__ void gs_last(uint32_t& x, uint32_t& y, tw32 ninv, tw32 w1n) const {
        const uint32_t s = x + y, d = x + (TIER == 2 ? q2 : q) - y;
        x = csub32(shoup(s, ninv), q);
        y = csub32(shoup(d, w1n), q);
    }
    // a 64-bit input word below 4q -> the transform's entry range (forward tier 2: [0,4q) as it is; inverse tier 2: [0,2q); tier 1: [0,q))
    template <bool INVERSE>
    __device__ __forceinline__ uint32_t enter(uint64_t v) const {
        const uint32_t lo = (uint32_t)v;
        if constexpr (TIER == 2) return INVERSE ? csub32(lo, q2) : lo;
        else {
            const uint32_t t = lo - q2;                       // 2^32 + lo - 2q when the word has bit 32 set (4q may exceed 2^32)
            const uint32_t r = (uint32_t)(v >> 32) ? t : csub32(lo, q2);
            return csub32(r, q);
        }
    }
    // a b mod q for a, b below 4q (tier 2) / below q (tier 1): Barrett with mu = floor(2^64 / q); the estimate is low by at most 1,
    // so the remainder lies in [0,2q) -- the inverse's entry range in tier 2 -- and tier 1 subtracts once more
    __device__ __forceinline__ uint32_t mulmod(uint32_t a, uint32_t b, uint64_t mu) const {
        const uint64_t p = (uint64_t)a * b;
        const uint32_t est = (uint32_t)__umul64hi(p, mu);
        const uint32_t r = (uint32_t)p - est * q;
        return TIER == 2 ? r : csub32(r, q);
    }
};

template <int L, int R, int TIER>
struct rb32_frame {
    using G = rb2_geom<L, R>;
    static constexpr int C = G::C, T = G::T, NP = G::NP;
    static constexpr uint32_t slab_words = (1u << L) + (L >= 5 ? (1u << (L >= 5 ? L - 5 : 0)) : 0u);
    // image word of coefficient e: one pad word per 32 (additive over disjoint bit fields: thread base + compile-time constant); of the
    // shifts 3..7 this one leaves the fewest bank conflicts for 32-bit accesses (32 banks per group of 32 lanes) at every (L, R) used here
    static __device__ __forceinline__ constexpr uint32_t img(uint32_t e) { return e + (e >> 5); }
    uint32_t tid;
    uint32_t* slab;
    q32_arith<TIER> a;

    template <int p>
    __device__ __forceinline__ uint32_t sbase() const {
        constexpr int rlo = G::rlo(p);
        return img((tid & ((1u << rlo) - 1u)) | ((tid >> rlo) << (rlo + R)));
    }
    template <int p>
    __device__ __forceinline__ void image_read(uint32_t (&x)[C]) const {
        const uint32_t sb = sbase<p>();
        static_for<0, C>([&](auto Rr) { constexpr int r = Rr; x[r] = slab[sb + img((uint32_t)r << G::rlo(p))]; });
    }
    template <int p>
    __device__ __forceinline__ void image_write(const uint32_t (&x)[C]) const {
        const uint32_t sb = sbase<p>();
        static_for<0, C>([&](auto Rr) { constexpr int r = Rr; slab[sb + img((uint32_t)r << G::rlo(p))] = x[r]; });
    }
    template <int p>
    __device__ __forceinline__ void exchange_sync() const {
        if constexpr (!G::exchange_is_wave_local(p)) __syncthreads();
        else wave_lds_sync();
    }
    // table entry j of pass p for this thread: scalar load when the column is wave-uniform, else uniform base + lane index
    template <int p>
    __device__ __forceinline__ tw32 entry(const tw32* tbl, int j) const {
        constexpr int rlo = G::rlo(p), H = G::H(p);
        if constexpr (G::uniform_pass(p)) {
            const uint32_t hcol = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> rlo));
            return load_uniform32(tbl + G::table_off(p) + (size_t)hcol * C + j);
        } else {
            const tw32* base = tbl + G::table_off(p) + (size_t)j * H;
            return base[tid >> rlo];
        }
    }

    // x in pass-0 layout (element tid + T r) -> forward transform in the last pass's layout (elements tid C .. tid C + C - 1)
    __device__ __forceinline__ void forward(uint32_t (&x)[C], const tw32* tbl, bool lazy_out) const {
        static_for<0, NP>([&](auto P) {
            constexpr int p = P;
            constexpr int rlo = G::rlo(p), hi = G::hi(p), ns = hi - rlo + 1;
            if constexpr (p > 0) image_read<p>(x);
            static_for<0, ns>([&](auto S) {
                constexpr int rb = (hi - rlo) - S;        // gap bits descend: Cooley-Tukey (ntt.cpp:155)
                constexpr int kk = R - 1 - rb;
                constexpr bool last_stage = (rlo + rb) == 0;
                static_for<0, C / 2>([&](auto B) {
                    constexpr int b = B;
                    constexpr int r0 = ((b >> rb) << (rb + 1)) | (b & ((1 << rb) - 1));
                    constexpr int r1 = r0 | (1 << rb);
                    constexpr int j = (1 << kk) + (r0 >> (rb + 1));
                    a.ct(x[r0], x[r1], entry<p>(tbl, j));
                    if constexpr (last_stage) {
                        x[r0] = a.final_fwd(x[r0], lazy_out);
                        x[r1] = a.final_fwd(x[r1], lazy_out);
                    }
                });
            });
            if constexpr (p < NP - 1) {
                image_write<p>(x);      // a thread overwrites exactly the words it read for this pass: only the read side needs ordering
                exchange_sync<p>();
            }
        });
    }
    // x in the last pass's layout, entry range -> inverse transform in pass-0 layout, fully reduced (n^-1 folded into the top stage)
    __device__ __forceinline__ void inverse(uint32_t (&x)[C], const tw32* itbl, tw32 ninv, tw32 w1n) const {
        static_for<0, NP>([&](auto Qp) {
            constexpr int p = NP - 1 - Qp;
            constexpr int rlo = G::rlo(p), hi = G::hi(p), ns = hi - rlo + 1;
            if constexpr (p < NP - 1) image_read<p>(x);
            static_for<0, ns>([&](auto S) {
                constexpr int rb = S;                     // gap bits ascend: Gentleman-Sande
                constexpr int kk = R - 1 - rb;
                constexpr bool top_stage = (rlo + rb) == L - 1;
                static_for<0, C / 2>([&](auto B) {
                    constexpr int b = B;
                    constexpr int r0 = ((b >> rb) << (rb + 1)) | (b & ((1 << rb) - 1));
                    constexpr int r1 = r0 | (1 << rb);
                    if constexpr (top_stage) a.gs_last(x[r0], x[r1], ninv, w1n);
                    else a.gs(x[r0], x[r1], entry<p>(itbl, (1 << kk) + (r0 >> (rb + 1))));
                });
            });
            if constexpr (p > 0) {
                image_write<p>(x);
                exchange_sync<p - 1>();
            }
        });
    }

    // last-pass layout -> lane-contiguous 64-bit stores (zero-extended), through the wave's own part of the image.
    // wide: 16 bytes per lane (two adjacent coefficients), needs 16-byte aligned frames; else 8 bytes per lane
    __device__ __forceinline__ void store_last_layout(const uint32_t (&x)[C], uint64_t* __restrict__ out, int64_t base, bool live, bool wide) const {
        const uint32_t own = img(tid << R);
        static_for<0, C>([&](auto Rr) { constexpr int r = Rr; slab[own + img((uint32_t)r)] = x[r]; });
        wave_lds_sync();
        const uint32_t wbase = (tid >> 6) << (6 + R), lane = tid & 63u;
        if (!live) return;
        if (wide) {      // wave-uniform
            static_for<0, C / 2>([&](auto K) {
                constexpr int k = K;
                const uint32_t e = wbase + 2u * lane + 128u * (uint32_t)k;      // even: e and e + 1 share a pad group
                const uint32_t s = img(e);
                u64x2 v;
                v.x = (uint64_t)slab[s];
                v.y = (uint64_t)slab[s + 1];
                __builtin_nontemporal_store(v, reinterpret_cast<u64x2*>(out + base + e));
            });
        } else {
            static_for<0, C>([&](auto Rr) {
                constexpr int r = Rr;
                const uint32_t e = wbase + lane + 64u * (uint32_t)r;
                __builtin_nontemporal_store((uint64_t)slab[img(e)], &out[base + e]);
            });
        }
    }
    // lane-contiguous 64-bit loads -> last-pass layout, optionally times in2 (the pointwise product fused into the load)
    template <bool INVERSE>
    __device__ __forceinline__ void load_last_layout(uint32_t (&x)[C], const uint64_t* __restrict__ in, const uint64_t* __restrict__ in2,
                                                     uint64_t mu, int64_t base, bool wide) const {
        const uint32_t wbase = (tid >> 6) << (6 + R), lane = tid & 63u;
        if (in2) {      // wave-uniform; the product fused into the load (agx_ntt_polymul's fallback path): no priority games, one word at a time
            __builtin_amdgcn_s_setprio(0);
            static_for<0, C>([&](auto Rr) {
                constexpr int r = Rr;
                const uint32_t e = wbase + lane + 64u * (uint32_t)r;
                slab[img(e)] = a.mulmod(a.template enter<false>(__builtin_nontemporal_load(&in[base + e])), a.template enter<false>(__builtin_nontemporal_load(&in2[base + e])), mu);
            });
        } else if (wide) {
            // every load is issued first (the kernels enter at raised priority), then the priority drops and the words are converted and staged
            u64x2 v[C / 2];
            static_for<0, C / 2>([&](auto K) { constexpr int k = K; v[k] = __builtin_nontemporal_load(reinterpret_cast<const u64x2*>(in + base + wbase + 2u * lane + 128u * (uint32_t)k)); });
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_setprio(0);
            static_for<0, C / 2>([&](auto K) {
                constexpr int k = K;
                const uint32_t s = img(wbase + 2u * lane + 128u * (uint32_t)k);
                slab[s] = a.template enter<INVERSE>(v[k].x);
                slab[s + 1] = a.template enter<INVERSE>(v[k].y);
            });
        } else {
            uint64_t v[C];
            static_for<0, C>([&](auto Rr) { constexpr int r = Rr; v[r] = __builtin_nontemporal_load(&in[base + wbase + lane + 64u * (uint32_t)r]); });
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_setprio(0);
            static_for<0, C>([&](auto Rr) { constexpr int r = Rr; slab[img(wbase + lane + 64u * (uint32_t)r)] = a.template enter<INVERSE>(v[r]); });
        }
        wave_lds_sync();
        const uint32_t own = img(tid << R);
        static_for<0, C>([&](auto Rr) { constexpr int r = Rr; x[r] = slab[own + img((uint32_t)r)]; });
    }
};

#define AGX_RB32_PROLOGUE                                                                         \
    using F = rb32_frame<L, R, TIER>;                                                             \
    constexpr int C = F::C, T = F::T;                                                             \
    static_assert(T >= 64, "one frame spans whole waves");                                        \
    F f;                                                                                          \
    f.tid = threadIdx.x & (T - 1);                                                                \
    const uint32_t slot = threadIdx.x / T;                                                        \
    uint64_t fx = (uint64_t)blockIdx.x * PPB + slot;                                              \
    const bool live = fx < frames_x;                                                              \
    if (!live) fx = frames_x - 1;                                                                 \
    const uint32_t prime = blockIdx.y;                                                            \
    const prime_consts pc = consts[prime];                                                        \
    f.a.init(pc.q);                                                                               \
    f.slab = reinterpret_cast<uint32_t*>(agx_dyn_lds) + (size_t)slot * F::slab_words;             \
    const tw32* tbl = reinterpret_cast<const tw32*>(tw_rb) + (size_t)prime * pairs_per_prime * 2; \
    const int64_t base = (int64_t)prime * prime_stride + (int64_t)fx * poly_stride;               \
    const bool wide = (flags & 2u) != 0

// flags: bit 0 = lazy outputs (forward), bit 1 = frames are 16-byte aligned (wide loads / stores legal)
template <int L, int R, int PPB, int TIER, int MINW>
__global__ void __launch_bounds__((1 << (L - R)) * PPB, MINW)
fwd_q32(const uint64_t* __restrict__ in, uint64_t* __restrict__ out, const prime_consts* __restrict__ consts,
        const twpair* __restrict__ tw_rb, uint32_t pairs_per_prime, uint64_t frames_x, int64_t prime_stride, int64_t poly_stride, uint32_t flags) {
    // a wave issues its frame loads at raised priority (new waves are the youngest on their SIMD and would otherwise queue behind
    // seven older ones' butterflies before their first load goes out): +3 ... +8 % at every size (profiles/r03_q32_priority.txt)
    __builtin_amdgcn_s_setprio(3);
    AGX_RB32_PROLOGUE;
    uint32_t x[C];
    {
    static_for<0, C>([&](auto Rr) {
        constexpr int r = Rr;
        if constexpr (TIER == 2) {
            // inputs are below 4q <= 2^32: only the low words are needed (ntt.cpp:331-332 accepts [0,4q))
            x[r] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(in + base + f.tid + (uint32_t)r * T));
        } else {
            x[r] = f.a.template enter<false>(__builtin_nontemporal_load(&in[base + f.tid + (uint32_t)r * T]));
        }
    });
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_setprio(0);
    f.forward(x, tbl, (flags & 1u) != 0);
    f.store_last_layout(x, out, base, live, wide);
}

template <int L, int R, int PPB, int TIER, int MINW>
__global__ void __launch_bounds__((1 << (L - R)) * PPB, MINW)
inv_q32(const uint64_t* __restrict__ in, const uint64_t* __restrict__ in2, uint64_t* __restrict__ out, const prime_consts* __restrict__ consts,
        const twpair* __restrict__ tw_rb, uint32_t pairs_per_prime, uint64_t frames_x, int64_t prime_stride, int64_t poly_stride, uint32_t flags) {
    __builtin_amdgcn_s_setprio(3);      // until the frame loads are out (load_last_layout lowers it)
    AGX_RB32_PROLOGUE;      // tw_rb = the plan's inverse table here
    uint32_t x[C];
    f.template load_last_layout<true>(x, in, in2, pc.mu_hi, base, wide);
    const tw32 ninv = make_uint2((uint32_t)pc.n_inv, (uint32_t)(pc.n_inv_p >> 32)), w1n = make_uint2((uint32_t)pc.w1n, (uint32_t)(pc.w1n_p >> 32));
    f.inverse(x, tbl, ninv, w1n);
    if (live) {
        static_for<0, C>([&](auto Rr) { constexpr int r = Rr; __builtin_nontemporal_store((uint64_t)x[r], &out[base + f.tid + (uint32_t)r * T]); });
    }
}

// c = INTT(NTT(a) o NTT(b)) for one frame in one launch: both forward results stay in registers (2^(R+1) VGPRs), 24n bytes of traffic
template <int L, int R, int PPB, int TIER, int MINW>
__global__ void __launch_bounds__((1 << (L - R)) * PPB, MINW)
polymul_q32(const uint64_t* __restrict__ pa, const uint64_t* __restrict__ pb, uint64_t* __restrict__ pcout, const prime_consts* __restrict__ consts,
            const twpair* __restrict__ tw_rb, const twpair* __restrict__ itw_rb, uint32_t pairs_per_prime, uint64_t frames_x,
            int64_t prime_stride, int64_t poly_stride, uint32_t flags) {
    __builtin_amdgcn_s_setprio(3);      // until a's loads are out
    AGX_RB32_PROLOGUE;
    (void)wide;
    const tw32* itbl = reinterpret_cast<const tw32*>(itw_rb) + (size_t)prime * pairs_per_prime * 2;
    uint32_t xa[C], xb[C];
    {
        uint64_t va[C];
        static_for<0, C>([&](auto Rr) { constexpr int r = Rr; va[r] = __builtin_nontemporal_load(&pa[base + f.tid + (uint32_t)r * T]); });
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_setprio(0);
        static_for<0, C>([&](auto Rr) { constexpr int r = Rr; xa[r] = f.a.template enter<false>(va[r]); });
    }
    f.forward(xa, tbl, true);
    asm volatile("" ::: "memory");      // b's loads stay behind NTT(a)
    static_for<0, C>([&](auto Rr) { constexpr int r = Rr; xb[r] = f.a.template enter<false>(__builtin_nontemporal_load(&pb[base + f.tid + (uint32_t)r * T])); });
    __syncthreads();      // the image is reused: every wave must be done reading NTT(a)'s exchanges
    {
        const tw32* tbl2 = tbl;
        asm volatile("" : "+s"(tbl2));      // opaque: or NTT(a)'s table entries are kept (spilled) for reuse instead of re-read from L2
        f.forward(xb, tbl2, true);
    }
    static_for<0, C>([&](auto Rr) { constexpr int r = Rr; xa[r] = f.a.mulmod(xa[r], xb[r], pc.mu_hi); });
    const tw32 ninv = make_uint2((uint32_t)pc.n_inv, (uint32_t)(pc.n_inv_p >> 32)), w1n = make_uint2((uint32_t)pc.w1n, (uint32_t)(pc.w1n_p >> 32));
    f.inverse(xa, itbl, ninv, w1n);
    if (live) {
        static_for<0, C>([&](auto Rr) { constexpr int r = Rr; __builtin_nontemporal_store((uint64_t)xa[r], &pcout[base + f.tid + (uint32_t)r * T]); });
    }
}

// ---- host side: table, launches, registry entry ------------------------------------------------------------------------------
// same geometry as build_table_t<L, R, true> (wave-uniform passes keep one column's C entries contiguous, per-lane passes one
// entry's columns), entries {w, precon >> 32}; appended to `out` as raw bytes (two entries per ulonglong2)
template <int L, int R>
void build_table32_t(const regblock_layout&, const uint64_t* tw, const uint64_t* pre, std::vector<ulonglong2>& out) {
    using G = rb_geom<L, R>;
    static_assert(G::table_pairs % 2 == 0, "two 8-byte entries per 16-byte slot");
    std::vector<uint2> t((size_t)G::table_pairs, make_uint2(0, 0));
    for (int p = 0; p < G::NP; ++p) {
        const int rlo = G::rlo(p), hi = G::hi(p), H = G::H(p);
        uint2* tp = t.data() + G::table_off(p);
        for (int j = 1; j < G::C; ++j) {
            int k = 0;
            while ((2 << k) <= j) ++k;
            const int o = j - (1 << k), rb_bit = R - 1 - k, b = rlo + rb_bit;
            if (b > hi) continue;      // stage belongs to an earlier pass (short last pass)
            const uint32_t m_local = 1u << (L - 1 - b);
            for (int h = 0; h < H; ++h) {
                const uint32_t idx = m_local + ((uint32_t)h << k) + (uint32_t)o;      // natural twiddle index m + i (ntt.cpp:298-300)
                const size_t at = rb2_geom<L, R>::uniform_pass(p) ? (size_t)h * G::C + j : (size_t)j * H + h;
                tp[at] = make_uint2((uint32_t)tw[idx], (uint32_t)(pre[idx] >> 32));
            }
        }
    }
    const size_t start = out.size();
    out.resize(start + (size_t)G::table_pairs / 2);
    std::memcpy(out.data() + start, t.data(), t.size() * sizeof(uint2));
}

template <int L, int R, int PPB>
constexpr size_t q32_lds_bytes() { return (size_t)((1u << L) + (1u << (L - 5))) * 4 * PPB; }

inline uint32_t q32_flags(const void* a, const void* b, const void* c, const frame_layout& fl) {
    const bool aligned = (((uintptr_t)a | (uintptr_t)b | (uintptr_t)c) & 15u) == 0 && ((fl.prime_stride | fl.poly_stride) & 1) == 0;
    return (fl.lazy_out ? 1u : 0u) | (aligned ? 2u : 0u);
}

template <int L, int R, int PPB, int TIER, int MINW>
hipError_t launch_q32_t(const plan_view& pv, const uint64_t* in, uint64_t* out, const frame_layout& fl, hipStream_t s) {
    using G = rb_geom<L, R>;
    dim3 grid((unsigned)((fl.batch + PPB - 1) / PPB), pv.num_primes);
    const size_t lds = q32_lds_bytes<L, R, PPB>();
    hipLaunchKernelGGL((fwd_q32<L, R, PPB, TIER, MINW>), grid, dim3(G::T * PPB), lds, s, in, out, pv.consts, pv.tw_rb,
                       pv.rb.pairs_per_prime, fl.batch, fl.prime_stride, fl.poly_stride, q32_flags(in, out, nullptr, fl));
    return hipGetLastError();
}
template <int L, int R, int PPB, int TIER, int MINW>
hipError_t launch_inv_q32_t(const plan_view& pv, const uint64_t* in, const uint64_t* in2, uint64_t* out, const frame_layout& fl, hipStream_t s) {
    using G = rb_geom<L, R>;
    dim3 grid((unsigned)((fl.batch + PPB - 1) / PPB), pv.num_primes);
    const size_t lds = q32_lds_bytes<L, R, PPB>();
    hipLaunchKernelGGL((inv_q32<L, R, PPB, TIER, MINW>), grid, dim3(G::T * PPB), lds, s, in, in2, out, pv.consts, pv.itw_rb,
                       pv.rb.pairs_per_prime, fl.batch, fl.prime_stride, fl.poly_stride, q32_flags(in, in2, out, fl));
    return hipGetLastError();
}
template <int L, int R, int PPB, int TIER, int MINW>
hipError_t launch_mul_q32_t(const plan_view& pv, const uint64_t* a, const uint64_t* b, uint64_t* c, const frame_layout& fl, hipStream_t s) {
    using G = rb_geom<L, R>;
    dim3 grid((unsigned)((fl.batch + PPB - 1) / PPB), pv.num_primes);
    const size_t lds = q32_lds_bytes<L, R, PPB>();
    hipLaunchKernelGGL((polymul_q32<L, R, PPB, TIER, MINW>), grid, dim3(G::T * PPB), lds, s, a, b, c, pv.consts, pv.tw_rb, pv.itw_rb,
                       pv.rb.pairs_per_prime, fl.batch, fl.prime_stride, fl.poly_stride, 0u);
    return hipGetLastError();
}
template <int L, int R, int PPB, int TIER, int MINW>
hipError_t init_q32_t() {
    const int bytes = (int)q32_lds_bytes<L, R, PPB>();
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fwd_q32<L, R, PPB, TIER, MINW>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&inv_q32<L, R, PPB, TIER, MINW>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&polymul_q32<L, R, PPB, TIER, MINW>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    return e;
}

// TIER 2 needs every modulus below 2^30, TIER 1 below 2^31 (rb_entry::narrow); the tables must honour the precon contract (arith >= 1)
template <int L, int R, int PPB, int TIER, int MINW>
constexpr rb_entry make_entry_q32(int id) {
    rb_entry e{id, L, R, PPB, MINW, (uint32_t)rb_geom<L, R>::table_pairs / 2, q32_lds_bytes<L, R, PPB>(),
               &build_table32_t<L, R>, &launch_q32_t<L, R, PPB, TIER, MINW>, &init_q32_t<L, R, PPB, TIER, MINW>, 1,
               &launch_inv_q32_t<L, R, PPB, TIER, MINW>, &launch_mul_q32_t<L, R, PPB, TIER, MINW>};
    e.narrow = TIER;
    return e;
}

}  // namespace AGX_TU
}  // namespace agx
