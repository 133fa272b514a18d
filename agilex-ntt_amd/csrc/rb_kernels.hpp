// rb_kernels.hpp -- the register-blocked kernels (forward, inverse, fused product; one workgroup per frame or a resident grid for the
// inverse of the large sizes) over rb2_frame (rb_frame.hpp), their launch glue and the registry-entry constructors.  Included by the
// reg_*.hip translation units, each of which instantiates one group of registry entries under its own namespace AGX_TU.
#pragma once
#include "rb_frame.hpp"

namespace agx {
namespace AGX_TU {

#define AGX_RB2_PROLOGUE                                                                          \
    using F = rb2_frame<L, R, (ARITH & 1) == 1, (ARITH >> 1)>;                                    \
    constexpr int C = F::C, T = F::T;                                                             \
    static_assert(T >= 64, "one frame must span whole waves");                                    \
    F f;                                                                                          \
    f.tid = threadIdx.x & (T - 1);                                                                \
    const uint32_t slot = threadIdx.x / T;                                                        \
    uint64_t fx = (uint64_t)blockIdx.x * PPB + slot;                                              \
    const bool live = fx < frames_x;                                                              \
    if (!live) fx = frames_x - 1;                         /* keep every thread on the barriers */ \
    const uint32_t prime = blockIdx.y;                                                            \
    f.init_consts(consts[prime].q, consts[prime].est);                                            \
    f.slab = reinterpret_cast<uint64_t*>(agx_dyn_lds) + (size_t)slot * F::slab_elems;             \
    const int64_t base = (int64_t)prime * prime_stride + (int64_t)fx * poly_stride

template <int L, int R, int PPB, int ARITH, int MINW>
__global__ void __launch_bounds__((1 << (L - R)) * PPB, MINW)
fwd_rb2(const uint64_t* __restrict__ in, uint64_t* __restrict__ out,
        const prime_consts* __restrict__ consts, const twpair* __restrict__ tw_rb,
        uint32_t pairs_per_prime, uint64_t frames_x,
        int64_t prime_stride, int64_t poly_stride, uint32_t lazy_out) {
    uint64_t t_entry = 0;
    if constexpr (((ARITH >> 1) & kOptTrace) != 0) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_entry) : : "memory");
    if constexpr (((ARITH >> 1) & kOptPrio) != 0) __builtin_amdgcn_s_setprio(3);
    AGX_RB2_PROLOGUE;
    f.lazy_out = lazy_out != 0;
    uint64_t x[C];
    if constexpr (((ARITH >> 1) & kOptNtLoad) != 0) {
        const uint64_t* src = in + base;     // wave-uniform
#pragma unroll
        for (int r = 0; r < C; ++r) x[r] = __builtin_nontemporal_load(src + (uint32_t)r * T + f.tid);
    } else if constexpr (F::SCALAR_BASE) {
        const uint64_t* src = in + base;     // wave-uniform
#pragma unroll
        for (int r = 0; r < C; ++r) x[r] = (src + (uint32_t)r * T)[f.tid];
    } else {
#pragma unroll
        for (int r = 0; r < C; ++r) x[r] = in[base + f.tid + (uint32_t)r * T];
    }
    if constexpr (F::PRIO && !F::PRIO_BARRIER) {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_setprio(0);
    }
    if constexpr (F::TRACE) {
        f.ts[0] = t_entry;
        f.template stamp<1>(x[C - 1]);
    }
    f.forward(x, tw_rb + (size_t)prime * pairs_per_prime);
    f.store_last_layout(x, out, base, live);
}

// Loop kernels of the inverse (n = 16384 / 32768): a resident grid (as many workgroups as the chip holds at once) walks over the
// frames.  A workgroup that transforms frame after frame issues the next frame's loads right behind the current frame's stores, so
// the store drain of one frame and the load latency of the next overlap -- with one or two workgroups per CU (the frame's image fills
// the LDS) nothing else on the CU could cover either.  The frame-to-frame hand-over of the LDS image needs one bare s_barrier.
// inv_rb2_dloop draws its frames from a ticket counter (a CU that runs a few per cent faster simply takes more frames): thread 0
// draws one frame ahead, right behind the frame loads, and hands the number round through a two-slot LDS mailbox that everybody reads
// behind the frame's own cross-wave barrier.  ticket[0] = frames handed out beyond the first round, ticket[1] = retired workgroups;
// the last workgroup out zeroes both (the plan keeps one pair per stream).  inv_rb2_loop walks with a fixed stride and carries no
// state: it serves launches that are being captured into a hipGraph and streams that have no pair (agx_ntt.cpp: plan_ticket_for).
// The forward transform measured slower in both forms (-3 ... -6 %, profiles/r03_streamed_kernels_sweeps.txt) and stays one
// workgroup per frame.
__device__ __forceinline__ void dloop_retire(uint32_t* ticket) {
    if (threadIdx.x == 0) {
        if (atomicAdd(ticket + 1, 1u) == gridDim.x - 1) {     // last workgroup out: reset the pair for its next launch
            __threadfence();
            ticket[0] = 0;
            ticket[1] = 0;
        }
    }
}

template <int L, int R, int ARITH, int MINW>
__global__ void __launch_bounds__((1 << (L - R)), MINW)
inv_rb2_dloop(const uint64_t* __restrict__ in, const uint64_t* __restrict__ in2, uint64_t* __restrict__ out,
              const prime_consts* __restrict__ consts, const twpair* __restrict__ itw_rb,
              uint32_t pairs_per_prime, uint32_t batch, uint32_t total, int64_t prime_stride, int64_t poly_stride,
              uint32_t* __restrict__ ticket) {
    using F = rb2_frame<L, R, (ARITH & 1) == 1, (ARITH >> 1)>;
    static_assert(!F::G::exchange_is_wave_local(0), "the mailbox is read behind the last exchange's workgroup barrier");
    constexpr int C = F::C, T = F::T;
    F f;
    f.tid = threadIdx.x;
    f.slab = reinterpret_cast<uint64_t*>(agx_dyn_lds);
    volatile uint32_t* mailbox = reinterpret_cast<volatile uint32_t*>(reinterpret_cast<unsigned char*>(f.slab) + F::image_bytes);
    uint32_t fr = blockIdx.x;
    for (uint32_t it = 0;; ++it) {
        const uint32_t prime = fr / batch, poly = fr % batch;
        const int64_t base = (int64_t)prime * prime_stride + (int64_t)poly * poly_stride;
        const prime_consts pc = consts[prime];
        const barrett128 bk{pc.q, pc.mu_hi, pc.mu_lo};
        f.init_consts(pc.q, pc.est);
        uint64_t x[C];
        f.load_last_issue(x, in, base);
        if (threadIdx.x == 0) mailbox[it & 1u] = atomicAdd(ticket, 1u) + gridDim.x;
        if (it) __builtin_amdgcn_s_barrier();   // every wave has read the previous frame's last exchange
        f.load_last_stage(x, in2, bk, base);
        f.inverse(x, itw_rb + (size_t)prime * pairs_per_prime, pc);      // its last exchange is a workgroup barrier: the mailbox is visible behind it
        const uint32_t next = (uint32_t)__builtin_amdgcn_readfirstlane((int)mailbox[it & 1u]);
#pragma unroll
        for (int r = 0; r < C; ++r) {
            if constexpr (((ARITH >> 1) & kOptNtStore) != 0) __builtin_nontemporal_store(x[r], &out[base + f.tid + (uint32_t)r * T]);
            else out[base + f.tid + (uint32_t)r * T] = x[r];
        }
        fr = next;
        if (fr >= total) break;
    }
    dloop_retire(ticket);
}

template <int L, int R, int ARITH, int MINW>
__global__ void __launch_bounds__((1 << (L - R)), MINW)
inv_rb2_loop(const uint64_t* __restrict__ in, const uint64_t* __restrict__ in2, uint64_t* __restrict__ out,
             const prime_consts* __restrict__ consts, const twpair* __restrict__ itw_rb,
             uint32_t pairs_per_prime, uint32_t batch, uint32_t total, int64_t prime_stride, int64_t poly_stride) {
    using F = rb2_frame<L, R, (ARITH & 1) == 1, (ARITH >> 1)>;
    constexpr int C = F::C, T = F::T;
    F f;
    f.tid = threadIdx.x;
    f.slab = reinterpret_cast<uint64_t*>(agx_dyn_lds);
    bool first = true;
    for (uint32_t fr = blockIdx.x; fr < total; fr += gridDim.x) {
        const uint32_t prime = fr / batch, poly = fr % batch;
        const int64_t base = (int64_t)prime * prime_stride + (int64_t)poly * poly_stride;
        const prime_consts pc = consts[prime];
        const barrett128 bk{pc.q, pc.mu_hi, pc.mu_lo};
        f.init_consts(pc.q, pc.est);
        uint64_t x[C];
        f.load_last_issue(x, in, base);
        // the previous frame's last exchange is read across waves: nobody may stage into the image before all have read
        if (!first) __builtin_amdgcn_s_barrier();
        f.load_last_stage(x, in2, bk, base);
        f.inverse(x, itw_rb + (size_t)prime * pairs_per_prime, pc);
#pragma unroll
        for (int r = 0; r < C; ++r) {
            if constexpr (((ARITH >> 1) & kOptNtStore) != 0) __builtin_nontemporal_store(x[r], &out[base + f.tid + (uint32_t)r * T]);
            else out[base + f.tid + (uint32_t)r * T] = x[r];
        }
        first = false;
    }
}

template <int L, int R, int PPB, int ARITH, int MINW>
__global__ void __launch_bounds__((1 << (L - R)) * PPB, MINW)
inv_rb2(const uint64_t* __restrict__ in, const uint64_t* __restrict__ in2, uint64_t* __restrict__ out,
        const prime_consts* __restrict__ consts, const twpair* __restrict__ itw_rb,
        uint32_t pairs_per_prime, uint64_t frames_x,
        int64_t prime_stride, int64_t poly_stride) {
    uint64_t t_entry = 0;
    if constexpr (((ARITH >> 1) & kOptTrace) != 0) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_entry) : : "memory");
    if constexpr (((ARITH >> 1) & kOptPrio) != 0) __builtin_amdgcn_s_setprio(3);   // until the frame loads are out
    AGX_RB2_PROLOGUE;
    const prime_consts pc = consts[prime];
    const barrett128 bk{pc.q, pc.mu_hi, pc.mu_lo};
    uint64_t x[C];
    f.load_last_issue(x, in, base);
    if constexpr (F::TRACE) {
        f.ts[0] = t_entry;
        f.template stamp<1>(x[C - 1]);      // frame arrived
    }
    f.load_last_stage(x, in2, bk, base);
    if constexpr (F::TRACE) f.template stamp<2>(x[C - 1]);      // staged through the image into the last pass's layout
    f.inverse(x, itw_rb + (size_t)prime * pairs_per_prime, pc);
    if (live) {
#pragma unroll
        for (int r = 0; r < C; ++r) {
            if constexpr (((ARITH >> 1) & kOptNtStore) != 0) __builtin_nontemporal_store(x[r], &out[base + f.tid + (uint32_t)r * T]);
            else out[base + f.tid + (uint32_t)r * T] = x[r];
        }
    }
    if constexpr (F::TRACE) {
        f.template stamp<10>(x[0]);         // stores issued
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        f.template stamp<11>(x[0]);         // stores retired
        f.trace_flush();
    }
}

// c = INTT(NTT(a) o NTT(b)) for one frame without leaving the chip: both forward transforms end in
// the same register layout, the product is taken there, and the inverse starts from it (no staging
// through the image at either seam).  HBM traffic 24n bytes per product.
template <int L, int R, int PPB, int ARITH, int MINW>
__global__ void __launch_bounds__((1 << (L - R)) * PPB, (MINW > AGX_POLYMUL_MAXW ? AGX_POLYMUL_MAXW : MINW))   // one frame more in registers
polymul_rb2(const uint64_t* __restrict__ a, const uint64_t* __restrict__ b, uint64_t* __restrict__ c,
            const prime_consts* __restrict__ consts, const twpair* __restrict__ tw_rb, const twpair* __restrict__ itw_rb,
            uint32_t pairs_per_prime, uint64_t frames_x, int64_t prime_stride, int64_t poly_stride) {
    AGX_RB2_PROLOGUE;
    const prime_consts pc = consts[prime];
    const barrett128 bk{pc.q, pc.mu_hi, pc.mu_lo};
    // q <= 2^60: the product below takes operands in [0,4q) (16 q^2 < 2^124, quotient < 16q <= 2^64), so
    // both forward transforms may skip their last two conditional subtracts
    f.lazy_out = F::LAZY16;
    uint64_t xa[C], xb[C];
    constexpr bool NTL = ((ARITH >> 1) & kOptNtLoad) != 0, NTS = ((ARITH >> 1) & kOptNtStore) != 0;
#pragma unroll
    for (int r = 0; r < C; ++r) xa[r] = NTL ? __builtin_nontemporal_load(&a[base + f.tid + (uint32_t)r * T]) : a[base + f.tid + (uint32_t)r * T];
    f.forward(xa, tw_rb + (size_t)prime * pairs_per_prime);
    // b is fetched only now: holding it across NTT(a) would cost 2^R more register pairs and spill
#pragma unroll
    for (int r = 0; r < C; ++r) xb[r] = NTL ? __builtin_nontemporal_load(&b[base + f.tid + (uint32_t)r * T]) : b[base + f.tid + (uint32_t)r * T];
    __syncthreads();   // the image is reused: every wave must be done reading NTT(a)'s exchanges
    f.forward(xb, tw_rb + (size_t)prime * pairs_per_prime);
#pragma unroll
    for (int r = 0; r < C; ++r) xa[r] = mul_mod_barrett(xa[r], xb[r], bk);
    f.inverse(xa, itw_rb + (size_t)prime * pairs_per_prime, pc);
    if (live) {
#pragma unroll
        for (int r = 0; r < C; ++r) {
            if constexpr (NTS) __builtin_nontemporal_store(xa[r], &c[base + f.tid + (uint32_t)r * T]);
            else c[base + f.tid + (uint32_t)r * T] = xa[r];
        }
    }
}

// The same product with only ONE frame in registers at a time: NTT(first) is parked in c's own frame (plain global
// memory: written, and read back a few microseconds later by the same workgroup, so it is served by the XCD's L2)
// while NTT(second) is computed, then fetched in the last pass's layout (each thread its 2^R consecutive
// coefficients), multiplied, and the inverse starts from there.  This is what lets n = 16384 (16 coefficients per
// thread: two frames do not fit 128 VGPRs) run the product in one launch, and what frees the smaller sizes from the
// second frame's registers.  The host passes as `first` the operand c aliases, if any: a workgroup reads all of
// `first` before it writes c, and `second` is then a different buffer.  Squaring (a == b, with or without c aliasing
// them) never comes here: `second` would be the parked words themselves when c aliases too, so the
// launcher sends it to polysquare_rb2 below (a branch on first == second inside this kernel cost 240-320 B of scratch).
template <int L, int R, int ARITH, int MINW>
__global__ void __launch_bounds__((1 << (L - R)), MINW)
polymul_rb2_park(const uint64_t* __restrict__ first, const uint64_t* __restrict__ second, uint64_t* __restrict__ c,
                 const prime_consts* __restrict__ consts, const twpair* __restrict__ tw_rb, const twpair* __restrict__ itw_rb,
                 uint32_t pairs_per_prime, uint64_t frames_x, int64_t prime_stride, int64_t poly_stride) {
    constexpr int PPB = 1;
    AGX_RB2_PROLOGUE;
    const prime_consts pc = consts[prime];
    const barrett128 bk{pc.q, pc.mu_hi, pc.mu_lo};
    f.lazy_out = F::LAZY16;     // the Barrett product takes operands in [0,4q) when q <= 2^60
    constexpr bool NTL = ((ARITH >> 1) & kOptNtLoad) != 0, NTS = ((ARITH >> 1) & kOptNtStore) != 0;
    uint64_t x[C];
#pragma unroll
    for (int r = 0; r < C; ++r) x[r] = NTL ? __builtin_nontemporal_load(&first[base + f.tid + (uint32_t)r * T]) : first[base + f.tid + (uint32_t)r * T];
    f.forward(x, tw_rb + (size_t)prime * pairs_per_prime);
    static_assert(F::STREAM_TW, "the parked product is a streamed single-frame kernel");
    {
        // Thread-private parking: every thread stores its own 2^R values of NTT(first) (register r at c[base + tid + r T]: lane-contiguous,
        // not the natural element order -- c's frame is only scratch here) and later reads back exactly the words it wrote, one
        // register at a time, so the product needs neither an LDS redistribution nor a second frame in registers.
        if (live) {
#pragma unroll
            for (int r = 0; r < C; ++r) c[base + f.tid + (uint32_t)r * T] = x[r];
        }
        asm volatile("" ::: "memory");      // the second operand's loads stay behind the parking stores (or two frames would be live)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < C; ++r) x[r] = NTL ? __builtin_nontemporal_load(&second[base + f.tid + (uint32_t)r * T]) : second[base + f.tid + (uint32_t)r * T];
        __syncthreads();   // the image is reused
        {
            const twpair* tbl2 = tw_rb + (size_t)prime * pairs_per_prime;
            asm volatile("" : "+s"(tbl2));      // opaque: or the first transform's table entries are kept (spilled) for reuse instead of re-read from L2
            f.forward(x, tbl2);
        }
        const uint64_t* parked = c;
        asm volatile("" : "+s"(parked));      // opaque: the compiler must re-load the parked words, not keep them in registers
        constexpr int GRP = 4;
        static_for<0, C / GRP>([&](auto Gq) {
            constexpr int g = Gq;
            uint64_t z[GRP];
            static_for<0, GRP>([&](auto I) { constexpr int r = g * GRP + (int)I; z[I] = parked[base + f.tid + (uint32_t)r * T]; });
            __builtin_amdgcn_sched_barrier(0);
            static_for<0, GRP>([&](auto I) {
                constexpr int r = g * GRP + (int)I;
                asm volatile("" : "+v"(x[r]));
                x[r] = mul_mod_barrett(z[I], x[r], bk);
                asm volatile("" : "+v"(x[r]));
            });
            __builtin_amdgcn_sched_barrier(0);
        });
        f.inverse(x, itw_rb + (size_t)prime * pairs_per_prime, pc);
        if (live) {
#pragma unroll
            for (int r = 0; r < C; ++r) {
                if constexpr (NTS) __builtin_nontemporal_store(x[r], &c[base + f.tid + (uint32_t)r * T]);
                else c[base + f.tid + (uint32_t)r * T] = x[r];
            }
        }
    }
}

// c = a * a in Z_q[X]/(X^n + 1): one forward transform, the product of every coefficient with itself where it sits in
// registers, the inverse.  16n bytes of traffic; a workgroup reads its whole frame before it writes any of it, so c may
// alias a.  Serves agx_ntt_polymul calls with a == b on the plans whose fused product is polymul_rb2_park.
template <int L, int R, int ARITH, int MINW>
__global__ void __launch_bounds__((1 << (L - R)), MINW)
polysquare_rb2(const uint64_t* __restrict__ a, uint64_t* __restrict__ c,
               const prime_consts* __restrict__ consts, const twpair* __restrict__ tw_rb, const twpair* __restrict__ itw_rb,
               uint32_t pairs_per_prime, uint64_t frames_x, int64_t prime_stride, int64_t poly_stride) {
    constexpr int PPB = 1;
    AGX_RB2_PROLOGUE;
    const prime_consts pc = consts[prime];
    const barrett128 bk{pc.q, pc.mu_hi, pc.mu_lo};
    f.lazy_out = F::LAZY16;     // the Barrett product takes operands in [0,4q) when q <= 2^60
    constexpr bool NTL = ((ARITH >> 1) & kOptNtLoad) != 0, NTS = ((ARITH >> 1) & kOptNtStore) != 0;
    uint64_t x[C];
#pragma unroll
    for (int r = 0; r < C; ++r) x[r] = NTL ? __builtin_nontemporal_load(&a[base + f.tid + (uint32_t)r * T]) : a[base + f.tid + (uint32_t)r * T];
    f.forward(x, tw_rb + (size_t)prime * pairs_per_prime);
#pragma unroll
    for (int r = 0; r < C; ++r) {
        asm volatile("" : "+v"(x[r]));
        x[r] = mul_mod_barrett(x[r], x[r], bk);
        asm volatile("" : "+v"(x[r]));
    }
    __syncthreads();   // the image is reused: every wave is done reading the forward transform's exchanges
    f.inverse(x, itw_rb + (size_t)prime * pairs_per_prime, pc);
    if (live) {
#pragma unroll
        for (int r = 0; r < C; ++r) {
            if constexpr (NTS) __builtin_nontemporal_store(x[r], &c[base + f.tid + (uint32_t)r * T]);
            else c[base + f.tid + (uint32_t)r * T] = x[r];
        }
    }
}

template <int L, int R>
void build_table_t(const regblock_layout&, const uint64_t* tw, const uint64_t* pre, std::vector<ulonglong2>& out) {
    using G = rb2_geom<L, R>;
    const size_t start = out.size();
    out.resize(start + (size_t)G::table_pairs, make_ulonglong2(0, 0));
    for (int p = 0; p < G::NP; ++p) {
        const int rlo = G::rlo(p), hi = G::hi(p), H = G::H(p);
        ulonglong2* t = out.data() + start + (size_t)G::table_off(p);
        for (int j = 1; j < G::C; ++j) {
            int k = 0;
            while ((2 << k) <= j) ++k;
            const int o = j - (1 << k), rb_bit = R - 1 - k, b = rlo + rb_bit;
            if (b > hi) continue;  // stage belongs to an earlier pass (short last pass)
            const uint32_t m_local = 1u << (L - 1 - b);
            for (int h = 0; h < H; ++h) {
                const uint32_t idx = m_local + ((uint32_t)h << k) + (uint32_t)o;      // natural twiddle index m + i (ntt.cpp:298-300)
                // wave-uniform passes keep one column's C entries contiguous (wide scalar loads);
                // per-lane passes keep one entry's columns contiguous (coalesced vector loads)
                const size_t at = G::uniform_pass(p) ? (size_t)h * G::C + j : (size_t)j * H + h;
                t[at] = make_ulonglong2(tw[idx], pre[idx]);
            }
        }
    }
}

template <int L, int R, int PPB, int ARITH>
constexpr size_t rb2_lds_bytes() {
    using F = rb2_frame<L, R, (ARITH & 1) == 1, (ARITH >> 1)>;
    return (size_t)F::image_bytes * PPB;
}

template <int ARITH>
constexpr int rb2_arith_level() { return (ARITH & 1) ? ((((ARITH >> 1) & kOptLazy16) != 0) ? 2 : 1) : 0; }      // rb_entry::arith

template <int L, int R, int PPB, int ARITH, int MINW>
hipError_t launch_rb2_t(const plan_view& pv, const uint64_t* in, uint64_t* out, const frame_layout& fl, hipStream_t s) {
    using G = rb_geom<L, R>;
    const size_t lds = rb2_lds_bytes<L, R, PPB, ARITH>();
    dim3 grid((unsigned)((fl.batch + PPB - 1) / PPB), pv.num_primes);
    hipLaunchKernelGGL((fwd_rb2<L, R, PPB, ARITH, MINW>), grid, dim3(G::T * PPB), lds, s, in, out, pv.consts, pv.tw_rb,
                       pv.rb.pairs_per_prime, fl.batch, fl.prime_stride, fl.poly_stride, (uint32_t)(fl.lazy_out ? 1 : 0));
    return hipGetLastError();
}

template <int L, int R, int PPB, int ARITH, int MINW>
hipError_t launch_inv_rb2_t(const plan_view& pv, const uint64_t* in, const uint64_t* in2, uint64_t* out, const frame_layout& fl, hipStream_t s) {
    using G = rb_geom<L, R>;
    dim3 grid((unsigned)((fl.batch + PPB - 1) / PPB), pv.num_primes);
    const size_t lds = rb2_lds_bytes<L, R, PPB, ARITH>();
    hipLaunchKernelGGL((inv_rb2<L, R, PPB, ARITH, MINW>), grid, dim3(G::T * PPB), lds, s, in, in2, out, pv.consts,
                       pv.itw_rb, pv.rb.pairs_per_prime, fl.batch, fl.prime_stride, fl.poly_stride);
    return hipGetLastError();
}

template <int L, int R, int PPB, int ARITH, int MINW>
hipError_t launch_mul_rb2_t(const plan_view& pv, const uint64_t* a, const uint64_t* b, uint64_t* c, const frame_layout& fl, hipStream_t s) {
    using G = rb_geom<L, R>;
    dim3 grid((unsigned)((fl.batch + PPB - 1) / PPB), pv.num_primes);
    const size_t lds = rb2_lds_bytes<L, R, PPB, ARITH>();
    hipLaunchKernelGGL((polymul_rb2<L, R, PPB, ARITH, MINW>), grid, dim3(G::T * PPB), lds, s, a, b, c, pv.consts,
                       pv.tw_rb, pv.itw_rb, pv.rb.pairs_per_prime, fl.batch, fl.prime_stride, fl.poly_stride);
    return hipGetLastError();
}

template <int L, int R, int ARITH, int MINW>
hipError_t launch_mul_park_t(const plan_view& pv, const uint64_t* a, const uint64_t* b, uint64_t* c, const frame_layout& fl, hipStream_t s) {
    using G = rb_geom<L, R>;
    dim3 grid((unsigned)fl.batch, pv.num_primes);
    const size_t lds = rb2_lds_bytes<L, R, 1, ARITH>();
    if (a == b) {      // squaring: NTT(a) times itself in registers (the parked form would read its own parked words back when c aliases too)
        hipLaunchKernelGGL((polysquare_rb2<L, R, ARITH, MINW>), grid, dim3(G::T), lds, s, a, c, pv.consts,
                           pv.tw_rb, pv.itw_rb, pv.rb.pairs_per_prime, fl.batch, fl.prime_stride, fl.poly_stride);
        return hipGetLastError();
    }
    // the operand c aliases (if any) must be the one that is read completely before c's frame is written
    const uint64_t* first = (c == b) ? b : a;
    const uint64_t* second = (c == b) ? a : b;
    hipLaunchKernelGGL((polymul_rb2_park<L, R, ARITH, MINW>), grid, dim3(G::T), lds, s, first, second, c, pv.consts,
                       pv.tw_rb, pv.itw_rb, pv.rb.pairs_per_prime, fl.batch, fl.prime_stride, fl.poly_stride);
    return hipGetLastError();
}

template <int L, int R, int ARITH, int MINW>
hipError_t init_mul_park_t() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&polymul_rb2_park<L, R, ARITH, MINW>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)rb2_lds_bytes<L, R, 1, ARITH>());
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&polysquare_rb2<L, R, ARITH, MINW>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 (int)rb2_lds_bytes<L, R, 1, ARITH>());
    return e;
}

template <int L, int R, int PPB, int ARITH, int MINW>
hipError_t init_rb2_t() {
    const int bytes = (int)rb2_lds_bytes<L, R, PPB, ARITH>();
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fwd_rb2<L, R, PPB, ARITH, MINW>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&inv_rb2<L, R, PPB, ARITH, MINW>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&polymul_rb2<L, R, PPB, ARITH, MINW>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    return e;
}

// PPB frames per workgroup, every transform one workgroup per frame, the product with both forward results in registers
// (the n = 4096 defaults: R = 3, 8 waves/SIMD)
template <int L, int R, int PPB, int ARITH, int MINW>
constexpr rb_entry make_entry2(int id) {
    rb_entry e{id, L, R, PPB, MINW, (uint32_t)rb_geom<L, R>::table_pairs, rb2_lds_bytes<L, R, PPB, ARITH>(),
               &build_table_t<L, R>, &launch_rb2_t<L, R, PPB, ARITH, MINW>, &init_rb2_t<L, R, PPB, ARITH, MINW>, rb2_arith_level<ARITH>(),
               &launch_inv_rb2_t<L, R, PPB, ARITH, MINW>, &launch_mul_rb2_t<L, R, PPB, ARITH, MINW>};
    return e;
}

// forward kernel only (a plan's forward companion: rb_entry::fwd_companion)
template <int L, int R, int PPB, int ARITH, int MINW>
hipError_t init_rb2_fwd_only_t() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&fwd_rb2<L, R, PPB, ARITH, MINW>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)rb2_lds_bytes<L, R, PPB, ARITH>());
}
template <int L, int R, int ARITH, int MINW>
constexpr rb_entry make_entry_single_fwd(int id) {
    rb_entry e{id, L, R, 1, MINW, (uint32_t)rb_geom<L, R>::table_pairs, rb2_lds_bytes<L, R, 1, ARITH>(),
               &build_table_t<L, R>, &launch_rb2_t<L, R, 1, ARITH, MINW>, &init_rb2_fwd_only_t<L, R, 1, ARITH, MINW>, rb2_arith_level<ARITH>(), nullptr, nullptr};
    return e;
}

// workgroups the current device holds at once at MINW waves per SIMD (the loop kernels' grid)
template <int L, int R, int MINW>
hipError_t resident_workgroups(unsigned* out) {
    int dev = 0, cus = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (e == hipSuccess) *out = (unsigned)cus * (unsigned)(MINW * 256 / rb_geom<L, R>::T);
    return e;
}

template <int L, int R, int ARITH, int MINW>
hipError_t launch_inv_rb2_loop_t(const plan_view& pv, const uint64_t* in, const uint64_t* in2, uint64_t* out, const frame_layout& fl, hipStream_t s) {
    using G = rb_geom<L, R>;
    unsigned resident = 0;
    hipError_t e = resident_workgroups<L, R, MINW>(&resident);
    if (e != hipSuccess) return e;
    const uint64_t total = fl.batch * pv.num_primes;
    if (total >= (1ull << 31)) return hipErrorInvalidValue;
    const unsigned grid = (unsigned)(total < resident ? total : resident);
    const size_t lds = rb2_lds_bytes<L, R, 1, ARITH>();
    hipLaunchKernelGGL((inv_rb2_loop<L, R, ARITH, MINW>), dim3(grid), dim3(G::T), lds, s, in, in2, out, pv.consts, pv.itw_rb,
                       pv.rb.pairs_per_prime, (uint32_t)fl.batch, (uint32_t)total, fl.prime_stride, fl.poly_stride);
    return hipGetLastError();
}

// A launch recorded into a hipGraph would keep its ticket pair for as long as the graph lives, while eager launches of the same
// stream keep drawing from it; captured launches therefore take the fixed-stride loop kernel, which carries no state.
inline bool stream_is_capturing(hipStream_t s) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    return hipStreamIsCapturing(s, &st) == hipSuccess && st == hipStreamCaptureStatusActive;
}

template <int L, int R, int ARITH, int MINW>
hipError_t launch_inv_rb2_dloop_t(const plan_view& pv, const uint64_t* in, const uint64_t* in2, uint64_t* out, const frame_layout& fl, hipStream_t s) {
    using G = rb_geom<L, R>;
    if (stream_is_capturing(s)) return launch_inv_rb2_loop_t<L, R, ARITH, MINW>(pv, in, in2, out, fl, s);
    unsigned resident = 0;
    hipError_t e = resident_workgroups<L, R, MINW>(&resident);
    if (e != hipSuccess) return e;
    const uint64_t total = fl.batch * pv.num_primes;
    if (total >= (1ull << 31)) return hipErrorInvalidValue;
    uint32_t* ticket = pv.ticket(s);      // taken last: from here on the launch is issued and ticket_done() follows it
    if (!ticket) return launch_inv_rb2_loop_t<L, R, ARITH, MINW>(pv, in, in2, out, fl, s);      // no pair provably free: stateless form
    const unsigned grid = (unsigned)(total < resident ? total : resident);
    const size_t lds = rb2_lds_bytes<L, R, 1, ARITH>() + 16;
    hipLaunchKernelGGL((inv_rb2_dloop<L, R, ARITH, MINW>), dim3(grid), dim3(G::T), lds, s, in, in2, out, pv.consts, pv.itw_rb,
                       pv.rb.pairs_per_prime, (uint32_t)fl.batch, (uint32_t)total, fl.prime_stride, fl.poly_stride, ticket);
    e = hipGetLastError();
    pv.ticket_done(s, ticket);      // an event behind the launch: when it has fired the pair is idle (and zero) and the plan may hand it to another stream
    return e;
}

// the streamed single-frame kernels (reg_s<n>.hip): one frame in registers per workgroup at any time (R = 5: a second frame cannot
// be held): forward, inverse, and the fused product by polymul_rb2_park / polysquare_rb2
template <int L, int R, int ARITH, int MINW>
hipError_t init_rb2_single_t() {
    const int bytes = (int)rb2_lds_bytes<L, R, 1, ARITH>();
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fwd_rb2<L, R, 1, ARITH, MINW>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&inv_rb2<L, R, 1, ARITH, MINW>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) e = init_mul_park_t<L, R, ARITH, MINW>();
    return e;
}
template <int L, int R, int ARITH, int MINW>
constexpr rb_entry make_entry_single(int id) {
    rb_entry e{id, L, R, 1, MINW, (uint32_t)rb_geom<L, R>::table_pairs, rb2_lds_bytes<L, R, 1, ARITH>(),
               &build_table_t<L, R>, &launch_rb2_t<L, R, 1, ARITH, MINW>, &init_rb2_single_t<L, R, ARITH, MINW>, rb2_arith_level<ARITH>(),
               &launch_inv_rb2_t<L, R, 1, ARITH, MINW>, &launch_mul_park_t<L, R, ARITH, MINW>};
    return e;
}

// entry e with forward calls of its plans routed to the forward-only entry `id` (rb_entry::fwd_companion)
constexpr rb_entry with_fwd_companion(rb_entry e, int id, uint32_t min_frames = 0) {
    e.fwd_companion = id;
    e.fwd_companion_min_frames = min_frames;
    return e;
}

// ... with the fused product by polymul_rb2 (both forward results in registers) at MULW waves per SIMD: small frames, where two
// frames of 2^R coefficients still fit the register budget and the parked product's round trip through c's frame costs more
template <int L, int R, int ARITH, int MINW, int MULW>
hipError_t init_rb2_single_mul2_t() {
    const int bytes = (int)rb2_lds_bytes<L, R, 1, ARITH>();
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fwd_rb2<L, R, 1, ARITH, MINW>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&inv_rb2<L, R, 1, ARITH, MINW>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&polymul_rb2<L, R, 1, ARITH, MULW>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    return e;
}
template <int L, int R, int ARITH, int MINW, int MULW>
constexpr rb_entry make_entry_single_mul2(int id) {
    rb_entry e = make_entry_single<L, R, ARITH, MINW>(id);
    e.init = &init_rb2_single_mul2_t<L, R, ARITH, MINW, MULW>;
    e.launch_mul = &launch_mul_rb2_t<L, R, 1, ARITH, MULW>;
    return e;
}

// ... with the inverse by the ticket-drawing loop kernel (captured launches and streams without a ticket pair take the fixed-stride form)
template <int L, int R, int ARITH, int MINW>
hipError_t init_rb2_single_invloop_t() {
    hipError_t e = init_rb2_single_t<L, R, ARITH, MINW>();
    const int bytes = (int)rb2_lds_bytes<L, R, 1, ARITH>();
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&inv_rb2_loop<L, R, ARITH, MINW>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&inv_rb2_dloop<L, R, ARITH, MINW>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes + 16);
    return e;
}
template <int L, int R, int ARITH, int MINW>
constexpr rb_entry make_entry_single_invloop(int id) {
    rb_entry e = make_entry_single<L, R, ARITH, MINW>(id);
    e.init = &init_rb2_single_invloop_t<L, R, ARITH, MINW>;
    e.launch_inv = &launch_inv_rb2_dloop_t<L, R, ARITH, MINW>;
    return e;
}

}  // namespace AGX_TU
}  // namespace agx
