// rb_kernels.hpp -- register-blocked gfx950 kernels (templates) and their launch glue.  Included by the
// reg_*.hip translation units, each of which instantiates one group of registry entries.
//
// Every thread keeps 2^R coefficients in VGPRs and runs R butterfly stages per pass with no memory
// traffic; passes exchange through one padded LDS image.  This is the throughput path (n >= 1024).
// No MFMA: this is 64-bit integer modular arithmetic (v_mad_u64_u32 / v_mul_hi_u32), bounded by VALU
// integer multiply issue and HBM bandwidth.
#pragma once
#include "rb_registry.hpp"
#include "modarith.hpp"

#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include <utility>

#ifndef AGX_POLYMUL_MAXW
#define AGX_POLYMUL_MAXW 5
#endif
#ifndef AGX_TU
#error "define AGX_TU (a per-translation-unit namespace name) before including rb_kernels.hpp"
#endif

namespace agx {
// Kernels live in a namespace named after the translation unit that instantiates them: two registry groups may
// instantiate the same template (e.g. the resident 8192-point kernel of the split and pair entries), and every
// code object must register its own copy under its own name.
namespace AGX_TU {



// ---------------------------------------------------------------------------------------
// helpers
// ---------------------------------------------------------------------------------------
template <int B, int E, typename F>
__device__ __host__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(std::forward<F>(f));
    }
}

extern __shared__ __attribute__((aligned(16))) unsigned char agx_dyn_lds[];

// ---------------------------------------------------------------------------------------
// register-blocked forward kernel.
//
// n = 2^L per workgroup-resident (sub-)transform, C = 2^R coefficients per thread, T = n / C
// threads per frame, PPB frames per workgroup.  Coefficient index bits are processed from the
// top (gap n/2) down to bit 0, R at a time:
//   pass p keeps index bits [rlo+R-1 : rlo] in the register number r, rlo = max(L - R(p+1), 0):
//       e(tid, r) = (tid & (2^rlo - 1)) | r << rlo | (tid >> rlo) << (rlo + R)
//   and runs the stages whose gap bit b lies in [L-1-Rp : rlo] entirely in registers.
// Between passes the coefficients cross threads through one LDS slab (index padded by one
// element per 16 to spread the strided pass layouts over the banks).
// The twiddle for registers (r0, r0 | 2^rb) at gap bit b = rlo + rb is natural index
//   2^(L-1-b) + ((tid >> rlo) << k) + (r0 >> (rb+1)),   k = R-1-rb,
// stored in the pass table at [(2^k + (r0 >> (rb+1))) * H + (tid >> rlo)], H = threads/2^rlo:
// consecutive lanes read consecutive 16-byte {w,w'} pairs; in pass 0 (H = 1) the address is
// wave-uniform and the loads are scalar.
// ---------------------------------------------------------------------------------------
template <int L, int R>
struct rb_geom {
    static constexpr int C = 1 << R;
    static constexpr int T = 1 << (L - R);
    static constexpr int NP = (L + R - 1) / R;
    static constexpr int rlo(int p) { return (L - R * (p + 1)) > 0 ? (L - R * (p + 1)) : 0; }
    static constexpr int hi(int p) { return L - 1 - R * p; }
    static constexpr int H(int p) { return 1 << (L - R - rlo(p)); }           // distinct (tid >> rlo)
    static constexpr int table_off(int p) { return p == 0 ? 0 : table_off(p - 1) + C * H(p - 1); }
    static constexpr int table_pairs = table_off(NP);
    static constexpr int lds_elems = (1 << L) + (1 << (L - 4));
};

__device__ __forceinline__ constexpr uint32_t lds_pad(uint32_t e) { return e + (e >> 4); }

template <int L, int R, int PPB, bool STAGE_OUT, int MINW>
__global__ void __launch_bounds__((1 << (L - R)) * PPB, MINW)
fwd_regblock(const uint64_t* __restrict__ in, uint64_t* __restrict__ out,
             const prime_consts* __restrict__ consts, const twpair* __restrict__ tw_rb,
             uint32_t pairs_per_prime, uint32_t split_log, uint64_t frames_x,
             int64_t prime_stride, int64_t poly_stride) {
    using G = rb_geom<L, R>;
    constexpr int C = G::C, T = G::T, NP = G::NP;
    uint64_t* lds = reinterpret_cast<uint64_t*>(agx_dyn_lds);

    const uint32_t tid = threadIdx.x & (T - 1);
    const uint32_t slot = threadIdx.x / T;
    uint64_t fx = (uint64_t)blockIdx.x * PPB + slot;      // (poly, blk) flattened
    const bool live = fx < frames_x;
    if (!live) fx = frames_x - 1;                         // keep every thread on the barriers
    const uint32_t prime = blockIdx.y;
    const uint64_t poly = fx >> split_log;
    const uint32_t blk = (uint32_t)(fx & ((1u << split_log) - 1u));
    const uint64_t q = consts[prime].q, q2 = q << 1;
    const twpair* tbl = tw_rb + (size_t)prime * pairs_per_prime;
    const int64_t base = (int64_t)prime * prime_stride + (int64_t)poly * poly_stride + ((int64_t)blk << L);
    uint64_t* slab = lds + (size_t)slot * G::lds_elems;

    uint64_t x[C];
#pragma unroll
    for (int r = 0; r < C; ++r) x[r] = in[base + tid + (uint32_t)r * T];

    static_for<0, NP>([&](auto P) {
        constexpr int p = P;
        constexpr int rlo = G::rlo(p), hi = G::hi(p), H = G::H(p);
        const uint32_t low = tid & ((1u << rlo) - 1u), high = tid >> rlo;
        const uint32_t ebase = low | (high << (rlo + R));
        if constexpr (p > 0) {
#pragma unroll
            for (int r = 0; r < C; ++r) x[r] = slab[lds_pad(ebase | ((uint32_t)r << rlo))];
        }
        // column of this thread in the pass table (blk selects the sub-transform's columns)
        const twpair* col = tbl + G::table_off(p) * (1u << split_log) + (size_t)blk * H + (H == 1 ? 0u : high);
        const uint32_t hstride = (uint32_t)H << split_log;
        static_for<0, hi - rlo + 1>([&](auto S) {
            constexpr int rb = (hi - rlo) - S;       // register bit of this stage, descending
            constexpr int k = R - 1 - rb;
            constexpr bool last_stage = (rlo + rb) == 0;
#pragma unroll
            for (int r0 = 0; r0 < C; ++r0) {
                if ((r0 >> rb) & 1) continue;
                const int r1 = r0 | (1 << rb);
                const int j = (1 << k) + (r0 >> (rb + 1));
                const twpair w = col[(size_t)j * hstride];
                ct_butterfly(x[r0], x[r1], w.x, w.y, q, q2);
                if constexpr (last_stage) {
                    x[r0] = reduce_4q(x[r0], q, q2);
                    x[r1] = reduce_4q(x[r1], q, q2);
                }
            }
        });
        if constexpr (p < NP - 1) {
            if constexpr (p > 0) __syncthreads();   // everyone has finished reading the slab
#pragma unroll
            for (int r = 0; r < C; ++r) slab[lds_pad(ebase | ((uint32_t)r << rlo))] = x[r];
            __syncthreads();
        }
    });

    // last pass has rlo = 0: thread holds C consecutive coefficients starting at tid * C
    if constexpr (STAGE_OUT) {
        if constexpr (NP > 1) __syncthreads();
#pragma unroll
        for (int r = 0; r < C; ++r) slab[lds_pad((tid << R) | (uint32_t)r)] = x[r];
        __syncthreads();
        if (live) {
#pragma unroll
            for (int r = 0; r < C; ++r) out[base + tid + (uint32_t)r * T] = slab[lds_pad(tid + (uint32_t)r * T)];
        }
    } else if (live) {
        ulonglong2* o = reinterpret_cast<ulonglong2*>(out + base + ((size_t)tid << R));
#pragma unroll
        for (int r = 0; r < C; r += 2) o[r >> 1] = make_ulonglong2(x[r], x[r + 1]);
    }
}

// ---------------------------------------------------------------------------------------
// second-generation register-blocked kernels (the throughput path): forward, inverse and fused
// polynomial product share rb2_frame below.
//
// Same pass structure as fwd_regblock; what changed, each item from a measurement:
//  * hand-selected butterfly forms (modarith.hpp): 15-19 VALU instead of ~32, because every VOP3
//    integer op issues at quarter/half rate on gfx950 and the kernel is VALU bound;
//  * R = 3 (8 coefficients per thread, <= 64 VGPRs): 8 waves/SIMD issue multiplies 34 % faster than 4;
//  * passes whose twiddle column depends only on the wave index (rlo >= 6) read their twiddles
//    with wide scalar loads into SGPRs: no VGPRs, no VALU, no vector-memory traffic for them;
//  * an exchange that only moves coefficients between lanes of the same wave needs no
//    workgroup barrier (one wave's LDS operations execute in program order); which exchanges
//    those are is decided at compile time by exchange_is_wave_local(): one s_barrier per frame;
//  * LDS image padded by one word per 16 (default) so every exchange access is thread base +
//    immediate offset; the XOR-swizzled image (conflict-free, exactly 8n bytes) is kept as an option;
//  * results leave through the LDS image as coalesced stores (each wave owns a contiguous
//    chunk of the frame after the first exchange); direct 16-byte strided stores measured 5 % slower.
// ---------------------------------------------------------------------------------------
template <int L, int R>
struct rb2_geom : rb_geom<L, R> {
    using G = rb_geom<L, R>;
    static constexpr uint32_t elem(int p, uint32_t tid, uint32_t r) {
        const int rlo = G::rlo(p);
        return (tid & ((1u << rlo) - 1u)) | (r << rlo) | ((tid >> rlo) << (rlo + R));
    }
    static constexpr uint32_t owner(int p, uint32_t e) {
        const int rlo = G::rlo(p);
        return (e & ((1u << rlo) - 1u)) | ((e >> (rlo + R)) << rlo);
    }
    // does the exchange between pass p and p+1 keep every coefficient inside one wave?
    static constexpr bool exchange_is_wave_local(int p) {
        for (uint32_t tid = 0; tid < (uint32_t)G::T; ++tid)
            for (uint32_t r = 0; r < (uint32_t)G::C; ++r)
                if ((owner(p + 1, elem(p, tid, r)) >> 6) != (tid >> 6)) return false;
        return true;
    }
    // is pass p's twiddle column the same for every lane of a wave?  Either the column index (tid >> rlo) only changes from wave to
    // wave (rlo >= 6), or the pass has a single column (H = 1: pass 0 of a whole frame; with fewer than 64 threads per frame -- the
    // wave-packed kernels of wp_kernels.hpp -- that is the only way).  Such passes read their entries with scalar loads into SGPRs.
    static constexpr bool uniform_pass(int p) { return G::rlo(p) >= 6 || G::H(p) == 1; }
    // after the last pass, does every wave hold one contiguous block of 64*C coefficients?
    static constexpr bool last_pass_wave_contiguous() { return G::rlo(G::NP - 1) == 0 && G::T >= 64; }
};

// LDS image index of coefficient e: XOR-swizzle of the low five bits (one 256-byte bank row of
// ds_read_b64) by bits 5..8, so that every lane pattern of the passes spreads over the banks
__device__ __forceinline__ constexpr uint32_t lds_swz(uint32_t e) {
    return e ^ ((e >> 5) & 7u) ^ (((e >> 6) & 3u) << 3);
}

// wave-uniform table entry through the constant address space: a scalar load into SGPRs
__device__ __forceinline__ twpair load_uniform(const twpair* p) {
    typedef const uint64_t __attribute__((address_space(4))) * const_ptr;
    const_ptr c = (const_ptr)(uintptr_t)p;
    twpair r;
    r.x = c[0];
    r.y = c[1];
    return r;
}

// Orders one wave's LDS stores before its following LDS loads of words OTHER lanes of the wave wrote (a wave-local
// exchange).  __builtin_amdgcn_wave_barrier() alone only stops the scheduler; the wavefront-scope release/acquire pair
// is what forbids the compiler to move the loads above the stores (both lower to nothing on gfx950: one wave's LDS
// operations execute in program order).
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// option bits of the second-generation kernels (A/B switches; the tuned defaults are in the registry)
constexpr int kOptPad = 1;       // padded LDS image, exchanges addressed base + immediate offset
constexpr int kOptSelect = 2;    // conditional subtract by compare + select instead of sign mask
constexpr int kOptTwAhead = 32;   // per-lane passes: first three table entries fetched one pass early, the rest at pass start
constexpr int kOptLazy16 = 16;   // q <= 2^60: 16q-lazy forward butterflies (conditional subtract on 5 of 12 stages)
constexpr int kOptPrio = 128;    // s_setprio 3 while a wave issues its frame loads (and, with kOptPrioStore, its stores)
constexpr int kOptPrioStore = 256;
constexpr int kOptPrioBarrier = 512;   // with kOptPrio: stay at priority until the frame's one s_barrier has been passed
constexpr int kOptLazyInv = 2048;      // with kOptLazy16: inverse butterflies keep sums up to 16q (26 instead of 48 conditional subtracts per thread at n=4096)
constexpr int kOptTwAheadInv = 4096;   // inverse: the next per-lane pass's first-stage twiddles (entries 4..7) fetched during the current pass's last stage
constexpr int kOptAblateTw = 8192;     // timing only (wrong results): every lane reads column 0 of the per-lane tables (L1-resident)
constexpr int kOptAblateHbm = 16384;   // timing only (wrong results): every workgroup transforms frame 0 of its prime (L2-resident)
constexpr int kOptAblateLdOnly = 32768, kOptAblateStOnly = 65536;   // with kOptAblateHbm: only the loads / only the stores go to the hot frame
constexpr int kOptNtLoad = 131072, kOptNtStore = 262144;   // non-temporal frame loads / result stores (data touched once)
constexpr int kOptScalarBase = 1024;   // frame loads as (uniform pointer per register) + lane offset: no per-load VALU address arithmetic
constexpr int kOptEstReduce = 524288;   // with kOptLazy16: tail-free subtract schedule + quotient-estimate final reduction
constexpr int kOptSplitWord = 1048576;  // forward only: exchanges move the low and the high 32-bit words in turn through an image of HALF
                                        // the size (4n bytes) -- the LDS footprint that lets R = 4 workgroups of 4 waves fill a CU (VERDICT r01 #1 ii)
constexpr int kOptMulLoCross = 2097152; // 16q-lazy forward butterflies: the four cross products as 32-bit v_mul_lo_u32 (energy A/B, tools/microbench pwr)
constexpr int kOptInvTwFirst = 4;       // inverse: the first (per-lane) pass's first-stage twiddles are requested right behind the frame loads, ahead of the
                                        // LDS staging of the frame (otherwise their L2 latency starts only after the HBM latency of the frame has been paid)
constexpr int kOptInvTwFirstAll = 8;    // with kOptInvTwFirst: every entry of that pass, not only its first stage's
constexpr int kOptInvPrioTail = 1 << 22;   // inverse A/B: s_setprio 3 from the cross-wave exchange to the end (finish and free the slot)
constexpr int kOptInvPrioAsc = 1 << 23;    // inverse A/B: priority rises pass by pass (0,1,2,3)
constexpr int kOptInvPrioDesc = 1 << 24;   // inverse A/B: priority falls pass by pass (3,2,1,0): the workgroup's laggards catch up before the barrier
constexpr int kOptStreamTw = 1 << 25;      // twiddles streamed in chunks of four table entries, one chunk requested ahead of the one in use, scheduling fenced per chunk:
                                           // bounds the registers a pass's table entries occupy (R = 5: 31 entries per pass would be 124 SGPRs / VGPRs if fetched up front)
constexpr int kOptPinBf = 1 << 26;         // with kOptStreamTw: butterflies are pinned in program order (their operands pass through ordered empty asm statements), so the
                                           // instruction selector cannot start the partial products of the whole stage at once -- what takes an R = 5 pass from ~205 VGPRs to
                                           // the 128 a 1024-thread workgroup may use; one wave cannot issue faster than one VALU per ~8 clocks anyway (ILP buys nothing there)
constexpr int kOptSaddrTw = 1 << 27;       // per-lane table entries addressed as wave-uniform base (SGPRs) + 32-bit lane index (always on with kOptStreamTw)
constexpr int kOptFinalMode = 1 << 28;     // the last stage branches once per stage on the final-reduction mode instead of once per coefficient (always on with kOptStreamTw)
constexpr int kOptStreamCh1 = 1 << 29;     // with kOptStreamTw: one table entry per chunk instead of two (8 fewer VGPRs: what the loop kernels need to stay out of scratch)
constexpr int kOptTrace = 64;    // diagnostics (tools/timeline.py): every wave records s_memtime at 12 phase boundaries

// where the kOptTrace kernels write: [wave][16] words, set through agx_ntt_debug_set_trace_buffer
__device__ uint64_t* g_trace_buf = nullptr;     // one copy per translation unit (namespace AGX_TU); only reg_diag.hip uses it
__device__ uint64_t g_trace_waves = 0;

// call-backs a kernel can thread into the forward passes (the streaming kernel uses both)
struct rb2_no_hooks {
    template <int p> __device__ __forceinline__ void before_image_write() const {}
    template <int p> __device__ __forceinline__ void after_twiddle_issue() const {}
    template <int p> __device__ __forceinline__ void after_exchange_sync() const {}
    template <int p, bool last_stage_of_pass, int b> __device__ __forceinline__ void after_butterfly() const {}
};

// per-frame state shared by the second-generation kernels
template <int L, int R, bool FAST, int OPT = 0, int S0 = 0>   // S0: stages already done before the resident transform
struct rb2_frame {
    using G = rb2_geom<L, R>;
    static constexpr int C = G::C, T = G::T, NP = G::NP;
    static constexpr bool PAD = (OPT & kOptPad) != 0, SEL = (OPT & kOptSelect) != 0, LAZY16 = FAST && (OPT & kOptLazy16) != 0;
    static constexpr bool TWA = (OPT & kOptTwAhead) != 0 && R >= 3;
    static constexpr bool TRACE = (OPT & kOptTrace) != 0;
    static constexpr bool PRIO = (OPT & kOptPrio) != 0, PRIO_STORE = (OPT & kOptPrioStore) != 0;
    static constexpr bool PRIO_BARRIER = PRIO && (OPT & kOptPrioBarrier) != 0, SCALAR_BASE = (OPT & kOptScalarBase) != 0;
    static constexpr bool LAZY_INV = LAZY16 && (OPT & kOptLazyInv) != 0;
    static constexpr bool NT_LOAD = (OPT & kOptNtLoad) != 0;
    static constexpr bool TWA_INV = (OPT & kOptTwAheadInv) != 0 && R == 3;
    static constexpr bool EST = LAZY16 && SEL && (OPT & kOptEstReduce) != 0;
    static constexpr bool SPLIT = (OPT & kOptSplitWord) != 0;
    static constexpr bool STREAM_TW = (OPT & kOptStreamTw) != 0;
    static constexpr bool SADDR_TW = STREAM_TW || (OPT & kOptSaddrTw) != 0, FINAL_MODE = STREAM_TW || (OPT & kOptFinalMode) != 0;
    static constexpr bool INV_TWF = (OPT & kOptInvTwFirst) != 0 && G::rlo(NP - 1) < 6, INV_TWF_ALL = INV_TWF && (OPT & kOptInvTwFirstAll) != 0;
    static_assert(!SPLIT || PAD, "the split-word image uses the padded index");
    static_assert(!EST || lazy16_tailfree::valid(S0 + L), "tail-free schedule must keep every stage within 16q");
    mutable uint64_t ts[12];
    uint64_t trace_wave = ~0ull;   // row of the trace buffer (default: launch-wide wave number)
    bool trace_wait_stores = true; // stamp 11 after the stores have retired (not in the streaming kernel: that would drain its prefetch)
    // phase stamp I, ordered after `anchor` is available and before anything that uses it afterwards
    template <int I>
    __device__ __forceinline__ void stamp(uint64_t& anchor) const {
        if constexpr (TRACE) {
            uint64_t t;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t), "+v"(anchor) : : "memory");
            ts[I] = t;
        }
    }
    __device__ __forceinline__ void trace_flush() const {
        if constexpr (TRACE) {
            const uint64_t wave = trace_wave != ~0ull ? trace_wave
                                                      : ((uint64_t)blockIdx.y * gridDim.x + blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6);
            if ((threadIdx.x & 63u) == 0 && g_trace_buf != nullptr && wave < g_trace_waves) {
                uint64_t* dst = g_trace_buf + wave * 16;
#pragma unroll
                for (int i = 0; i < 12; ++i) dst[i] = ts[i];
                dst[12] = __builtin_amdgcn_s_getreg((31 << 11) | 4);     // HW_ID: wave, simd, cu, sh, se
                dst[13] = __builtin_amdgcn_s_getreg((31 << 11) | 20);    // XCC_ID
            }
        }
    }
    // a per-lane pass with all R stages: the shape the look-ahead twiddle fetch handles
    static constexpr bool lane_full_pass(int p) { return p >= 0 && p < NP && G::rlo(p) < 6 && G::hi(p) - G::rlo(p) + 1 == R; }
    // pad: one image word per 16 coefficients for the 64-bit image (ds_read_b64: 64 banks), one per 32 for the split-word image, whose
    // 32-bit accesses see 32 banks per group of 32 lanes: with it every exchange pattern of the R = 5 kernels is conflict-free, with one
    // per 16 every one of them was two-way conflicted (SQ_LDS_BANK_CONFLICT 45 % of the LDS cycles, profiles/r03c_fwd4096_summary.md)
    static constexpr int PADS = (OPT & kOptSplitWord) != 0 ? 5 : 4;
    static constexpr uint32_t slab_elems = PAD ? (1u << L) + (1u << (L - PADS)) : (1u << L);
    static constexpr uint32_t image_bytes = slab_elems * (((OPT & kOptSplitWord) != 0) ? 4u : 8u);   // one frame's LDS image
    // image word of coefficient e; both forms are additive over disjoint bit fields, which is what
    // lets an exchange address register r as (thread base) combined with a compile-time constant
    static __device__ __forceinline__ constexpr uint32_t img(uint32_t e) { return PAD ? e + (e >> PADS) : lds_swz(e); }
    static __device__ __forceinline__ constexpr uint32_t join(uint32_t base, uint32_t delta) { return PAD ? base + delta : base ^ delta; }
    uint32_t tid, blk, split_log;
    bool lazy_out = false;   // forward only: leave results in [0,4q) (wave-uniform)
    uint64_t* slab;
    bf_consts k;
    final_consts fc;

    __device__ __forceinline__ void init_consts(uint64_t q, uint64_t est = 0) {
        k.est_inv = __uint_as_float((uint32_t)est);
        k.q = q;
        k.nq = 0 - q;
        k.m = FAST ? (q << 2) : (q << 1);
        k.nm = opaque_sgpr64(0 - k.m);
        k.one_a = opaque_one<0>();
        k.one_b = opaque_one<1>();
        fc.q2 = q << 1;
        fc.nq2 = opaque_sgpr64(0 - fc.q2);
        fc.q1 = q;
        fc.nq1 = opaque_sgpr64(0 - fc.q1);
        fc.q8 = q << 3;
        fc.nq8 = opaque_sgpr64(0 - fc.q8);
    }

    // forward butterfly number `stage` of the whole transform in this frame's arithmetic
    template <int stage>
    __device__ __forceinline__ void butterfly(uint64_t& a, uint64_t& b, const twpair& w) const {
        if constexpr (EST) ct_butterfly_lazy16<SEL, lazy16_tailfree::subtracts(stage, S0 + L), false, (OPT & kOptMulLoCross) ? 1 : 0>(a, b, w.x, w.y, k, fc);
        else if constexpr (LAZY16) ct_butterfly_lazy16<SEL, lazy16_schedule::subtracts(stage), stage == S0 + L - 1>(a, b, w.x, w.y, k, fc);
        else if constexpr (FAST) ct_butterfly_fast<SEL>(a, b, w.x, w.y, k);
        else ct_butterfly_exact(a, b, w.x, w.y, k);
    }

    // final reduction of one coefficient.  MODE bit 0: lazy outputs; bit 1: quotient estimate (EST kernels, q >= 2^58).  The caller branches ONCE per group of butterflies on the wave-uniform conditions (lazy_out, est_inv) and
    // passes the outcome here as a constant, so the last stage is straight-line code (a branch per coefficient splits it into dozens of
    // basic blocks, which costs registers: R = 5 kernels went from 204 to the VGPRs of the arithmetic proper).
    template <int MODE>
    __device__ __forceinline__ uint64_t final_reduce(uint64_t v) const {
        if constexpr (MODE == 4) {      // undecided: branch per coefficient (the old form)
            if constexpr (EST) return reduce_final_est<SEL>(v, k, fc, lazy_out);
            else if constexpr (LAZY16) return reduce_final_lazy16<SEL>(v, k, fc, lazy_out);
            else return reduce_final<FAST, SEL>(v, k, fc, lazy_out);
        }
        if constexpr (EST) return reduce_final_est<SEL, (MODE & 2) ? 1 : 0>(v, k, fc, (MODE & 1) != 0);
        else if constexpr (LAZY16) return reduce_final_lazy16<SEL>(v, k, fc, (MODE & 1) != 0);
        else return reduce_final<FAST, SEL>(v, k, fc, (MODE & 1) != 0);
    }
    // run body(integral_constant<int, MODE>) under the wave-uniform choice of the final-reduction mode
    template <class Body>
    __device__ __forceinline__ void with_final_mode(Body&& body) const {
        if constexpr (!FINAL_MODE) {
            body(std::integral_constant<int, 4>{});      // the kernels tuned at the 64-VGPR edge keep the per-coefficient form
            return;
        }
        if (lazy_out) {
            // EST kernels: lazy outputs of q >= 2^58 still take the estimate (MODE 1 inside reduce_final_est keys on est_inv itself)
            if constexpr (EST) {
                if (k.est_inv != 0.0f) body(std::integral_constant<int, 3>{});
                else body(std::integral_constant<int, 1>{});
            } else body(std::integral_constant<int, 1>{});
        } else {
            if constexpr (EST) {
                if (k.est_inv != 0.0f) body(std::integral_constant<int, 2>{});
                else body(std::integral_constant<int, 0>{});
            } else body(std::integral_constant<int, 0>{});
        }
    }

    // image word of (pass p, register r) for this thread
    template <int p>
    __device__ __forceinline__ uint32_t sbase() const {
        constexpr int rlo = G::rlo(p);
        return img((tid & ((1u << rlo) - 1u)) | ((tid >> rlo) << (rlo + R)));
    }

    // fetch the twiddles of pass p: scalar loads when the column is wave-uniform
    template <int p>
    struct tw_src {
        twpair tw[C];
        const twpair* col;
        uint32_t hstride;
    };
    template <int p>
    __device__ __forceinline__ void fetch(tw_src<p>& t, const twpair* tbl) const {
        constexpr int rlo = G::rlo(p), H = G::H(p);
        const uint32_t high = tid >> rlo;
        if constexpr (G::uniform_pass(p)) {
            const uint32_t hcol = (uint32_t)__builtin_amdgcn_readfirstlane((int)high);
            const twpair* ucol = tbl + G::table_off(p) * (1u << split_log) + ((size_t)blk * H + hcol) * C;
            if constexpr (STREAM_TW) {
                t.col = ucol;      // entries are read chunk by chunk (stream_load)
            } else {
#pragma unroll
                for (int j = 1; j < C; ++j) t.tw[j] = load_uniform(ucol + j);      // merged into wide s_loads
                t.col = nullptr;
            }
            t.hstride = 0;
        } else {
            t.col = tbl;
            t.hstride = (uint32_t)H << split_log;
        }
    }
    // Entry j of per-lane pass p's table for this lane, addressed as (wave-uniform base of the entry, in SGPRs) + (32-bit lane index):
    // the load takes the saddr form and no entry needs a 64-bit VGPR address of its own (R = 5: 31 entries per pass would be 62 VGPRs).
    template <int p>
    __device__ __forceinline__ twpair lane_entry(const twpair* tbl, int j) const {
        constexpr int rlo = G::rlo(p), H = G::H(p);
        if constexpr (!SADDR_TW) {
            const twpair* col = tbl + G::table_off(p) * (1u << split_log) + (size_t)blk * H + ((OPT & kOptAblateTw) ? 0u : (tid >> rlo));
            return col[(size_t)j * ((uint32_t)H << split_log)];
        }
        const twpair* base = tbl + G::table_off(p) * (1u << split_log) + (size_t)blk * H + (size_t)j * ((uint32_t)H << split_log);
        const uint32_t lane = (OPT & kOptAblateTw) ? 0u : (tid >> rlo);
        return base[lane];
    }
    template <int p>
    __device__ __forceinline__ twpair twiddle(const tw_src<p>& t, int j) const {
        if constexpr (G::uniform_pass(p)) return t.tw[j];
        else return lane_entry<p>(t.col, j);
    }

    template <int p>
    __device__ __forceinline__ void image_read(uint64_t (&x)[C]) const {
        const uint32_t sb = sbase<p>();
        static_for<0, C>([&](auto Rr) { constexpr int r = Rr; x[r] = slab[join(sb, img((uint32_t)r << G::rlo(p)))]; });
    }
    template <int p>
    __device__ __forceinline__ void image_write(const uint64_t (&x)[C]) const {
        const uint32_t sb = sbase<p>();
        static_for<0, C>([&](auto Rr) { constexpr int r = Rr; slab[join(sb, img((uint32_t)r << G::rlo(p)))] = x[r]; });
    }
    // order the read side of the exchange between passes p and p+1 (either direction)
    template <int p>
    __device__ __forceinline__ void exchange_sync() const {
        if constexpr (!G::exchange_is_wave_local(p)) __syncthreads();
        else wave_lds_sync();
    }

    // SPLIT: the whole exchange between passes p and p+1 through a 32-bit image -- low words out, low words in, high words
    // out, high words in.  The image holds n words of 4 bytes; every lane is active in every step and no register is
    // needed beyond x.  Three synchronisations instead of one: the middle one keeps anyone from overwriting low words
    // that another thread has not read yet (between full exchanges the "a thread overwrites only what it read" rule
    // makes that unnecessary).
    template <int p>
    __device__ __forceinline__ void split_exchange(uint64_t (&x)[C]) const {
        uint32_t* w = reinterpret_cast<uint32_t*>(slab);
        const uint32_t sb = sbase<p>(), nb = sbase<p + 1>();
        static_for<0, C>([&](auto Rr) { constexpr int r = Rr; w[join(sb, img((uint32_t)r << G::rlo(p)))] = (uint32_t)x[r]; });
        exchange_sync<p>();
        uint32_t lo[C];
        static_for<0, C>([&](auto Rr) { constexpr int r = Rr; lo[r] = w[join(nb, img((uint32_t)r << G::rlo(p + 1)))]; });
        exchange_sync<p>();
        static_for<0, C>([&](auto Rr) { constexpr int r = Rr; w[join(sb, img((uint32_t)r << G::rlo(p)))] = (uint32_t)(x[r] >> 32); });
        exchange_sync<p>();
        static_for<0, C>([&](auto Rr) { constexpr int r = Rr; x[r] = (uint64_t)lo[r] | ((uint64_t)w[join(nb, img((uint32_t)r << G::rlo(p + 1)))] << 32); });
    }

    // forward passes [P0, P1): pass P0 reads the image unless it is pass 0 (x already holds the
    // pass-0 layout) and first orders the exchange that precedes it; every pass but the last
    // writes the image
    template <int P0, int P1>
    __device__ __forceinline__ void forward_passes(uint64_t (&x)[C], const twpair* tbl) const {
        rb2_no_hooks none;
        forward_passes<P0, P1>(x, tbl, none);
    }
    // ---- streamed twiddles (STREAM_TW) -------------------------------------------------------------------------------
    // A pass of ns stages reads, at its stage S, the 2^kk table entries j = 2^kk + o (kk = R - ns + S, o = b >> rb for butterfly b):
    // they are taken in chunks of up to CH entries, in stage order; chunk q+1 is requested before chunk q's butterflies run.
    static constexpr int CH = (OPT & kOptStreamCh1) != 0 ? 1 : R >= 5 ? 2 : 4;      // R = 5: two entries (8 VGPRs per buffer) keep the pass inside 128 VGPRs
    // forward: stage S of a pass has kk = R - ns + S (1, 2, 4 ... entries); inverse (INV): stages run the other way, kk = R - 1 - S
    static constexpr int st_kk(int ns, int S, bool inv = false) { return inv ? R - 1 - S : R - ns + S; }
    static constexpr int st_chunks_in_stage(int ns, int S, bool inv = false) { return (1 << st_kk(ns, S, inv)) > CH ? (1 << st_kk(ns, S, inv)) / CH : 1; }
    static constexpr int st_total(int ns, bool inv = false) { int t = 0; for (int S = 0; S < ns; ++S) t += st_chunks_in_stage(ns, S, inv); return t; }
    static constexpr int st_stage(int ns, int q, bool inv = false) { int S = 0; while (q >= st_chunks_in_stage(ns, S, inv)) { q -= st_chunks_in_stage(ns, S, inv); ++S; } return S; }
    static constexpr int st_chunk(int ns, int q, bool inv = false) { int S = 0; while (q >= st_chunks_in_stage(ns, S, inv)) { q -= st_chunks_in_stage(ns, S, inv); ++S; } return q; }
    static constexpr int st_count(int ns, int S, bool inv = false) { return (1 << st_kk(ns, S, inv)) < CH ? (1 << st_kk(ns, S, inv)) : CH; }
    struct tw_chunk {
        twpair e[CH];
    };
    template <int p, int q, bool INV = false>
    __device__ __forceinline__ void stream_load(tw_chunk& c, const tw_src<p>& t) const {
        constexpr int ns = G::hi(p) - G::rlo(p) + 1, S = st_stage(ns, q, INV), cc = st_chunk(ns, q, INV), kk = st_kk(ns, S, INV);
        static_for<0, st_count(ns, S, INV)>([&](auto I) {
            constexpr int j = (1 << kk) + cc * CH + (int)I;
            if constexpr (G::uniform_pass(p)) c.e[I] = load_uniform(t.col + j);
            else c.e[I] = lane_entry<p>(t.col, j);
        });
    }
    template <int P0, int P1, class Hooks>
    __device__ __forceinline__ void forward_passes_streamed(uint64_t (&x)[C], const twpair* tbl, Hooks& hooks) const {
        static_assert(P0 == 0 && P1 == NP, "whole transform");
        static_for<P0, P1>([&](auto P) {
            constexpr int p = P;
            constexpr int rlo = G::rlo(p), hi = G::hi(p), ns = hi - rlo + 1, NQ = st_total(ns);
            tw_src<p> t;
            fetch<p>(t, tbl);
            tw_chunk buf[2];
            stream_load<p, 0>(buf[0], t);
            if constexpr (p > 0 && !SPLIT) image_read<p>(x);
            static_for<0, NQ>([&](auto Qc) {
                constexpr int q = Qc;
                constexpr int S = st_stage(ns, q), cc = st_chunk(ns, q), cnt = st_count(ns, S);
                constexpr int rb = (hi - rlo) - S;
                constexpr bool last_stage = (rlo + rb) == 0;
                if constexpr (q + 1 < NQ) stream_load<p, q + 1>(buf[(q + 1) & 1], t);
                __builtin_amdgcn_sched_barrier(0);
                auto chunk_body = [&](auto M) {
                    static_for<(cc * CH) << rb, (cc * CH + cnt) << rb>([&](auto B) {
                        constexpr int b = B;
                        constexpr int r0 = ((b >> rb) << (rb + 1)) | (b & ((1 << rb) - 1));
                        constexpr int r1 = r0 | (1 << rb);
                        constexpr int stage = S0 + L - 1 - (rlo + rb);
                        if constexpr ((OPT & kOptPinBf) != 0) asm volatile("" : "+v"(x[r0]), "+v"(x[r1]));
                        butterfly<stage>(x[r0], x[r1], buf[q & 1].e[(b >> rb) - cc * CH]);
                        if constexpr (last_stage) {
                            x[r0] = final_reduce<decltype(M)::value>(x[r0]);
                            x[r1] = final_reduce<decltype(M)::value>(x[r1]);
                        }
                        if constexpr ((OPT & kOptPinBf) != 0) asm volatile("" : "+v"(x[r0]), "+v"(x[r1]));
                    });
                };
                if constexpr (last_stage) with_final_mode(chunk_body);
                else chunk_body(std::integral_constant<int, 0>{});
                __builtin_amdgcn_sched_barrier(0);
            });
            if constexpr (p < NP - 1) {
                hooks.template before_image_write<p>();
                if constexpr (SPLIT) split_exchange<p>(x);
                else {
                    image_write<p>(x);
                    exchange_sync<p>();
                }
                if constexpr (PRIO_BARRIER && !G::exchange_is_wave_local(p)) __builtin_amdgcn_s_setprio(0);
                hooks.template after_exchange_sync<p>();
            }
        });
    }

    template <int P0, int P1, class Hooks>
    __device__ __forceinline__ void forward_passes(uint64_t (&x)[C], const twpair* tbl, Hooks& hooks) const {
        if constexpr (STREAM_TW) {
            forward_passes_streamed<P0, P1>(x, tbl, hooks);
            return;
        }
        // look-ahead twiddles (TWA): entries 1..C/2-1 of the next per-lane pass are requested during the
        // last stage of the current pass, entries C/2..C-1 at the start of their own pass, so the L2
        // latency of the per-lane table reads overlaps butterflies instead of stalling the wave
        twpair ahead[C / 2];      // entries 1 .. C/2-1 (all stages but the pass's last)
        static_for<P0, P1>([&](auto P) {
            constexpr int p = P;
            constexpr int rlo = G::rlo(p), hi = G::hi(p), ns = hi - rlo + 1;
            constexpr bool twa_here = TWA && lane_full_pass(p);
            constexpr bool twa_prev = TWA && p > P0 && lane_full_pass(p);   // the previous pass fetched `ahead` for us
            constexpr bool twa_next = TWA && p + 1 < P1 && lane_full_pass(p + 1);
            tw_src<p> t;
            fetch<p>(t, tbl);
            twpair late[C / 2];   // entries C/2 .. C-1 (the pass's last stage)
            if constexpr (twa_here) {
                if constexpr (!twa_prev) {
                    static_for<1, C / 2>([&](auto J) { constexpr int j = J; ahead[j] = lane_entry<p>(tbl, j); });
                }
                static_for<0, C / 2>([&](auto J) { constexpr int j = J; late[j] = lane_entry<p>(tbl, j + C / 2); });
            }
            hooks.template after_twiddle_issue<p>();
            if constexpr (p > 0 && !SPLIT) {
                if constexpr (p == P0) exchange_sync<p - 1>();
                image_read<p>(x);
                if constexpr (2 * p + 1 < 12) stamp<2 * p + 1>(x[C - 1]);
            }
            static_for<0, ns>([&](auto S) {
                constexpr int rb = (hi - rlo) - S;        // gap bits descend: Cooley-Tukey
                constexpr int kk = R - 1 - rb;
                constexpr bool last_stage = (rlo + rb) == 0;
                if constexpr (twa_next && S == (ns > 1 ? ns - 1 : 0)) {
                    // `ahead` is free once this pass's first two stages are done
                    constexpr int pn = p + 1;
                    static_for<1, C / 2>([&](auto J) { constexpr int j = J; ahead[j] = lane_entry<pn>(tbl, j); });
                }
                // the transform's last stage: one wave-uniform branch around the whole stage picks the final-reduction mode
                auto stage_body = [&](auto M) {
                    static_for<0, C / 2>([&](auto B) {
                        // B-th butterfly of the stage: insert a 0 at register bit rb
                        constexpr int b = B;
                        constexpr int r0 = ((b >> rb) << (rb + 1)) | (b & ((1 << rb) - 1));
                        constexpr int r1 = r0 | (1 << rb);
                        constexpr int j = (1 << kk) + (r0 >> (rb + 1));
                        twpair w;
                        if constexpr (twa_here && j < C / 2) w = ahead[j];
                        else if constexpr (twa_here) w = late[j - C / 2];
                        else w = twiddle<p>(t, j);
                        constexpr int stage = S0 + L - 1 - (rlo + rb);   // position in the whole transform
                        butterfly<stage>(x[r0], x[r1], w);
                        if constexpr (last_stage) {
                            x[r0] = final_reduce<decltype(M)::value>(x[r0]);
                            x[r1] = final_reduce<decltype(M)::value>(x[r1]);
                        }
                        hooks.template after_butterfly<p, S == ns - 1, b>();
                    });
                };
                if constexpr (last_stage) with_final_mode(stage_body);
                else stage_body(std::integral_constant<int, 0>{});
            });
            if constexpr (2 * p + 2 < 12) stamp<2 * p + 2>(x[C - 1]);
            if constexpr (p < NP - 1) {
                // A thread overwrites exactly the image words it read for this pass, so no other
                // thread can still need them: only the read side of an exchange has to be ordered.
                hooks.template before_image_write<p>();
                if constexpr (SPLIT) {
                    static_assert(!SPLIT || (P0 == 0 && P1 == NP), "split-word exchanges run the whole transform in one call");
                    split_exchange<p>(x);
                    if constexpr (PRIO_BARRIER && !G::exchange_is_wave_local(p)) __builtin_amdgcn_s_setprio(0);
                } else {
                image_write<p>(x);
                if constexpr (p < P1 - 1) {
                    exchange_sync<p>();
                    if constexpr (PRIO_BARRIER && !G::exchange_is_wave_local(p)) __builtin_amdgcn_s_setprio(0);
                    hooks.template after_exchange_sync<p>();
                }
                }
            }
        });
    }
    // x in pass-0 layout (element tid + T*r, any values in [0,2m)) -> forward transform, x in the
    // last pass's layout (elements tid*C .. tid*C+C-1), fully reduced
    __device__ __forceinline__ void forward(uint64_t (&x)[C], const twpair* tbl) const {
        forward_passes<0, NP>(x, tbl);
    }
    template <class Hooks>
    __device__ __forceinline__ void forward(uint64_t (&x)[C], const twpair* tbl, Hooks& hooks) const {
        forward_passes<0, NP>(x, tbl, hooks);
    }

    // INV_TWF: twiddles of the inverse's first pass (the last pass's table, per lane) held in registers from before the
    // frame has arrived; entry j serves stage rb = R-1-floor(log2 j) of that pass
    struct inv_pre {
        twpair tw[C];
    };
    static constexpr int ilog2(int v) { return v <= 1 ? 0 : 1 + ilog2(v >> 1); }
    static constexpr bool inv_pre_has(int j) {
        constexpr int p = NP - 1;
        const int rb = R - 1 - ilog2(j);
        return INV_TWF && rb <= G::hi(p) - G::rlo(p) && (INV_TWF_ALL || rb == 0);
    }
    __device__ __forceinline__ void inverse_prefetch(inv_pre& pre, const twpair* itbl) const {
        if constexpr (INV_TWF) {
            constexpr int p = NP - 1;
            tw_src<p> t;
            fetch<p>(t, itbl);
            static_for<1, C>([&](auto J) {
                constexpr int j = J;
                if constexpr (inv_pre_has(j)) pre.tw[j] = lane_entry<p>(itbl, j);
            });
        }
    }
    __device__ __forceinline__ void inverse(uint64_t (&x)[C], const twpair* itbl, const prime_consts& pc) const {
        inv_pre pre;
        inverse_prefetch(pre, itbl);
        inverse(x, itbl, pc, pre);
    }

    // SPLIT: the exchange between inverse passes p and p-1 through the 32-bit image (the mirror of split_exchange)
    template <int p>
    __device__ __forceinline__ void split_exchange_inv(uint64_t (&x)[C]) const {
        uint32_t* w = reinterpret_cast<uint32_t*>(slab);
        const uint32_t sb = sbase<p>(), nb = sbase<p - 1>();
        static_for<0, C>([&](auto Rr) { constexpr int r = Rr; w[join(sb, img((uint32_t)r << G::rlo(p)))] = (uint32_t)x[r]; });
        exchange_sync<p - 1>();
        uint32_t lo[C];
        static_for<0, C>([&](auto Rr) { constexpr int r = Rr; lo[r] = w[join(nb, img((uint32_t)r << G::rlo(p - 1)))]; });
        exchange_sync<p - 1>();
        static_for<0, C>([&](auto Rr) { constexpr int r = Rr; w[join(sb, img((uint32_t)r << G::rlo(p)))] = (uint32_t)(x[r] >> 32); });
        exchange_sync<p - 1>();
        static_for<0, C>([&](auto Rr) { constexpr int r = Rr; x[r] = (uint64_t)lo[r] | ((uint64_t)w[join(nb, img((uint32_t)r << G::rlo(p - 1)))] << 32); });
    }

    // STREAM_TW form of the inverse (whole frames only: split_log = 0, so the top stage folds n^-1 in): twiddles in chunks, one chunk
    // ahead, butterflies pinned in program order (kOptPinBf) -- see forward_passes_streamed
    __device__ __forceinline__ void inverse_streamed(uint64_t (&x)[C], const twpair* itbl, const prime_consts& pc) const {
        static_for<0, NP>([&](auto Qp) {
            constexpr int p = NP - 1 - Qp;
            constexpr int rlo = G::rlo(p), hi = G::hi(p), ns = hi - rlo + 1, NQ = st_total(ns, true);
            constexpr int B0 = (p == NP - 1) ? 4 : 8;
            tw_src<p> t;
            fetch<p>(t, itbl);
            tw_chunk buf[2];
            stream_load<p, 0, true>(buf[0], t);
            if constexpr (p < NP - 1 && !SPLIT) image_read<p>(x);
            static_for<0, NQ>([&](auto Qc) {
                constexpr int q = Qc;
                constexpr int S = st_stage(ns, q, true), cc = st_chunk(ns, q, true), cnt = st_count(ns, S, true);
                constexpr int rb = S;                         // gap bits ascend
                constexpr bool top_stage = (rlo + rb) == L - 1;
                if constexpr (q + 1 < NQ) stream_load<p, q + 1, true>(buf[(q + 1) & 1], t);
                __builtin_amdgcn_sched_barrier(0);
                static_for<(cc * CH) << rb, (cc * CH + cnt) << rb>([&](auto Bf) {
                    constexpr int b = Bf;
                    constexpr int r0 = ((b >> rb) << (rb + 1)) | (b & ((1 << rb) - 1));
                    constexpr int r1 = r0 | (1 << rb);
                    constexpr int BND = gs_bound(B0, rb, r0);
                    if constexpr ((OPT & kOptPinBf) != 0) asm volatile("" : "+v"(x[r0]), "+v"(x[r1]));
                    if constexpr (top_stage) {
                        if constexpr (LAZY_INV) gs_last_lazy16<BND, SEL>(x[r0], x[r1], pc.n_inv, pc.n_inv_p, pc.w1n, pc.w1n_p, k, fc);
                        else gs_last_form<FAST>(x[r0], x[r1], pc.n_inv, pc.n_inv_p, pc.w1n, pc.w1n_p, k);
                        x[r0] = reduce_final_inv<FAST, SEL>(x[r0], k, fc);
                        x[r1] = reduce_final_inv<FAST, SEL>(x[r1], k, fc);
                    } else {
                        const twpair w = buf[q & 1].e[(b >> rb) - cc * CH];
                        if constexpr (LAZY_INV) gs_butterfly_lazy16<BND, SEL>(x[r0], x[r1], w.x, w.y, k, fc);
                        else gs_butterfly_form<FAST, SEL>(x[r0], x[r1], w.x, w.y, k);
                    }
                    if constexpr ((OPT & kOptPinBf) != 0) asm volatile("" : "+v"(x[r0]), "+v"(x[r1]));
                });
                __builtin_amdgcn_sched_barrier(0);
            });
            if constexpr (LAZY_INV && p > 0) {
                static_for<0, C>([&](auto Rr) {
                    constexpr int r = Rr;
                    if constexpr (gs_bound(B0, ns, r) == 16) x[r] = csub_8q<SEL>(x[r], fc);
                });
            }
            if constexpr (p > 0) {
                if constexpr (SPLIT) split_exchange_inv<p>(x);
                else {
                    image_write<p>(x);
                    exchange_sync<p - 1>();
                }
            }
        });
    }

    // x in the last pass's layout, values in [0,m) -> inverse transform (Gentleman-Sande, gap bits
    // ascending), x in pass-0 layout, fully reduced.  With split_log = 0 the top stage also
    // multiplies by n^-1; otherwise inv_global_stage finishes the transform.
    __device__ __forceinline__ void inverse(uint64_t (&x)[C], const twpair* itbl, const prime_consts& pc, const inv_pre& pre) const {
        if constexpr (STREAM_TW) {
            inverse_streamed(x, itbl, pc);
            return;
        }
        twpair first[4];     // TWA_INV: entries 4..7 of the pass about to start (its first stage), fetched one pass early
        static_for<0, NP>([&](auto Q) {
            constexpr int p = NP - 1 - Q;
            constexpr int rlo = G::rlo(p), hi = G::hi(p);
            constexpr bool twa_have = TWA_INV && p < NP - 1 && lane_full_pass(p) && lane_full_pass(p + 1);   // the previous pass fetched `first`
            constexpr bool twa_next = TWA_INV && p > 0 && lane_full_pass(p) && lane_full_pass(p - 1);
            if constexpr ((OPT & kOptInvPrioAsc) != 0) __builtin_amdgcn_s_setprio(Q >= 3 ? 3 : (int)Q);
            if constexpr ((OPT & kOptInvPrioDesc) != 0) __builtin_amdgcn_s_setprio(Q >= 3 ? 0 : 3 - (int)Q);
            tw_src<p> t;
            fetch<p>(t, itbl);
            if constexpr (p < NP - 1) {
                image_read<p>(x);
                if constexpr (2 + 2 * Q < 12) stamp<2 + 2 * Q>(x[C - 1]);     // trace: exchange done (Q = passes completed)
            }
            // 16q-lazy form: every register of the first pass starts below 4q, of the later ones below 8q
            constexpr int B0 = (p == NP - 1) ? 4 : 8;
            static_for<0, hi - rlo + 1>([&](auto S) {
                constexpr int rb = S;                     // gap bits ascend
                constexpr int kk = R - 1 - rb;
                constexpr bool top_stage = (rlo + rb) == L - 1;
                if constexpr (twa_next && S == hi - rlo) {
                    // `first` is free: this pass's first stage is long done
                    constexpr int pn = p - 1;
                    static_for<0, 4>([&](auto J) { constexpr int jj = J; first[jj] = lane_entry<pn>(itbl, jj + 4); });
                }
                static_for<0, C / 2>([&](auto Bf) {
                    constexpr int b = Bf;
                    constexpr int r0 = ((b >> rb) << (rb + 1)) | (b & ((1 << rb) - 1));
                    constexpr int r1 = r0 | (1 << rb);
                    constexpr int BND = gs_bound(B0, rb, r0);
                    if (top_stage && split_log == 0) {
                        if constexpr (LAZY_INV) gs_last_lazy16<BND, SEL>(x[r0], x[r1], pc.n_inv, pc.n_inv_p, pc.w1n, pc.w1n_p, k, fc);
                        else gs_last_form<FAST>(x[r0], x[r1], pc.n_inv, pc.n_inv_p, pc.w1n, pc.w1n_p, k);
                    } else {
                        constexpr int j = (1 << kk) + (r0 >> (rb + 1));
                        twpair w;
                        if constexpr (twa_have && j >= 4) w = first[j - 4];
                        else if constexpr (p == NP - 1 && inv_pre_has(j)) w = pre.tw[j];
                        else w = twiddle<p>(t, j);
                        if constexpr (LAZY_INV) {
                            gs_butterfly_lazy16<BND, SEL>(x[r0], x[r1], w.x, w.y, k, fc);
                            if constexpr (top_stage) {      // a split transform's resident part: back below 4q for the final reduction
                                x[r0] = csub_8q<SEL>(x[r0], fc);
                                x[r0] = SEL ? csub_select(x[r0], k) : csub_sign(x[r0], k);
                            }
                        } else {
                            gs_butterfly_form<FAST, SEL>(x[r0], x[r1], w.x, w.y, k);
                        }
                    }
                    if constexpr (top_stage) {
                        x[r0] = reduce_final_inv<FAST, SEL>(x[r0], k, fc);
                        x[r1] = reduce_final_inv<FAST, SEL>(x[r1], k, fc);
                    }
                });
            });
            if constexpr (3 + 2 * Q < 12) stamp<3 + 2 * Q>(x[C - 1]);       // trace: this pass's butterflies done
            if constexpr (LAZY_INV && p > 0) {
                // the next pass assumes 8q: bring the registers that ended at 16q back
                static_for<0, C>([&](auto Rr) {
                    constexpr int r = Rr;
                    if constexpr (gs_bound(B0, hi - rlo + 1, r) == 16) x[r] = csub_8q<SEL>(x[r], fc);
                });
            }
            if constexpr (p > 0) {
                if constexpr ((OPT & kOptInvPrioTail) != 0 && !G::exchange_is_wave_local(p - 1)) __builtin_amdgcn_s_setprio(3);
                image_write<p>(x);
                exchange_sync<p - 1>();
            }
        });
    }

    // last-pass layout <-> lane-contiguous global accesses, through the image (wave-local: after
    // the last forward pass / before the first inverse pass a wave owns 64*C contiguous elements)
    __device__ __forceinline__ void store_last_layout(const uint64_t (&x)[C], uint64_t* __restrict__ out, int64_t base, bool live) const {
        static_assert(G::last_pass_wave_contiguous(), "store path assumes a wave-contiguous last pass");
        if constexpr (SPLIT) {
            uint32_t* w = reinterpret_cast<uint32_t*>(slab);
            const uint32_t own32 = img(tid << R);
            const uint32_t e0 = ((tid >> 6) << (6 + R)) + (tid & 63u), s0 = img(e0);
            static_for<0, C>([&](auto Rr) { constexpr int r = Rr; w[join(own32, img((uint32_t)r))] = (uint32_t)x[r]; });
            wave_lds_sync();
            uint32_t lo[C];
            static_for<0, C>([&](auto Rr) { constexpr int r = Rr; lo[r] = w[join(s0, img(64u * (uint32_t)r))]; });
            wave_lds_sync();
            static_for<0, C>([&](auto Rr) { constexpr int r = Rr; w[join(own32, img((uint32_t)r))] = (uint32_t)(x[r] >> 32); });
            wave_lds_sync();
            if (live) {
                static_for<0, C>([&](auto Rr) {
                    constexpr int r = Rr;
                    const uint64_t v = (uint64_t)lo[r] | ((uint64_t)w[join(s0, img(64u * (uint32_t)r))] << 32);
                    if constexpr ((OPT & kOptNtStore) != 0) __builtin_nontemporal_store(v, &out[base + e0 + 64u * (uint32_t)r]);
                    else out[base + e0 + 64u * (uint32_t)r] = v;
                });
            }
            return;
        }
        const uint32_t own = img(tid << R);
        static_for<0, C>([&](auto Rr) { constexpr int r = Rr; slab[join(own, img((uint32_t)r))] = x[r]; });
        wave_lds_sync();
        if constexpr (TRACE) {
            const uint32_t e0 = ((tid >> 6) << (6 + R)) + (tid & 63u), s0 = img(e0);
            uint64_t y[C];
            static_for<0, C>([&](auto Rr) { constexpr int r = Rr; y[r] = slab[join(s0, img(64u * (uint32_t)r))]; });
            stamp<9>(y[C - 1]);
            if (live) static_for<0, C>([&](auto Rr) { constexpr int r = Rr; out[base + e0 + 64u * (uint32_t)r] = y[r]; });
            stamp<10>(y[0]);
            if (trace_wait_stores) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            stamp<11>(y[0]);
            trace_flush();
            return;
        }
        if constexpr (PRIO_STORE) __builtin_amdgcn_s_setprio(3);
        if (live) {
            const uint32_t e0 = ((tid >> 6) << (6 + R)) + (tid & 63u), s0 = img(e0);
            static_for<0, C>([&](auto Rr) {
                constexpr int r = Rr;
                if constexpr ((OPT & kOptNtStore) != 0) __builtin_nontemporal_store(slab[join(s0, img(64u * (uint32_t)r))], &out[base + e0 + 64u * (uint32_t)r]);
                else out[base + e0 + 64u * (uint32_t)r] = slab[join(s0, img(64u * (uint32_t)r))];
            });
        }
    }
    // `in2` (may be null): the coefficient-wise product in * in2 mod q is taken while loading, so a
    // polynomial product needs no separate pointwise pass before its inverse transform
    __device__ __forceinline__ void load_last_layout(uint64_t (&x)[C], const uint64_t* __restrict__ in, const uint64_t* __restrict__ in2,
                                                     const barrett128& bk, int64_t base) const {
        load_last_issue(x, in, base);
        load_last_stage(x, in2, bk, base);
    }
    // first half: the frame's lane-contiguous global loads (no LDS traffic yet, so a loop kernel can put its
    // image hand-over barrier between the two halves, behind the load latency)
    __device__ __forceinline__ void load_last_issue(uint64_t (&x)[C], const uint64_t* __restrict__ in, int64_t base) const {
        const uint32_t e0 = ((tid >> 6) << (6 + R)) + (tid & 63u);
#pragma unroll
        for (int r = 0; r < C; ++r)
            x[r] = (OPT & kOptNtLoad) ? __builtin_nontemporal_load(&in[base + e0 + 64u * (uint32_t)r]) : in[base + e0 + 64u * (uint32_t)r];
    }
    // second half: optional coefficient-wise product with in2, staging through the wave's own part of the image
    __device__ __forceinline__ void load_last_stage(uint64_t (&x)[C], const uint64_t* __restrict__ in2, const barrett128& bk, int64_t base) const {
        const uint32_t e0 = ((tid >> 6) << (6 + R)) + (tid & 63u), s0 = img(e0);
        if constexpr (SPLIT) {
            // 32-bit image: low words through the wave's part of the image, then the high words
            uint32_t* w = reinterpret_cast<uint32_t*>(slab);
            const uint32_t own32 = img(tid << R);
#pragma unroll
            for (int r = 0; r < C; ++r) {
                uint64_t v = x[r];
                if (in2) {   // wave-uniform
                    const uint64_t u = (OPT & kOptNtLoad) ? __builtin_nontemporal_load(&in2[base + e0 + 64u * (uint32_t)r]) : in2[base + e0 + 64u * (uint32_t)r];
                    v = mul_mod_barrett(reduce_4q(v, k.q, k.q << 1), reduce_4q(u, k.q, k.q << 1), bk);
                }
                if constexpr (!FAST) v = csub(v, k.m);
                x[r] = v;
            }
            static_for<0, C>([&](auto Rr) { constexpr int r = Rr; w[join(s0, img(64u * (uint32_t)r))] = (uint32_t)x[r]; });
            if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
            wave_lds_sync();
            uint32_t lo[C];
            static_for<0, C>([&](auto Rr) { constexpr int r = Rr; lo[r] = w[join(own32, img((uint32_t)r))]; });
            wave_lds_sync();
            static_for<0, C>([&](auto Rr) { constexpr int r = Rr; w[join(s0, img(64u * (uint32_t)r))] = (uint32_t)(x[r] >> 32); });
            wave_lds_sync();
            static_for<0, C>([&](auto Rr) { constexpr int r = Rr; x[r] = (uint64_t)lo[r] | ((uint64_t)w[join(own32, img((uint32_t)r))] << 32); });
            return;
        }
#pragma unroll
        for (int r = 0; r < C; ++r) {
            uint64_t v = x[r];
            if (in2) {   // wave-uniform
                const uint64_t u = (OPT & kOptNtLoad) ? __builtin_nontemporal_load(&in2[base + e0 + 64u * (uint32_t)r]) : in2[base + e0 + 64u * (uint32_t)r];
                v = mul_mod_barrett(reduce_4q(v, k.q, k.q << 1), reduce_4q(u, k.q, k.q << 1), bk);
            }
            if constexpr (!FAST) v = csub(v, k.m);    // exact form wants [0,2q); inputs may be < 4q
            slab[join(s0, img(64u * (uint32_t)r))] = v;
        }
        if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
        wave_lds_sync();
        const uint32_t own = img(tid << R);
#pragma unroll
        for (int r = 0; r < C; ++r) x[r] = slab[join(own, img((uint32_t)r))];
    }
};

#define AGX_RB2_PROLOGUE                                                                          \
    using F = rb2_frame<L, R, (ARITH & 1) == 1, (ARITH >> 1)>;                                    \
    constexpr int C = F::C, T = F::T;                                                             \
    static_assert(T >= 64, "one frame must span whole waves");                                    \
    F f;                                                                                          \
    f.tid = threadIdx.x & (T - 1);                                                                \
    const uint32_t slot = threadIdx.x / T;                                                        \
    uint64_t fx = (uint64_t)blockIdx.x * PPB + slot;                                              \
    const bool live = fx < frames_x;                                                              \
    if (!live) fx = frames_x - 1;                                                                 \
    const uint32_t prime = blockIdx.y;                                                            \
    const uint64_t poly = fx >> split_log;                                                        \
    f.blk = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(fx & ((1u << split_log) - 1u)));   /* wave-uniform (T >= 64) */ \
    f.split_log = split_log;                                                                      \
    f.init_consts(consts[prime].q, consts[prime].est);                                                               \
    f.slab = reinterpret_cast<uint64_t*>(agx_dyn_lds) + (size_t)slot * F::slab_elems;             \
    const int64_t base = (int64_t)prime * prime_stride + (int64_t)poly * poly_stride + ((int64_t)f.blk << L)

template <int L, int R, int PPB, int ARITH, int MINW>
__global__ void __launch_bounds__((1 << (L - R)) * PPB, MINW)
fwd_rb2(const uint64_t* __restrict__ in, uint64_t* __restrict__ out,
        const prime_consts* __restrict__ consts, const twpair* __restrict__ tw_rb,
        uint32_t pairs_per_prime, uint32_t split_log, uint64_t frames_x,
        int64_t prime_stride, int64_t poly_stride, uint32_t lazy_out) {
    uint64_t t_entry = 0;
    if constexpr (((ARITH >> 1) & kOptTrace) != 0) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_entry) : : "memory");
    if constexpr (((ARITH >> 1) & kOptPrio) != 0) __builtin_amdgcn_s_setprio(3);
    AGX_RB2_PROLOGUE;
    f.lazy_out = lazy_out != 0;
    uint64_t x[C];
    if constexpr (((ARITH >> 1) & kOptAblateHbm) != 0) {
        const int64_t hot = (int64_t)prime * prime_stride;     // frame 0 of the prime, in and out
        const uint64_t* src = in + ((((ARITH >> 1) & kOptAblateStOnly) != 0) ? base : hot);
#pragma unroll
        for (int r = 0; r < C; ++r) x[r] = (src + (uint32_t)r * T)[f.tid];
        f.forward(x, tw_rb + (size_t)prime * pairs_per_prime);
        f.store_last_layout(x, out, (((ARITH >> 1) & kOptAblateLdOnly) != 0) ? base : hot, live);
        return;
    }
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

template <class F, int PF>
struct rb2_stream_hooks {
    static constexpr int C = F::C, T = F::T;
    const uint64_t* in;
    uint64_t (&xn)[C];
    const F& f;
    volatile uint32_t* mailbox;
    uint32_t* ticket;
    uint32_t slot, total, batch;
    int64_t prime_stride, poly_stride;
    uint32_t next;
    uint32_t pending;     // thread 0: the ticket drawn one frame ahead (TK == 1)
    template <int p> __device__ __forceinline__ void before_image_write() const {
        if constexpr (p == 0) __builtin_amdgcn_s_barrier();     // barrier A: the image is free again
    }
    template <int p> __device__ __forceinline__ void after_exchange_sync() {
        if constexpr (p == 0) next = (uint32_t)__builtin_amdgcn_readfirstlane((int)mailbox[slot]);
    }

    template <int p> __device__ __forceinline__ void after_twiddle_issue() {
        if constexpr (PF == 1 && p == F::NP - 1) {
            if (f.tid == 0) pending = atomicAdd(ticket, 1u);    // for the frame after next; back long before the loop top
        }
        if constexpr (PF == 0 && p == F::NP - 1) {
            if (next < total) {     // wave-uniform
                const int64_t nb = (int64_t)(next / batch) * prime_stride + (int64_t)(next % batch) * poly_stride;
#pragma unroll
                for (int r = 0; r < C; ++r) xn[r] = F::NT_LOAD ? __builtin_nontemporal_load(&in[nb + f.tid + (uint32_t)r * T]) : in[nb + f.tid + (uint32_t)r * T];
            }
        }
    }
    // PF 1: two registers of the next frame after each butterfly of the last pass's last stage, into the
    // VGPRs its twiddle has just left (a frame past the end re-reads the last frame: no branch, so the
    // fences below fix where the loads are issued)
    template <int p, bool last_stage_of_pass, int b> __device__ __forceinline__ void after_butterfly() const {
        if constexpr (PF == 1 && p == F::NP - 1 && last_stage_of_pass) {
            const uint32_t fn = next < total ? next : total - 1;
            const uint64_t* src = in + (int64_t)(fn / batch) * prime_stride + (int64_t)(fn % batch) * poly_stride;
            asm volatile("" ::: "memory");
            xn[2 * b] = F::NT_LOAD ? __builtin_nontemporal_load(src + (uint32_t)(2 * b) * T + f.tid) : (src + (uint32_t)(2 * b) * T)[f.tid];
            xn[2 * b + 1] = F::NT_LOAD ? __builtin_nontemporal_load(src + (uint32_t)(2 * b + 1) * T + f.tid) : (src + (uint32_t)(2 * b + 1) * T)[f.tid];
            asm volatile("" ::: "memory");
        }
    }
};

// Streaming forward kernel: a resident grid of workgroups (as many as fit the chip at once) draws frame
// numbers from a ticket counter, so a workgroup slot is never empty while the dispatcher refills it
// (tools/timeline.py: 5.7 of 8 wave slots occupied in fwd_rb2), and the next frame's coefficients are
// requested while the last pass of the current frame still computes.  Per frame: barrier A (bare
// s_barrier) before the first image write -- every wave has then finished reading the previous frame's
// staged results -- and the usual barrier after it.  `ticket[0]` hands out frames (grid size + k),
// `ticket[1]` counts retired workgroups; the last one to leave zeroes both for the next launch, so one
// ticket pair must not be shared by launches that can run at the same time.
template <int L, int R, int ARITH, int MINW, int PF>
__global__ void __launch_bounds__((1 << (L - R)), MINW)
fwd_rb2_stream(const uint64_t* __restrict__ in, uint64_t* __restrict__ out,
               const prime_consts* __restrict__ consts, const twpair* __restrict__ tw_rb,
               uint32_t pairs_per_prime, uint32_t batch, uint32_t total,
               int64_t prime_stride, int64_t poly_stride, uint32_t lazy_out, uint32_t* __restrict__ ticket) {
    using F = rb2_frame<L, R, (ARITH & 1) == 1, (ARITH >> 1)>;
    constexpr int C = F::C, T = F::T;
    F f;
    f.tid = threadIdx.x;
    f.blk = 0;
    f.split_log = 0;
    f.lazy_out = lazy_out != 0;
    f.slab = reinterpret_cast<uint64_t*>(agx_dyn_lds);
    volatile uint32_t* mailbox = reinterpret_cast<volatile uint32_t*>(reinterpret_cast<unsigned char*>(f.slab) + F::image_bytes);   // two words behind the image


    uint32_t fr = blockIdx.x;
    if (fr >= total) return;
    uint64_t xn[C];
    {
        const int64_t b0 = (int64_t)(fr / batch) * prime_stride + (int64_t)(fr % batch) * poly_stride;
#pragma unroll
        for (int r = 0; r < C; ++r) xn[r] = F::NT_LOAD ? __builtin_nontemporal_load(&in[b0 + f.tid + (uint32_t)r * T]) : in[b0 + f.tid + (uint32_t)r * T];
    }
    rb2_stream_hooks<F, PF> hooks{in, xn, f, mailbox, ticket, 0, total, batch, prime_stride, poly_stride, 0, 0};
    if constexpr (PF == 1) {
        if (threadIdx.x == 0) hooks.pending = atomicAdd(ticket, 1u);
    }
    for (uint32_t it = 0;; ++it) {
        const uint32_t prime = fr / batch, poly = fr % batch;
        const int64_t base = (int64_t)prime * prime_stride + (int64_t)poly * poly_stride;
        f.init_consts(consts[prime].q, consts[prime].est);
        hooks.slot = it & 1u;
        if constexpr (PF == 1) {
            if (threadIdx.x == 0) mailbox[it & 1u] = hooks.pending + gridDim.x;      // drawn during the previous frame's last pass
        } else {
            if (threadIdx.x == 0) mailbox[it & 1u] = atomicAdd(ticket, 1u) + gridDim.x;   // read by everyone after this frame's barrier
        }
        uint64_t x[C];
        if constexpr (F::TRACE) {
            uint64_t t_top;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_top) : : "memory");
            f.ts[0] = t_top;
            f.trace_wave = (uint64_t)fr * (T >> 6) + (threadIdx.x >> 6);
            f.trace_wait_stores = false;
        }
#pragma unroll
        for (int r = 0; r < C; ++r) x[r] = xn[r];
        if constexpr (F::TRACE) f.template stamp<1>(x[C - 1]);
        f.forward(x, tw_rb + (size_t)prime * pairs_per_prime, hooks);
        f.store_last_layout(x, out, base, true);
        fr = hooks.next;
        if (fr >= total) break;
    }
    if (threadIdx.x == 0) {
        if (atomicAdd(ticket + 1, 1u) == gridDim.x - 1) {     // last workgroup out: reset for the next launch
            __threadfence();
            ticket[0] = 0;
            ticket[1] = 0;
        }
    }
}

// Loop kernels: a resident grid (as many workgroups as the chip holds at once) walks over the frames with a fixed
// stride.  A workgroup that transforms frame after frame issues the next frame's loads right behind the current
// frame's stores, so the store drain of one frame and the load latency of the next overlap -- with one workgroup
// per CU (n = 16384: the frame's image fills the LDS) nothing else on the CU could cover either.  The frame-to-frame
// hand-over of the LDS image needs one bare s_barrier (no memory wait in front of it: the stores keep draining).
struct rb2_loop_hooks {
    bool first;
    // every wave has finished reading the previous frame's staged results before this frame's first exchange scatters
    template <int p> __device__ __forceinline__ void before_image_write() const {
        if constexpr (p == 0) {
            if (!first) __builtin_amdgcn_s_barrier();
        }
    }
    template <int p> __device__ __forceinline__ void after_twiddle_issue() const {}
    template <int p> __device__ __forceinline__ void after_exchange_sync() const {}
    template <int p, bool last_stage_of_pass, int b> __device__ __forceinline__ void after_butterfly() const {}
};

template <int L, int R, int ARITH, int MINW>
__global__ void __launch_bounds__((1 << (L - R)), MINW)
fwd_rb2_loop(const uint64_t* __restrict__ in, uint64_t* __restrict__ out,
             const prime_consts* __restrict__ consts, const twpair* __restrict__ tw_rb,
             uint32_t pairs_per_prime, uint32_t batch, uint32_t total,
             int64_t prime_stride, int64_t poly_stride, uint32_t lazy_out) {
    using F = rb2_frame<L, R, (ARITH & 1) == 1, (ARITH >> 1)>;
    constexpr int C = F::C, T = F::T;
    F f;
    f.tid = threadIdx.x;
    f.blk = 0;
    f.split_log = 0;
    f.lazy_out = lazy_out != 0;
    f.slab = reinterpret_cast<uint64_t*>(agx_dyn_lds);
    rb2_loop_hooks hooks{true};
    for (uint32_t fr = blockIdx.x; fr < total; fr += gridDim.x) {     // wave-uniform trip count
        const uint32_t prime = fr / batch, poly = fr % batch;
        const int64_t base = (int64_t)prime * prime_stride + (int64_t)poly * poly_stride;
        f.init_consts(consts[prime].q, consts[prime].est);
        uint64_t x[C];
        const uint64_t* src = in + base;
#pragma unroll
        for (int r = 0; r < C; ++r) x[r] = F::NT_LOAD ? __builtin_nontemporal_load(src + (uint32_t)r * T + f.tid) : (src + (uint32_t)r * T)[f.tid];
        f.forward(x, tw_rb + (size_t)prime * pairs_per_prime, hooks);
        f.store_last_layout(x, out, base, true);
        hooks.first = false;
    }
}

// Dynamic form of the loop kernels: the resident workgroups draw their next frame from a ticket counter instead of
// walking with a fixed stride, so a CU that runs a few per cent faster simply takes more frames (the fixed stride cost
// the forward loop kernel 4 %).  Thread 0 draws the ticket one frame ahead, right behind the frame loads -- the atomic's
// latency hides behind them -- and hands it round the workgroup through a two-slot LDS mailbox that everybody reads
// after the frame's own cross-wave barrier.  ticket[0] = frames handed out beyond the first round, ticket[1] = retired
// workgroups; the last workgroup out zeroes both (the plan gives every launch its own pair from a ring).
struct rb2_dloop_hooks {
    bool first;
    volatile uint32_t* mailbox;
    uint32_t slot;
    uint32_t next;
    template <int p> __device__ __forceinline__ void before_image_write() const {
        if constexpr (p == 0) {
            if (!first) __builtin_amdgcn_s_barrier();
        }
    }
    template <int p> __device__ __forceinline__ void after_twiddle_issue() const {}
    template <int p> __device__ __forceinline__ void after_exchange_sync() {
        if constexpr (p == 0) next = (uint32_t)__builtin_amdgcn_readfirstlane((int)mailbox[slot]);
    }
    template <int p, bool last_stage_of_pass, int b> __device__ __forceinline__ void after_butterfly() const {}
};

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
fwd_rb2_dloop(const uint64_t* __restrict__ in, uint64_t* __restrict__ out,
              const prime_consts* __restrict__ consts, const twpair* __restrict__ tw_rb,
              uint32_t pairs_per_prime, uint32_t batch, uint32_t total,
              int64_t prime_stride, int64_t poly_stride, uint32_t lazy_out, uint32_t* __restrict__ ticket) {
    using F = rb2_frame<L, R, (ARITH & 1) == 1, (ARITH >> 1)>;
    static_assert(!F::G::exchange_is_wave_local(0), "the mailbox is read behind exchange 0's workgroup barrier");
    constexpr int C = F::C, T = F::T;
    F f;
    f.tid = threadIdx.x;
    f.blk = 0;
    f.split_log = 0;
    f.lazy_out = lazy_out != 0;
    f.slab = reinterpret_cast<uint64_t*>(agx_dyn_lds);
    volatile uint32_t* mailbox = reinterpret_cast<volatile uint32_t*>(reinterpret_cast<unsigned char*>(f.slab) + F::image_bytes);   // two words behind the image
    rb2_dloop_hooks hooks{true, mailbox, 0, 0};
    uint32_t fr = blockIdx.x;       // launch guarantees gridDim.x <= total
    for (uint32_t it = 0;; ++it) {
        const uint32_t prime = fr / batch, poly = fr % batch;
        const int64_t base = (int64_t)prime * prime_stride + (int64_t)poly * poly_stride;
        f.init_consts(consts[prime].q, consts[prime].est);
        uint64_t x[C];
        const uint64_t* src = in + base;
#pragma unroll
        for (int r = 0; r < C; ++r) x[r] = F::NT_LOAD ? __builtin_nontemporal_load(src + (uint32_t)r * T + f.tid) : (src + (uint32_t)r * T)[f.tid];
        hooks.slot = it & 1u;
        if (threadIdx.x == 0) mailbox[it & 1u] = atomicAdd(ticket, 1u) + gridDim.x;   // read by everyone behind this frame's barrier
        f.forward(x, tw_rb + (size_t)prime * pairs_per_prime, hooks);
        f.store_last_layout(x, out, base, true);
        hooks.first = false;
        fr = hooks.next;
        if (fr >= total) break;
    }
    dloop_retire(ticket);
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
    f.blk = 0;
    f.split_log = 0;
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
    f.blk = 0;
    f.split_log = 0;
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

// Forward transform of frames of 2^(L+S) coefficients by workgroups that keep 2^L of them: block
// `blk` of a frame computes the S leading stages for its own contiguous 2^L outputs straight from
// global memory (reading the 2^S strided partners of each coefficient, so those stages' multiplies
// are done 2^S-1 times over) and then runs the resident transform.  Against a separate pass for the
// leading stages this saves 16n bytes of HBM traffic per stage and keeps the 64 KiB-per-frame
// occupancy (8 waves/SIMD) for n = 16384 and 32768.  NOT safe in place: every block reads the whole
// frame, so the host only selects it when out != in.
template <int L, int R, int PPB, int ARITH, int MINW, int S>
__global__ void __launch_bounds__((1 << (L - R)) * PPB, MINW)
fwd_rb2_split(const uint64_t* __restrict__ in, uint64_t* __restrict__ out,
              const prime_consts* __restrict__ consts, const twpair* __restrict__ tw_nat, const twpair* __restrict__ tw_rb,
              uint32_t pairs_per_prime, uint64_t frames_x, int64_t prime_stride, int64_t poly_stride, uint32_t lazy_out) {
    static_assert(S == 1 || S == 2, "one or two leading stages");
    using F = rb2_frame<L, R, (ARITH & 1) == 1, (ARITH >> 1), S>;
    constexpr int C = F::C, T = F::T;
    constexpr uint32_t split_log = S;
    F f;
    f.tid = threadIdx.x & (T - 1);
    const uint32_t slot = threadIdx.x / T;
    uint64_t fx = (uint64_t)blockIdx.x * PPB + slot;
    const bool live = fx < frames_x;
    if (!live) fx = frames_x - 1;
    const uint32_t prime = blockIdx.y;
    const uint64_t poly = fx >> split_log;
    f.blk = (uint32_t)(fx & ((1u << split_log) - 1u));
    f.split_log = split_log;
    f.lazy_out = lazy_out != 0;
    f.init_consts(consts[prime].q, consts[prime].est);
    f.slab = reinterpret_cast<uint64_t*>(agx_dyn_lds) + (size_t)slot * F::slab_elems;
    const int64_t frame = (int64_t)prime * prime_stride + (int64_t)poly * poly_stride;
    const int64_t base = frame + ((int64_t)f.blk << L);
    const twpair* nat = tw_nat + ((size_t)prime << (L + S));
    const uint32_t blk = f.blk;     // wave-uniform

    uint64_t x[C];
    if constexpr (S == 1) {
        const twpair w1 = load_uniform(nat + 1);
        static_for<0, C>([&](auto Rr) {
            constexpr int r = Rr;
            const int64_t e = frame + f.tid + (uint32_t)r * T;
            uint64_t a = in[e], b = in[e + (1 << L)];
            f.template butterfly<0>(a, b, w1);
            x[r] = blk ? b : a;
        });
    } else {
        const twpair w1 = load_uniform(nat + 1), w2 = load_uniform(nat + 2 + (blk >> 1));
        static_for<0, C>([&](auto Rr) {
            constexpr int r = Rr;
            // this block's half of the frame after stage 0 needs both stage-0 partners of its two quarters
            uint64_t lo0 = in[frame + f.tid + (uint32_t)r * T], hi0 = in[frame + f.tid + (uint32_t)r * T + (2 << L)];
            uint64_t lo1 = in[frame + f.tid + (uint32_t)r * T + (1 << L)], hi1 = in[frame + f.tid + (uint32_t)r * T + (3 << L)];
            f.template butterfly<0>(lo0, hi0, w1);
            f.template butterfly<0>(lo1, hi1, w1);
            uint64_t p = (blk >> 1) ? hi0 : lo0, q = (blk >> 1) ? hi1 : lo1;
            f.template butterfly<1>(p, q, w2);
            x[r] = (blk & 1) ? q : p;
        });
    }
    f.forward(x, tw_rb + (size_t)prime * pairs_per_prime);
    f.store_last_layout(x, out, base, live);
}

// Frames of 2^(L+1) coefficients, in-place safe, 16n bytes of traffic: one workgroup loads the whole
// frame (each thread its 2^R butterfly pairs of the leading stage), runs that stage once, then
// transforms the two resident halves one after the other through the same LDS image while the
// second half waits in registers.  Nothing is stored before everything has been loaded.
template <int L, int R, int ARITH, int MINW>
__global__ void __launch_bounds__((1 << (L - R)), MINW)
fwd_rb2_pair(const uint64_t* __restrict__ in, uint64_t* __restrict__ out,
             const prime_consts* __restrict__ consts, const twpair* __restrict__ tw_nat, const twpair* __restrict__ tw_rb,
             uint32_t pairs_per_prime, int64_t prime_stride, int64_t poly_stride, uint32_t lazy_out) {
    using F = rb2_frame<L, R, (ARITH & 1) == 1, (ARITH >> 1), 1>;
    constexpr int C = F::C, T = F::T;
    F f;
    f.tid = threadIdx.x;
    const uint32_t prime = blockIdx.y;
    f.split_log = 1;
    f.lazy_out = lazy_out != 0;
    f.init_consts(consts[prime].q, consts[prime].est);
    f.slab = reinterpret_cast<uint64_t*>(agx_dyn_lds);
    const int64_t frame = (int64_t)prime * prime_stride + (int64_t)blockIdx.x * poly_stride;
    const twpair* tbl = tw_rb + (size_t)prime * pairs_per_prime;
    const twpair w1 = load_uniform(tw_nat + ((size_t)prime << (L + 1)) + 1);

    uint64_t lo[C], hi[C];
    constexpr bool NT = ((ARITH >> 1) & kOptNtLoad) != 0;
    static_for<0, C>([&](auto Rr) { constexpr int r = Rr; lo[r] = NT ? __builtin_nontemporal_load(&in[frame + f.tid + (uint32_t)r * T]) : in[frame + f.tid + (uint32_t)r * T]; });
    static_for<0, C>([&](auto Rr) { constexpr int r = Rr; hi[r] = NT ? __builtin_nontemporal_load(&in[frame + (1 << L) + f.tid + (uint32_t)r * T]) : in[frame + (1 << L) + f.tid + (uint32_t)r * T]; });
    static_for<0, C>([&](auto Rr) { constexpr int r = Rr; f.template butterfly<0>(lo[r], hi[r], w1); });
    f.blk = 0;
    f.forward(lo, tbl);
    f.store_last_layout(lo, out, frame, true);
    __syncthreads();   // the second half's first exchange scatters over the whole image
    f.blk = 1;
    f.forward(hi, tbl);
    f.store_last_layout(hi, out, frame + (1 << L), true);
}

template <int L, int R, int PPB, int ARITH, int MINW>
__global__ void __launch_bounds__((1 << (L - R)) * PPB, MINW)
inv_rb2(const uint64_t* __restrict__ in, const uint64_t* __restrict__ in2, uint64_t* __restrict__ out,
        const prime_consts* __restrict__ consts, const twpair* __restrict__ itw_rb,
        uint32_t pairs_per_prime, uint32_t split_log, uint64_t frames_x,
        int64_t prime_stride, int64_t poly_stride) {
    uint64_t t_entry = 0;
    if constexpr (((ARITH >> 1) & kOptTrace) != 0) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_entry) : : "memory");
    if constexpr (((ARITH >> 1) & kOptPrio) != 0) __builtin_amdgcn_s_setprio(3);   // until the frame loads are out
    AGX_RB2_PROLOGUE;
    const prime_consts pc = consts[prime];
    const barrett128 bk{pc.q, pc.mu_hi, pc.mu_lo};
    uint64_t x[C];
    typename F::inv_pre pre;
    if constexpr (((ARITH >> 1) & kOptAblateHbm) != 0) {
        // timing only (wrong results): frame 0 of the prime (L2-resident) instead of the workgroup's own frame, for the loads, the stores or both
        const int64_t hot = (int64_t)prime * prime_stride;
        f.load_last_issue(x, in, (((ARITH >> 1) & kOptAblateStOnly) != 0) ? base : hot);
        f.inverse_prefetch(pre, itw_rb + (size_t)prime * pairs_per_prime);
        f.load_last_stage(x, in2, bk, base);
        f.inverse(x, itw_rb + (size_t)prime * pairs_per_prime, pc, pre);
        const int64_t ob = (((ARITH >> 1) & kOptAblateLdOnly) != 0) ? base : hot;
        if (live) {
#pragma unroll
            for (int r = 0; r < C; ++r) out[ob + f.tid + (uint32_t)r * T] = x[r];
        }
        return;
    }
    f.load_last_issue(x, in, base);
    f.inverse_prefetch(pre, itw_rb + (size_t)prime * pairs_per_prime);      // behind the frame loads, ahead of their wait
    if constexpr (F::TRACE) {
        f.ts[0] = t_entry;
        f.template stamp<1>(x[C - 1]);      // frame arrived
    }
    f.load_last_stage(x, in2, bk, base);
    if constexpr (F::TRACE) f.template stamp<2>(x[C - 1]);      // staged through the image into the last pass's layout
    f.inverse(x, itw_rb + (size_t)prime * pairs_per_prime, pc, pre);
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

// Inverse of frames of 2^(L+1) coefficients in one pass (the mirror of fwd_rb2_pair): one workgroup inverts
// the frame's two resident halves in turn, the first waiting in registers, and then runs the transform's
// last stage (gap 2^L, n^-1 folded in) on the register pairs -- instead of a separate inv_global_stage
// pass over HBM.  Everything is loaded before anything is stored, so in place is safe.
template <int L, int R, int ARITH, int MINW>
__global__ void __launch_bounds__((1 << (L - R)), MINW)
inv_rb2_pair(const uint64_t* __restrict__ in, const uint64_t* __restrict__ in2, uint64_t* __restrict__ out,
             const prime_consts* __restrict__ consts, const twpair* __restrict__ itw_rb,
             uint32_t pairs_per_prime, int64_t prime_stride, int64_t poly_stride) {
    using F = rb2_frame<L, R, (ARITH & 1) == 1, (ARITH >> 1)>;
    constexpr int C = F::C, T = F::T;
    constexpr bool FAST = (ARITH & 1) == 1;
    F f;
    f.tid = threadIdx.x;
    const uint32_t prime = blockIdx.y;
    f.split_log = 1;
    const prime_consts pc = consts[prime];
    f.init_consts(pc.q, pc.est);
    f.slab = reinterpret_cast<uint64_t*>(agx_dyn_lds);
    const barrett128 bk{pc.q, pc.mu_hi, pc.mu_lo};
    const int64_t frame = (int64_t)prime * prime_stride + (int64_t)blockIdx.x * poly_stride;
    const twpair* itbl = itw_rb + (size_t)prime * pairs_per_prime;

    uint64_t lo[C], hi[C];
    f.blk = 0;
    f.load_last_layout(lo, in, in2, bk, frame);
    f.inverse(lo, itbl, pc);
    __syncthreads();   // the second half's staging overwrites image words other waves may still be reading
    f.blk = 1;
    f.load_last_layout(hi, in, in2, bk, frame + (1 << L));
    f.inverse(hi, itbl, pc);
    static_for<0, C>([&](auto Rr) {
        constexpr int r = Rr;
        gs_last_form<FAST>(lo[r], hi[r], pc.n_inv, pc.n_inv_p, pc.w1n, pc.w1n_p, f.k);
        lo[r] = reduce_final_inv<FAST, F::SEL>(lo[r], f.k, f.fc);
        hi[r] = reduce_final_inv<FAST, F::SEL>(hi[r], f.k, f.fc);
    });
    constexpr bool NTS = ((ARITH >> 1) & kOptNtStore) != 0;
    static_for<0, C>([&](auto Rr) {
        constexpr int r = Rr;
        if constexpr (NTS) __builtin_nontemporal_store(lo[r], &out[frame + f.tid + (uint32_t)r * T]);
        else out[frame + f.tid + (uint32_t)r * T] = lo[r];
    });
    static_for<0, C>([&](auto Rr) {
        constexpr int r = Rr;
        if constexpr (NTS) __builtin_nontemporal_store(hi[r], &out[frame + (1 << L) + f.tid + (uint32_t)r * T]);
        else out[frame + (1 << L) + f.tid + (uint32_t)r * T] = hi[r];
    });
}

// c = INTT(NTT(a) o NTT(b)) for one frame without leaving the chip: both forward transforms end in
// the same register layout, the product is taken there, and the inverse starts from it (no staging
// through the image at either seam).  HBM traffic 24n bytes per product.
template <int L, int R, int PPB, int ARITH, int MINW>
__global__ void __launch_bounds__((1 << (L - R)) * PPB, (MINW > AGX_POLYMUL_MAXW ? AGX_POLYMUL_MAXW : MINW))   // one frame more in registers
polymul_rb2(const uint64_t* __restrict__ a, const uint64_t* __restrict__ b, uint64_t* __restrict__ c,
            const prime_consts* __restrict__ consts, const twpair* __restrict__ tw_rb, const twpair* __restrict__ itw_rb,
            uint32_t pairs_per_prime, uint64_t frames_x, int64_t prime_stride, int64_t poly_stride) {
    constexpr uint32_t split_log = 0;
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
    constexpr uint32_t split_log = 0;
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
    if constexpr (F::STREAM_TW) {
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
        return;
    }
    f.store_last_layout(x, c, base, live);     // parked (coalesced, through the image)
#pragma unroll
    for (int r = 0; r < C; ++r) x[r] = NTL ? __builtin_nontemporal_load(&second[base + f.tid + (uint32_t)r * T]) : second[base + f.tid + (uint32_t)r * T];
    __syncthreads();   // the image is reused; and every wave's parked stores have been acknowledged (the barrier waits for them)
    f.forward(x, tw_rb + (size_t)prime * pairs_per_prime);
    {
        // NTT(first) back in the last pass's layout: lane-contiguous loads of the wave's own block (served by L2; the
        // non-temporal policy bypasses this CU's L1) redistributed through the wave's part of the image
        uint64_t z[C];
        f.load_last_issue(z, c, base);
        f.load_last_stage(z, nullptr, bk, base);
#pragma unroll
        for (int r = 0; r < C; ++r) x[r] = mul_mod_barrett(z[r], x[r], bk);
    }
    __syncthreads();   // nobody overwrites c's frame (below) before everybody has fetched its parked part
    f.inverse(x, itw_rb + (size_t)prime * pairs_per_prime, pc);
    if (live) {
#pragma unroll
        for (int r = 0; r < C; ++r) {
            if constexpr (NTS) __builtin_nontemporal_store(x[r], &c[base + f.tid + (uint32_t)r * T]);
            else c[base + f.tid + (uint32_t)r * T] = x[r];
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
    constexpr uint32_t split_log = 0;
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

template <int L, int R, bool col_major = false>
void build_table_t(const regblock_layout& rb, const uint64_t* tw, const uint64_t* pre, std::vector<ulonglong2>& out) {
    using G = rb_geom<L, R>;
    const uint32_t nblk = 1u << rb.log_split;
    const size_t start = out.size();
    out.resize(start + (size_t)G::table_pairs * nblk, make_ulonglong2(0, 0));
    for (int p = 0; p < G::NP; ++p) {
        const int rlo = G::rlo(p), hi = G::hi(p), H = G::H(p);
        ulonglong2* t = out.data() + start + (size_t)G::table_off(p) * nblk;
        for (int j = 1; j < G::C; ++j) {
            int k = 0;
            while ((2 << k) <= j) ++k;
            const int o = j - (1 << k), rb_bit = R - 1 - k, b = rlo + rb_bit;
            if (b > hi) continue;  // stage belongs to an earlier pass (short last pass)
            const uint32_t m_local = 1u << (L - 1 - b);
            for (uint32_t blk = 0; blk < nblk; ++blk)
                for (int h = 0; h < H; ++h) {
                    const uint32_t idx = (m_local << rb.log_split) + blk * m_local + ((uint32_t)h << k) + (uint32_t)o;
                    const size_t col = (size_t)blk * H + h;
                    // wave-uniform passes keep one column's C entries contiguous (wide scalar loads);
                    // per-lane passes keep one entry's columns contiguous (coalesced vector loads)
                    const size_t at = (col_major && rlo >= 6) ? col * G::C + j : (size_t)j * H * nblk + col;
                    t[at] = make_ulonglong2(tw[idx], pre[idx]);
                }
        }
    }
}

template <int L, int R, int PPB, bool STAGE_OUT, int MINW>
hipError_t launch_rb_t(const plan_view& pv, const uint64_t* in, uint64_t* out, const frame_layout& fl, hipStream_t s) {
    using G = rb_geom<L, R>;
    const uint64_t frames_x = fl.batch << pv.rb.log_split;
    const size_t lds = (size_t)G::lds_elems * 8 * PPB;
    dim3 grid((unsigned)((frames_x + PPB - 1) / PPB), pv.num_primes);
    hipLaunchKernelGGL((fwd_regblock<L, R, PPB, STAGE_OUT, MINW>), grid, dim3(G::T * PPB), lds, s, in, out, pv.consts, pv.tw_rb,
                       pv.rb.pairs_per_prime, (uint32_t)pv.rb.log_split, frames_x, fl.prime_stride, fl.poly_stride);
    return hipGetLastError();
}

template <int L, int R, int PPB, bool STAGE_OUT, int MINW>
hipError_t init_rb_t() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&fwd_regblock<L, R, PPB, STAGE_OUT, MINW>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)rb_geom<L, R>::lds_elems * 8 * PPB));
}

template <int L, int R, int PPB, int ARITH>
constexpr size_t rb2_lds_bytes() {
    using F = rb2_frame<L, R, (ARITH & 1) == 1, (ARITH >> 1)>;
    return (size_t)F::slab_elems * (F::SPLIT ? 4 : 8) * PPB;
}

template <int L, int R, int PPB, int ARITH, int MINW>
hipError_t launch_rb2_t(const plan_view& pv, const uint64_t* in, uint64_t* out, const frame_layout& fl, hipStream_t s) {
    using G = rb_geom<L, R>;
    const uint64_t frames_x = fl.batch << pv.rb.log_split;
    const size_t lds = rb2_lds_bytes<L, R, PPB, ARITH>();
    dim3 grid((unsigned)((frames_x + PPB - 1) / PPB), pv.num_primes);
    hipLaunchKernelGGL((fwd_rb2<L, R, PPB, ARITH, MINW>), grid, dim3(G::T * PPB), lds, s, in, out, pv.consts, pv.tw_rb,
                       pv.rb.pairs_per_prime, (uint32_t)pv.rb.log_split, frames_x, fl.prime_stride, fl.poly_stride,
                       (uint32_t)(fl.lazy_out ? 1 : 0));
    return hipGetLastError();
}


template <int L, int R, int PPB, int ARITH, int MINW>
hipError_t launch_inv_rb2_t(const plan_view& pv, const uint64_t* in, const uint64_t* in2, uint64_t* out, const frame_layout& fl, hipStream_t s) {
    using G = rb_geom<L, R>;
    const uint64_t frames_x = fl.batch << pv.rb.log_split;
    dim3 grid((unsigned)((frames_x + PPB - 1) / PPB), pv.num_primes);
    const size_t lds = rb2_lds_bytes<L, R, PPB, ARITH>();
    hipLaunchKernelGGL((inv_rb2<L, R, PPB, ARITH, MINW>), grid, dim3(G::T * PPB), lds, s, in, in2, out, pv.consts,
                       pv.itw_rb, pv.rb.pairs_per_prime, (uint32_t)pv.rb.log_split, frames_x, fl.prime_stride, fl.poly_stride);
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

template <int L, int R, int PPB, int ARITH, int MINW>
constexpr rb_entry make_entry2(int id) {
    return rb_entry{id, L, R, PPB, true, MINW, (uint32_t)rb_geom<L, R>::table_pairs, rb2_lds_bytes<L, R, PPB, ARITH>(),
                    &build_table_t<L, R, true>, &launch_rb2_t<L, R, PPB, ARITH, MINW>, &init_rb2_t<L, R, PPB, ARITH, MINW>,
                    (ARITH & 1) ? ((((ARITH >> 1) & kOptLazy16) != 0) ? 2 : 1) : 0,
                    &launch_inv_rb2_t<L, R, PPB, ARITH, MINW>, &launch_mul_rb2_t<L, R, PPB, ARITH, MINW>, 0, nullptr, false};
}

// forward kernel only (no inverse / fused product instantiated: the plan's inverse falls back to the radix-2 kernel)
template <int L, int R, int PPB, int ARITH, int MINW>
hipError_t init_rb2_fwd_only_t() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&fwd_rb2<L, R, PPB, ARITH, MINW>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)rb2_lds_bytes<L, R, PPB, ARITH>());
}
template <int L, int R, int PPB, int ARITH, int MINW>
constexpr rb_entry make_entry_fwd_only(int id) {
    return rb_entry{id, L, R, PPB, true, MINW, (uint32_t)rb_geom<L, R>::table_pairs, rb2_lds_bytes<L, R, PPB, ARITH>(),
                    &build_table_t<L, R, true>, &launch_rb2_t<L, R, PPB, ARITH, MINW>, &init_rb2_fwd_only_t<L, R, PPB, ARITH, MINW>,
                    (ARITH & 1) ? ((((ARITH >> 1) & kOptLazy16) != 0) ? 2 : 1) : 0, nullptr, nullptr, 0, nullptr, false};
}

template <int L, int R, int ARITH, int MINW, int PF>
hipError_t launch_rb2_stream_t(const plan_view& pv, const uint64_t* in, uint64_t* out, const frame_layout& fl, hipStream_t s) {
    using G = rb_geom<L, R>;
    int dev = 0, cus = 0;
    hipError_t de = hipGetDevice(&dev);
    if (de == hipSuccess) de = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (de != hipSuccess) return de;
    const int resident = cus * (MINW * 256 / G::T);   // workgroups the current device holds at MINW waves per SIMD
    const uint64_t total = fl.batch * pv.num_primes;
    uint32_t* ticket = pv.ticket(s);
    if (total >= (1ull << 31) || !ticket) return hipErrorInvalidValue;
    const size_t lds = rb2_lds_bytes<L, R, 1, ARITH>() + 16;
    const unsigned grid = (unsigned)(total < (uint64_t)resident ? total : (uint64_t)resident);
    hipLaunchKernelGGL((fwd_rb2_stream<L, R, ARITH, MINW, PF>), dim3(grid), dim3(G::T), lds, s, in, out, pv.consts, pv.tw_rb,
                       pv.rb.pairs_per_prime, (uint32_t)fl.batch, (uint32_t)total, fl.prime_stride, fl.poly_stride,
                       (uint32_t)(fl.lazy_out ? 1 : 0), ticket);
    return hipGetLastError();
}

template <int L, int R, int ARITH, int MINW, int PF>
hipError_t init_rb2_stream_t() {
    hipError_t e = init_rb2_t<L, R, 1, ARITH, MINW>();
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fwd_rb2_stream<L, R, ARITH, MINW, PF>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)rb2_lds_bytes<L, R, 1, ARITH>() + 16);
    return e;
}

template <int L, int R, int ARITH, int MINW, int PF = 0>
constexpr rb_entry make_entry_stream(int id) {
    rb_entry e = make_entry2<L, R, 1, ARITH, MINW>(id);
    e.launch = &launch_rb2_stream_t<L, R, ARITH, MINW, PF>;
    e.init = &init_rb2_stream_t<L, R, ARITH, MINW, PF>;
    return e;
}

template <int L, int R, int PPB, int ARITH, int MINW, int S>
hipError_t launch_rb2_split_t(const plan_view& pv, const uint64_t* in, uint64_t* out, const frame_layout& fl, hipStream_t s) {
    using G = rb_geom<L, R>;
    const uint64_t frames_x = fl.batch << S;
    const size_t lds = rb2_lds_bytes<L, R, PPB, ARITH>();
    dim3 grid((unsigned)((frames_x + PPB - 1) / PPB), pv.num_primes);
    hipLaunchKernelGGL((fwd_rb2_split<L, R, PPB, ARITH, MINW, S>), grid, dim3(G::T * PPB), lds, s, in, out, pv.consts, pv.tw, pv.tw_rb,
                       pv.rb.pairs_per_prime, frames_x, fl.prime_stride, fl.poly_stride, (uint32_t)(fl.lazy_out ? 1 : 0));
    return hipGetLastError();
}

template <int L, int R, int PPB, int ARITH, int MINW, int S>
hipError_t init_rb2_split_t() {
    hipError_t e = init_rb2_t<L, R, PPB, ARITH, MINW>();
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fwd_rb2_split<L, R, PPB, ARITH, MINW, S>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)rb2_lds_bytes<L, R, PPB, ARITH>());
    return e;
}

template <int L, int R, int ARITH, int MINW>
hipError_t launch_rb2_pair_t(const plan_view& pv, const uint64_t* in, uint64_t* out, const frame_layout& fl, hipStream_t s) {
    using G = rb_geom<L, R>;
    const size_t lds = rb2_lds_bytes<L, R, 1, ARITH>();
    dim3 grid((unsigned)fl.batch, pv.num_primes);
    hipLaunchKernelGGL((fwd_rb2_pair<L, R, ARITH, MINW>), grid, dim3(G::T), lds, s, in, out, pv.consts, pv.tw, pv.tw_rb,
                       pv.rb.pairs_per_prime, fl.prime_stride, fl.poly_stride, (uint32_t)(fl.lazy_out ? 1 : 0));
    return hipGetLastError();
}

template <int L, int R, int ARITH, int MINW>
hipError_t init_rb2_pair_t() {
    hipError_t e = init_rb2_t<L, R, 1, ARITH, MINW>();
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fwd_rb2_pair<L, R, ARITH, MINW>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)rb2_lds_bytes<L, R, 1, ARITH>());
    return e;
}

template <int L, int R, int ARITH, int MINW>
hipError_t launch_inv_rb2_pair_t(const plan_view& pv, const uint64_t* in, const uint64_t* in2, uint64_t* out, const frame_layout& fl, hipStream_t s) {
    using G = rb_geom<L, R>;
    const size_t lds = rb2_lds_bytes<L, R, 1, ARITH>();
    dim3 grid((unsigned)fl.batch, pv.num_primes);
    hipLaunchKernelGGL((inv_rb2_pair<L, R, ARITH, MINW>), grid, dim3(G::T), lds, s, in, in2, out, pv.consts, pv.itw_rb,
                       pv.rb.pairs_per_prime, fl.prime_stride, fl.poly_stride);
    return hipGetLastError();
}

template <int L, int R, int ARITH, int MINW>
hipError_t init_rb2_invpair_t() {
    hipError_t e = init_rb2_t<L, R, 1, ARITH, MINW>();
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&inv_rb2_pair<L, R, ARITH, MINW>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)rb2_lds_bytes<L, R, 1, ARITH>());
    return e;
}

// a second-generation entry whose inverse at n = 2^(L+1) is the one-launch pair kernel
template <int L, int R, int ARITH, int MINW>
constexpr rb_entry make_entry2_invpair(int id) {
    rb_entry e = make_entry2<L, R, 1, ARITH, MINW>(id);
    e.init = &init_rb2_invpair_t<L, R, ARITH, MINW>;
    e.launch_inv_pair = &launch_inv_rb2_pair_t<L, R, ARITH, MINW>;
    return e;
}

// loop kernels: grid = the workgroups the current device holds at once (MINW waves per SIMD), fixed-stride walk
template <int L, int R, int MINW>
hipError_t resident_workgroups(unsigned* out) {
    int dev = 0, cus = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (e == hipSuccess) *out = (unsigned)cus * (unsigned)(MINW * 256 / rb_geom<L, R>::T);
    return e;
}

template <int L, int R, int ARITH, int MINW>
hipError_t launch_rb2_loop_t(const plan_view& pv, const uint64_t* in, uint64_t* out, const frame_layout& fl, hipStream_t s) {
    using G = rb_geom<L, R>;
    unsigned resident = 0;
    hipError_t e = resident_workgroups<L, R, MINW>(&resident);
    if (e != hipSuccess) return e;
    const uint64_t total = fl.batch * pv.num_primes;
    if (total >= (1ull << 31)) return hipErrorInvalidValue;
    const unsigned grid = (unsigned)(total < resident ? total : resident);
    const size_t lds = rb2_lds_bytes<L, R, 1, ARITH>();
    hipLaunchKernelGGL((fwd_rb2_loop<L, R, ARITH, MINW>), dim3(grid), dim3(G::T), lds, s, in, out, pv.consts, pv.tw_rb,
                       pv.rb.pairs_per_prime, (uint32_t)fl.batch, (uint32_t)total, fl.prime_stride, fl.poly_stride, (uint32_t)(fl.lazy_out ? 1 : 0));
    return hipGetLastError();
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

template <int L, int R, int ARITH, int MINW>
hipError_t init_rb2_loop_t();

// A launch recorded into a hipGraph would keep its ticket pair for as long as the graph lives, while eager launches keep
// walking round the plan's ring; captured launches therefore take the fixed-stride loop kernels, which carry no state.
inline bool stream_is_capturing(hipStream_t s) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    return hipStreamIsCapturing(s, &st) == hipSuccess && st == hipStreamCaptureStatusActive;
}

template <int L, int R, int ARITH, int MINW>
hipError_t launch_rb2_dloop_t(const plan_view& pv, const uint64_t* in, uint64_t* out, const frame_layout& fl, hipStream_t s) {
    using G = rb_geom<L, R>;
    if (stream_is_capturing(s)) return launch_rb2_loop_t<L, R, ARITH, MINW>(pv, in, out, fl, s);
    uint32_t* ticket = pv.ticket(s);
    if (!ticket) return launch_rb2_loop_t<L, R, ARITH, MINW>(pv, in, out, fl, s);      // no pair provably free: stateless form
    unsigned resident = 0;
    hipError_t e = resident_workgroups<L, R, MINW>(&resident);
    if (e != hipSuccess) return e;
    const uint64_t total = fl.batch * pv.num_primes;
    if (total >= (1ull << 31)) return hipErrorInvalidValue;
    const unsigned grid = (unsigned)(total < resident ? total : resident);
    const size_t lds = rb2_lds_bytes<L, R, 1, ARITH>() + 16;
    hipLaunchKernelGGL((fwd_rb2_dloop<L, R, ARITH, MINW>), dim3(grid), dim3(G::T), lds, s, in, out, pv.consts, pv.tw_rb,
                       pv.rb.pairs_per_prime, (uint32_t)fl.batch, (uint32_t)total, fl.prime_stride, fl.poly_stride, (uint32_t)(fl.lazy_out ? 1 : 0), ticket);
    return hipGetLastError();
}

template <int L, int R, int ARITH, int MINW>
hipError_t launch_inv_rb2_dloop_t(const plan_view& pv, const uint64_t* in, const uint64_t* in2, uint64_t* out, const frame_layout& fl, hipStream_t s) {
    using G = rb_geom<L, R>;
    if (stream_is_capturing(s)) return launch_inv_rb2_loop_t<L, R, ARITH, MINW>(pv, in, in2, out, fl, s);
    uint32_t* ticket = pv.ticket(s);
    if (!ticket) return launch_inv_rb2_loop_t<L, R, ARITH, MINW>(pv, in, in2, out, fl, s);
    unsigned resident = 0;
    hipError_t e = resident_workgroups<L, R, MINW>(&resident);
    if (e != hipSuccess) return e;
    const uint64_t total = fl.batch * pv.num_primes;
    if (total >= (1ull << 31)) return hipErrorInvalidValue;
    const unsigned grid = (unsigned)(total < resident ? total : resident);
    const size_t lds = rb2_lds_bytes<L, R, 1, ARITH>() + 16;
    hipLaunchKernelGGL((inv_rb2_dloop<L, R, ARITH, MINW>), dim3(grid), dim3(G::T), lds, s, in, in2, out, pv.consts, pv.itw_rb,
                       pv.rb.pairs_per_prime, (uint32_t)fl.batch, (uint32_t)total, fl.prime_stride, fl.poly_stride, ticket);
    return hipGetLastError();
}

template <int L, int R, int ARITH, int MINW>
hipError_t init_rb2_dloop_t() {
    hipError_t e = init_rb2_loop_t<L, R, ARITH, MINW>();      // (the fixed-stride loop kernels serve captured launches)
    const int bytes = (int)rb2_lds_bytes<L, R, 1, ARITH>() + 16;
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fwd_rb2_dloop<L, R, ARITH, MINW>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&inv_rb2_dloop<L, R, ARITH, MINW>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    return e;
}

// forward and inverse by the dynamic loop kernels (FWD / INV select which of the two; the other stays one workgroup per frame)
template <hipError_t (*BASE)(), int L, int R, int ARITH, int MINW>
hipError_t init_plus_park_t() {
    hipError_t e = BASE();
    if (e == hipSuccess) e = init_mul_park_t<L, R, ARITH, MINW>();
    return e;
}

// PARK: the fused product by polymul_rb2_park (one frame in registers, the other parked in c's frame)
template <int L, int R, int ARITH, int MINW, bool FWD, bool INV, bool PARK = false>
constexpr rb_entry make_entry_dloop(int id) {
    rb_entry e = make_entry2_invpair<L, R, ARITH, MINW>(id);
    e.init = PARK ? &init_plus_park_t<&init_rb2_dloop_t<L, R, ARITH, MINW>, L, R, ARITH, MINW> : &init_rb2_dloop_t<L, R, ARITH, MINW>;
    if (FWD) e.launch = &launch_rb2_dloop_t<L, R, ARITH, MINW>;
    if (INV) e.launch_inv_loop = &launch_inv_rb2_dloop_t<L, R, ARITH, MINW>;
    if (PARK) {
        e.launch_mul = &launch_mul_park_t<L, R, ARITH, MINW>;
        e.mul_parked = true;
    }
    return e;
}

// whole-frame kernels with one frame in registers per workgroup at any time (R = 5: a second frame cannot be held):
// forward, inverse, and the fused product by polymul_rb2_park; polymul_rb2 (two frames in registers) is not instantiated
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
    rb_entry e{id, L, R, 1, true, MINW, (uint32_t)rb_geom<L, R>::table_pairs, rb2_lds_bytes<L, R, 1, ARITH>(),
               &build_table_t<L, R, true>, &launch_rb2_t<L, R, 1, ARITH, MINW>, &init_rb2_single_t<L, R, ARITH, MINW>,
               (ARITH & 1) ? ((((ARITH >> 1) & kOptLazy16) != 0) ? 2 : 1) : 0,
               &launch_inv_rb2_t<L, R, 1, ARITH, MINW>, &launch_mul_park_t<L, R, ARITH, MINW>, 0, nullptr, false};
    e.mul_parked = true;
    e.whole_only = true;       // the streamed inverse folds n^-1 into its top stage
    return e;
}

// entry e with forward calls of its plans routed to the forward-only entry `id` (rb_entry::fwd_companion)
constexpr rb_entry with_fwd_companion(rb_entry e, int id, uint32_t min_frames = 0) {
    e.fwd_companion = id;
    e.fwd_companion_min_frames = min_frames;
    return e;
}

// forward kernel only of a streamed single-frame shape (a plan's forward companion: rb_entry::fwd_companion)
template <int L, int R, int ARITH, int MINW>
constexpr rb_entry make_entry_single_fwd(int id) {
    rb_entry e = make_entry_fwd_only<L, R, 1, ARITH, MINW>(id);
    e.whole_only = true;
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
    e.mul_parked = false;
    return e;
}

// ... with the forward (FWD) and / or inverse (INV) launches by the dynamic loop kernels (resident grid, ticket counter; captured
// launches take the fixed-stride form)
template <int L, int R, int ARITH, int MINW>
hipError_t init_rb2_single_dloop_t() {
    hipError_t e = init_rb2_single_t<L, R, ARITH, MINW>();
    const int bytes = (int)rb2_lds_bytes<L, R, 1, ARITH>();
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fwd_rb2_loop<L, R, ARITH, MINW>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&inv_rb2_loop<L, R, ARITH, MINW>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fwd_rb2_dloop<L, R, ARITH, MINW>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes + 16);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&inv_rb2_dloop<L, R, ARITH, MINW>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes + 16);
    return e;
}
template <int L, int R, int ARITH, int MINW, bool FWD, bool INV>
constexpr rb_entry make_entry_single_dloop(int id) {
    rb_entry e = make_entry_single<L, R, ARITH, MINW>(id);
    e.init = &init_rb2_single_dloop_t<L, R, ARITH, MINW>;
    if (FWD) e.launch = &launch_rb2_dloop_t<L, R, ARITH, MINW>;
    if (INV) e.launch_inv_loop = &launch_inv_rb2_dloop_t<L, R, ARITH, MINW>;
    return e;
}

// a plain second-generation entry whose fused product is the parked-operand kernel
template <int L, int R, int ARITH, int MINW>
constexpr rb_entry make_entry2_park(int id) {
    rb_entry e = make_entry2<L, R, 1, ARITH, MINW>(id);
    e.init = &init_plus_park_t<&init_rb2_t<L, R, 1, ARITH, MINW>, L, R, ARITH, MINW>;
    e.launch_mul = &launch_mul_park_t<L, R, ARITH, MINW>;
    e.mul_parked = true;
    return e;
}

template <int L, int R, int ARITH, int MINW>
hipError_t init_rb2_loop_t() {
    hipError_t e = init_rb2_invpair_t<L, R, ARITH, MINW>();
    const int bytes = (int)rb2_lds_bytes<L, R, 1, ARITH>();
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fwd_rb2_loop<L, R, ARITH, MINW>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&inv_rb2_loop<L, R, ARITH, MINW>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    return e;
}

// a second-generation entry whose forward and inverse launches are the loop kernels (split_log = 0 only; the
// n = 2^(L+1) inverse stays the one-launch pair kernel)
template <int L, int R, int ARITH, int MINW, bool FWD_LOOP = true>
constexpr rb_entry make_entry_loop(int id) {
    rb_entry e = make_entry2_invpair<L, R, ARITH, MINW>(id);
    e.init = &init_rb2_loop_t<L, R, ARITH, MINW>;
    if (FWD_LOOP) e.launch = &launch_rb2_loop_t<L, R, ARITH, MINW>;
    e.launch_inv_loop = &launch_inv_rb2_loop_t<L, R, ARITH, MINW>;
    return e;
}

// n = 2^(L+1) with one workgroup per frame transforming its two halves in turn (in-place safe);
// the inverse of such a plan runs on the 2^L blocks + inv_global_stage
template <int L, int R, int ARITH, int MINW>
constexpr rb_entry make_entry_pair(int id) {
    rb_entry e = make_entry2<L, R, 1, ARITH, MINW>(id);
    e.init = &init_rb2_pair_t<L, R, ARITH, MINW>;
    e.fused_split = 1;
    e.launch_fused = &launch_rb2_pair_t<L, R, ARITH, MINW>;
    e.fused_in_place_ok = true;
    return e;
}

// a second-generation entry for n = 2^(L+S): resident blocks of 2^L, leading S stages fused into the
// forward kernel when out != in (in place they run as separate fwd_global_stage passes)
template <int L, int R, int PPB, int ARITH, int MINW, int S>
constexpr rb_entry make_entry_split(int id) {
    rb_entry e = make_entry2<L, R, PPB, ARITH, MINW>(id);
    e.init = &init_rb2_split_t<L, R, PPB, ARITH, MINW, S>;
    e.fused_split = S;
    e.launch_fused = &launch_rb2_split_t<L, R, PPB, ARITH, MINW, S>;
    return e;
}

template <int L, int R, int PPB, bool STAGE_OUT, int MINW>
constexpr rb_entry make_entry(int id) {
    return rb_entry{id, L, R, PPB, STAGE_OUT, MINW, (uint32_t)rb_geom<L, R>::table_pairs, (size_t)rb_geom<L, R>::lds_elems * 8 * PPB,
                    &build_table_t<L, R>, &launch_rb_t<L, R, PPB, STAGE_OUT, MINW>, &init_rb_t<L, R, PPB, STAGE_OUT, MINW>, 0, nullptr, nullptr, 0, nullptr, false};
}

}  // namespace AGX_TU
}  // namespace agx
