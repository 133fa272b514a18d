// rb_frame.hpp -- geometry, option bits and the per-frame pass machinery (rb2_frame) of the register-blocked gfx950 kernels.
// rb_kernels.hpp wraps it into kernels and launch glue; the reg_*.hip translation units instantiate registry entries.
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
// geometry of the register-blocked kernels.
//
// n = 2^L per frame, C = 2^R coefficients per thread, T = n / C
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
};

// ---------------------------------------------------------------------------------------
// the register-blocked kernels (the throughput path): forward, inverse and fused polynomial product share
// rb2_frame below.  What shaped it, each item from a measurement:
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

// option bits of the kernels (the tuned sets are in the registry groups; values are stable: rocprof kernel names carry them)
constexpr int kOptPad = 1;       // padded LDS image, exchanges addressed base + immediate offset
constexpr int kOptSelect = 2;    // conditional subtract by compare + select instead of sign mask
constexpr int kOptLazy16 = 16;   // q <= 2^60: 16q-lazy forward butterflies (conditional subtract on 5 of 12 stages)
constexpr int kOptTwAhead = 32;   // per-lane passes: first three table entries fetched one pass early, the rest at pass start
constexpr int kOptTrace = 64;    // diagnostics (tools/timeline.py): every wave records s_memtime at 12 phase boundaries
constexpr int kOptPrio = 128;    // s_setprio 3 while a wave issues its frame loads
constexpr int kOptPrioBarrier = 512;   // with kOptPrio: stay at priority until the frame's one s_barrier has been passed
constexpr int kOptScalarBase = 1024;   // frame loads as (uniform pointer per register) + lane offset: no per-load VALU address arithmetic
constexpr int kOptLazyInv = 2048;      // with kOptLazy16: inverse butterflies keep sums up to 16q (26 instead of 48 conditional subtracts per thread at n=4096)
constexpr int kOptTwAheadInv = 4096;   // inverse: the next per-lane pass's first-stage twiddles (entries 4..7) fetched during the current pass's last stage
constexpr int kOptNtLoad = 131072, kOptNtStore = 262144;   // non-temporal frame loads / result stores (data touched once)
constexpr int kOptEstReduce = 524288;   // with kOptLazy16: tail-free subtract schedule + quotient-estimate final reduction
constexpr int kOptSplitWord = 1048576;  // exchanges move the low and the high 32-bit words in turn through an image of HALF the size (4n bytes): what lets an
                                        // n = 32768 frame (136 KiB image) be resident at all, and two n = 16384 workgroups share a CU
constexpr int kOptStreamTw = 1 << 25;      // twiddles streamed in chunks of a few table entries, one chunk requested ahead of the one in use, scheduling fenced per chunk:
                                           // bounds the registers a pass's table entries occupy (R = 5: 31 entries per pass would be 124 SGPRs / VGPRs if fetched up front);
                                           // per-lane entries are addressed as wave-uniform base (SGPRs) + 32-bit lane index, and the last stage branches once per chunk
                                           // on the final-reduction mode instead of once per coefficient
constexpr int kOptPinBf = 1 << 26;         // with kOptStreamTw: butterflies are pinned in program order (their operands pass through ordered empty asm statements), so the
                                           // instruction selector cannot start the partial products of the whole stage at once -- what takes an R = 5 pass from ~205 VGPRs to
                                           // the 128 a 1024-thread workgroup may use; one wave cannot issue faster than one VALU per ~8 clocks anyway (ILP buys nothing there)
constexpr int kOptStreamCh1 = 1 << 29;     // with kOptStreamTw: one table entry per chunk instead of two or four (R = 4 passes then fit 64 VGPRs: 8 waves/SIMD)
// Measured and removed in round 4 (the records stay in profiles/ and DESIGN.md 3.4-3.6): the XOR-swizzled image, cross products as 32-bit multiplies,
// the inverse's twiddle-first / priority policies, timing ablations, persistent streaming / loop forms of the forward, resident sub-blocks with
// recomputed or separate leading stages (split / pair kernels), the first-generation portable-butterfly kernel.

// where the kOptTrace kernels write: [wave][16] words, set through agx_ntt_debug_set_trace_buffer
__device__ uint64_t* g_trace_buf = nullptr;     // one copy per translation unit (namespace AGX_TU); only reg_diag.hip uses it
__device__ uint64_t g_trace_waves = 0;

// per-frame state shared by the kernels
template <int L, int R, bool FAST, int OPT = 0>
struct rb2_frame {
    using G = rb2_geom<L, R>;
    static constexpr int C = G::C, T = G::T, NP = G::NP;
    static constexpr bool SEL = (OPT & kOptSelect) != 0, LAZY16 = FAST && (OPT & kOptLazy16) != 0;
    static_assert((OPT & kOptPad) != 0, "every kernel uses the padded image");
    static constexpr bool TWA = (OPT & kOptTwAhead) != 0 && R >= 3;
    static constexpr bool TRACE = (OPT & kOptTrace) != 0;
    static constexpr bool PRIO = (OPT & kOptPrio) != 0;
    static constexpr bool PRIO_BARRIER = PRIO && (OPT & kOptPrioBarrier) != 0, SCALAR_BASE = (OPT & kOptScalarBase) != 0;
    static constexpr bool LAZY_INV = LAZY16 && (OPT & kOptLazyInv) != 0;
    static constexpr bool NT_LOAD = (OPT & kOptNtLoad) != 0;
    static constexpr bool TWA_INV = (OPT & kOptTwAheadInv) != 0 && R == 3;
    static constexpr bool EST = LAZY16 && SEL && (OPT & kOptEstReduce) != 0;
    static constexpr bool SPLIT = (OPT & kOptSplitWord) != 0;
    static constexpr bool STREAM_TW = (OPT & kOptStreamTw) != 0;
    static constexpr bool SADDR_TW = STREAM_TW, FINAL_MODE = STREAM_TW;
    static_assert(!EST || lazy16_tailfree::valid(L), "tail-free schedule must keep every stage within 16q");
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
    static constexpr uint32_t slab_elems = (1u << L) + (L >= PADS ? (1u << (L >= PADS ? L - PADS : 0)) : 0u);      // frames below 2^PADS coefficients (wave-packed kernels, one lane per frame) never exchange: no pad of their own
    static constexpr uint32_t image_bytes = slab_elems * (((OPT & kOptSplitWord) != 0) ? 4u : 8u);   // one frame's LDS image
    // image word of coefficient e: additive over disjoint bit fields, which is what lets an exchange address
    // register r as (thread base) + compile-time constant
    static __device__ __forceinline__ constexpr uint32_t img(uint32_t e) { return e + (e >> PADS); }
    static __device__ __forceinline__ constexpr uint32_t join(uint32_t base, uint32_t delta) { return base + delta; }
    uint32_t tid;
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
        if constexpr (EST) ct_butterfly_lazy16<SEL, lazy16_tailfree::subtracts(stage, L), false>(a, b, w.x, w.y, k, fc);
        else if constexpr (LAZY16) ct_butterfly_lazy16<SEL, lazy16_schedule::subtracts(stage), stage == L - 1>(a, b, w.x, w.y, k, fc);
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
            const twpair* ucol = tbl + G::table_off(p) + (size_t)hcol * C;
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
            t.hstride = (uint32_t)H;
        }
    }
    // Entry j of per-lane pass p's table for this lane, addressed as (wave-uniform base of the entry, in SGPRs) + (32-bit lane index):
    // the load takes the saddr form and no entry needs a 64-bit VGPR address of its own (R = 5: 31 entries per pass would be 62 VGPRs).
    template <int p>
    __device__ __forceinline__ twpair lane_entry(const twpair* tbl, int j) const {
        constexpr int rlo = G::rlo(p), H = G::H(p);
        if constexpr (!SADDR_TW) {
            const twpair* col = tbl + G::table_off(p) + (tid >> rlo);
            return col[(size_t)j * (uint32_t)H];
        }
        const twpair* base = tbl + G::table_off(p) + (size_t)j * (uint32_t)H;
        return base[tid >> rlo];
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
    template <int P0, int P1>
    __device__ __forceinline__ void forward_passes_streamed(uint64_t (&x)[C], const twpair* tbl) const {
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
                        constexpr int stage = L - 1 - (rlo + rb);
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
                if constexpr (SPLIT) split_exchange<p>(x);
                else {
                    image_write<p>(x);
                    exchange_sync<p>();
                }
                if constexpr (PRIO_BARRIER && !G::exchange_is_wave_local(p)) __builtin_amdgcn_s_setprio(0);
            }
        });
    }

    template <int P0, int P1>
    __device__ __forceinline__ void forward_passes(uint64_t (&x)[C], const twpair* tbl) const {
        if constexpr (STREAM_TW) {
            forward_passes_streamed<P0, P1>(x, tbl);
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
                        constexpr int stage = L - 1 - (rlo + rb);   // position in the whole transform
                        butterfly<stage>(x[r0], x[r1], w);
                        if constexpr (last_stage) {
                            x[r0] = final_reduce<decltype(M)::value>(x[r0]);
                            x[r1] = final_reduce<decltype(M)::value>(x[r1]);
                        }
                    });
                };
                if constexpr (last_stage) with_final_mode(stage_body);
                else stage_body(std::integral_constant<int, 0>{});
            });
            if constexpr (2 * p + 2 < 12) stamp<2 * p + 2>(x[C - 1]);
            if constexpr (p < NP - 1) {
                // A thread overwrites exactly the image words it read for this pass, so no other
                // thread can still need them: only the read side of an exchange has to be ordered.
                if constexpr (SPLIT) {
                    static_assert(!SPLIT || (P0 == 0 && P1 == NP), "split-word exchanges run the whole transform in one call");
                    split_exchange<p>(x);
                    if constexpr (PRIO_BARRIER && !G::exchange_is_wave_local(p)) __builtin_amdgcn_s_setprio(0);
                } else {
                image_write<p>(x);
                if constexpr (p < P1 - 1) {
                    exchange_sync<p>();
                    if constexpr (PRIO_BARRIER && !G::exchange_is_wave_local(p)) __builtin_amdgcn_s_setprio(0);
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

    static constexpr int ilog2(int v) { return v <= 1 ? 0 : 1 + ilog2(v >> 1); }
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
    // ascending), x in pass-0 layout, fully reduced; the top stage also multiplies by n^-1.
    __device__ __forceinline__ void inverse(uint64_t (&x)[C], const twpair* itbl, const prime_consts& pc) const {
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
                    if constexpr (top_stage) {
                        if constexpr (LAZY_INV) gs_last_lazy16<BND, SEL>(x[r0], x[r1], pc.n_inv, pc.n_inv_p, pc.w1n, pc.w1n_p, k, fc);
                        else gs_last_form<FAST>(x[r0], x[r1], pc.n_inv, pc.n_inv_p, pc.w1n, pc.w1n_p, k);
                    } else {
                        constexpr int j = (1 << kk) + (r0 >> (rb + 1));
                        twpair w;
                        if constexpr (twa_have && j >= 4) w = first[j - 4];
                        else w = twiddle<p>(t, j);
                        if constexpr (LAZY_INV) gs_butterfly_lazy16<BND, SEL>(x[r0], x[r1], w.x, w.y, k, fc);
                        else gs_butterfly_form<FAST, SEL>(x[r0], x[r1], w.x, w.y, k);
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

}  // namespace AGX_TU
}  // namespace agx
