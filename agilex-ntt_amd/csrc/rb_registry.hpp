// rb_registry.hpp -- the registry of register-blocked kernel configurations, shared by the translation
// units that instantiate the kernels (reg_*.hip) and the one that looks entries up (ntt_kernels.hip).
#pragma once
#include "ntt_kernels.hpp"

namespace agx {

// ---- registry of register-blocked configurations -------------------------------------
// ids are stable handles (agx_ntt_plan_set_variant(plan, AGX_VARIANT_REGBLOCK_BASE + id); rocprof summaries and tests name them),
// not indices.  Every entry transforms WHOLE frames of n = 2^log_n coefficients resident on chip.
struct rb_entry {
    int id, log_n, r, ppb;   // ppb: frames per workgroup
    int min_waves;
    uint32_t table_pairs;   // pass-table length per prime, in {w,w'} pairs
    size_t lds_bytes;
    void (*build)(const regblock_layout&, const uint64_t*, const uint64_t*, std::vector<ulonglong2>&);
    hipError_t (*launch)(const plan_view&, const uint64_t*, uint64_t*, const frame_layout&, hipStream_t);
    hipError_t (*init)();
    int arith;   // 0: exact (reference op sequence, q < 2^62); 1: fast (q <= 2^61); 2: 16q-lazy (q <= 2^60)
    hipError_t (*launch_inv)(const plan_view&, const uint64_t*, const uint64_t*, uint64_t*, const frame_layout&, hipStream_t);      // in2 != null: in * in2 is transformed
    hipError_t (*launch_mul)(const plan_view&, const uint64_t*, const uint64_t*, uint64_t*, const frame_layout&, hipStream_t);      // c = INTT(NTT(a) o NTT(b)) in one launch
    int fwd_companion = 0;     // registry id of a forward-only entry that serves forward calls of a plan whose main entry is this one (0: none)
    uint32_t fwd_companion_min_frames = 0;   // ... for launches of at least this many frames (batch x primes): a shape with fewer threads per frame wins
                                             // on throughput but loses on the latency of a launch that does not fill the chip
    int narrow = 0;            // 0: 64-bit arithmetic; 1: 32-bit arithmetic, every modulus < 2^31; 2: every modulus < 2^30 (rb32_kernels.hpp)
};

struct rb_span {
    const rb_entry* first;
    size_t count;
};

// one group per translation unit, so the ~50 kernel instantiations compile in parallel
// product groups (their A/B extras are compiled in under AGX_DIAG)
rb_span rb_entries_n4096();
rb_span rb_entries_s1024();      // the streamed single-frame kernels, one group (translation unit) per size
rb_span rb_entries_s2048();
rb_span rb_entries_s4096();
rb_span rb_entries_s8192();
rb_span rb_entries_s16384();
rb_span rb_entries_s32768();
rb_span rb_entries_q32a();       // 32-bit arithmetic, tier 2 (every modulus < 2^30) / tier 1 (< 2^31)
rb_span rb_entries_q32b();
rb_span rb_entries_wp();         // wave-packed kernels of n = 32 ... 512 (wp_kernels.hpp): 64-bit arithmetic / 32-bit arithmetic
rb_span rb_entries_wp32();
#ifdef AGX_DIAG
rb_span rb_entries_diag();                                      // lib/libagxntt_diag.so only: the trace twin of the n = 4096 default (tools/timeline.py)
hipError_t regblock_set_trace(uint64_t* buf, uint64_t waves);   // where the trace kernel writes
#endif

}  // namespace agx
