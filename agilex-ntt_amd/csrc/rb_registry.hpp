// rb_registry.hpp -- the registry of register-blocked kernel configurations, shared by the translation
// units that instantiate the kernels (reg_*.hip) and the one that looks entries up (ntt_kernels.hip).
#pragma once
#include "ntt_kernels.hpp"

namespace agx {

static constexpr int kMaxLdsLog = 14;  // 16384 coefficients = 128 KiB of the CU's 160 KiB LDS

// ---- registry of register-blocked configurations -------------------------------------
// id 0..: first entry for a given log_local is the tuned default; the others are kept for
// A/B measurements (agx_ntt_plan_set_variant(plan, AGX_VARIANT_REGBLOCK_BASE + id)).
struct rb_entry {
    int id, log_local, r, ppb;
    bool stage_out;
    int min_waves;
    uint32_t table_pairs;   // per sub-block
    size_t lds_bytes;
    void (*build)(const regblock_layout&, const uint64_t*, const uint64_t*, std::vector<ulonglong2>&);
    hipError_t (*launch)(const plan_view&, const uint64_t*, uint64_t*, const frame_layout&, hipStream_t);
    hipError_t (*init)();
    int arith;   // 0: exact (reference op sequence, q < 2^62); 1: fast (q <= 2^61); 2: 16q-lazy (q <= 2^60)
    hipError_t (*launch_inv)(const plan_view&, const uint64_t*, const uint64_t*, uint64_t*, const frame_layout&, hipStream_t);
    hipError_t (*launch_mul)(const plan_view&, const uint64_t*, const uint64_t*, uint64_t*, const frame_layout&, hipStream_t);
    int fused_split;   // S > 0: `launch` is only for out != in and computes the S leading stages itself (n = 2^(log_local+S))
    hipError_t (*launch_fused)(const plan_view&, const uint64_t*, uint64_t*, const frame_layout&, hipStream_t);
    bool fused_in_place_ok;   // launch_fused loads a whole frame before it stores any of it
    // n = 2^(log_local+1): whole inverse (both resident halves + the last stage) in one launch, or null
    hipError_t (*launch_inv_pair)(const plan_view&, const uint64_t*, const uint64_t*, uint64_t*, const frame_layout&, hipStream_t);
    // whole-frame inverse (log_split = 0) by a resident grid walking over the frames, or null
    hipError_t (*launch_inv_loop)(const plan_view&, const uint64_t*, const uint64_t*, uint64_t*, const frame_layout&, hipStream_t) = nullptr;
    bool mul_parked = false;   // launch_mul keeps one frame in registers (the other parked in c's frame): legal at every log_local
    bool whole_only = false;   // the kernels assume the whole frame is resident (log_split = 0): never serves n = 2^(log_local + k)
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
// groups that only exist in lib/libagxntt_diag.so: earlier generations and measured-and-rejected shapes, kept selectable for A/B runs
rb_span rb_entries_gen1();
rb_span rb_entries_n1024();      // the R = 3 kernels of n = 1024 / 2048 / 8192 (round-2 defaults, superseded by the streamed single-frame kernels)
rb_span rb_entries_n2048();
rb_span rb_entries_n8192();
rb_span rb_entries_n4096_ab();
rb_span rb_entries_n8192_split();
rb_span rb_entries_n8192_pair();
rb_span rb_entries_n16384();
rb_span rb_entries_diag();                                      // trace twin, streaming A/B kernels, timing ablations
hipError_t regblock_set_trace(uint64_t* buf, uint64_t waves);   // where the trace kernels write
#endif

}  // namespace agx
