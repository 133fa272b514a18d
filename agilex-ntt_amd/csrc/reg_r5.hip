// reg_r5.hip -- one group of the kernel registry (rb_registry.hpp); ids are stable handles for tests and A/B runs
// (AGX_VARIANT_REGBLOCK_BASE + id), not indices.
//
// The streamed single-frame kernels (rb_kernels.hpp: kOptStreamTw, kOptPinBf, kOptSplitWord): every thread keeps 32 coefficients
// (R = 5; 16 at n = 1024), three passes, exchanges through the split-word image (4n bytes of LDS), twiddles streamed two table
// entries at a time, butterflies pinned in program order -- 108-126 VGPRs and no scratch, where the same arithmetic scheduled freely
// wants 205.  They are the defaults of every size but n = 4096:
//   n = 1024 (R = 4) / 2048: ONE WAVE per frame -- no workgroup barrier anywhere in the transform;  +5 / +10 % forward, +11 / +8 % inverse
//                            over the 128- / 256-thread R = 3 kernels at 8 waves/SIMD (the fewer, fatter waves win: 5 / 4 waves per SIMD)
//   n = 8192:  256 threads per frame, four workgroups per CU;                                         +7 % forward, +5 % inverse, +7 % product
//   n = 16384: 512 threads, 68 KiB image, two workgroups per CU;                                      forward 37 -> 41 % of 8 TB/s, product 16.9 -> 19.6 %
//   n = 32768: 1024 threads, 136 KiB image, the WHOLE frame resident (16n bytes of traffic);          forward 27-29 -> 34-37 %, inverse 30 -> 32-36 %,
//                                                                                                      product in ONE launch 11.8 -> 16.6 % of its 24n bytes
//   (n = 4096 as 128 threads x 32 coefficients, id 127: +1 % over the R = 3 default at the same 1400 W -- not adopted)
// One frame in registers at a time: the fused product parks NTT(first) thread-privately in c's frame (polymul_rb2_park); n = 1024,
// where two frames of 16 coefficients fit, keeps both in registers (polymul_rb2).
#define AGX_TU tu_r5
#include "rb_kernels.hpp"

namespace agx {
namespace AGX_TU {
constexpr int kLazy = 1 | ((kOptPad | kOptSelect | kOptLazy16 | kOptLazyInv | kOptNtLoad | kOptNtStore | kOptEstReduce | kOptSplitWord | kOptStreamTw | kOptPinBf) << 1);   // q <= 2^60
constexpr int kFast = 1 | ((kOptPad | kOptSelect | kOptNtLoad | kOptNtStore | kOptSplitWord | kOptStreamTw | kOptPinBf) << 1);                                            // q <= 2^61
constexpr int kExact = 0 | ((kOptPad | kOptNtLoad | kOptNtStore | kOptSplitWord | kOptStreamTw | kOptPinBf) << 1);                                                        // q < 2^62, reference op sequence
const rb_entry kEntries[] = {
    // n = 1024: one wave per frame, R = 4, five waves per SIMD; product with both frames in registers
    make_entry_single_mul2<10, 4, kLazy, 5, 4>(150),
    make_entry_single_mul2<10, 4, kFast, 5, 4>(151),
    make_entry_single_mul2<10, 4, kExact, 5, 4>(152),
    // n = 2048: one wave per frame, R = 5
    make_entry_single<11, 5, kLazy, 4>(153),
    make_entry_single<11, 5, kFast, 4>(154),
    make_entry_single<11, 5, kExact, 4>(155),
    // n = 8192
    make_entry_single<13, 5, kLazy, 4>(156),
    make_entry_single<13, 5, kFast, 4>(157),
    make_entry_single<13, 5, kExact, 4>(158),
    // n = 4096, FORWARD ONLY: 128 threads x 32 coefficients, frame loads at raised priority; the forward companion of the R = 3 default (id 93):
    // 59.5 vs 58.2 M NTT/s, 18.5 vs 18.9 uJ per NTT at the same 1400 W (profiles/r03_energy_ab.txt); its inverse (-1 %) and parked
    // product (-4 %) lose to id 93's, so only the forward kernel ships (A/B twin with all three transforms: id 147)
    make_entry_single_fwd<12, 5, kLazy | (kOptPrio << 1), 4>(159),
    // n = 16384 / 32768: forward one workgroup per frame, inverse by the ticket-drawing loop kernel (+3 % / +9 % at 8,192 frames of n = 32768)
    make_entry_single_dloop<14, 5, kLazy, 4, false, true>(117),
    make_entry_single_dloop<15, 5, kLazy, 4, false, true>(119),
    make_entry_single<14, 5, kFast, 4>(120),
    make_entry_single<15, 5, kFast, 4>(121),
    make_entry_single<14, 5, kExact, 4>(122),
    make_entry_single<15, 5, kExact, 4>(123),
#ifdef AGX_DIAG
    // A/B: inverse one workgroup per frame too (114 / 115); forward by the loop kernel as well (116 / 118: 24-52 B of scratch, -3..-6 %)
    make_entry_single<15, 5, kLazy, 4>(114),
    make_entry_single<14, 5, kLazy, 4>(115),
    make_entry_single_dloop<15, 5, kLazy, 4, true, true>(116),
    make_entry_single_dloop<14, 5, kLazy, 4, true, true>(118),
    // A/B: frame loads issued at raised wave priority (what gave the 32-bit kernels +3..8 %): nothing here (VALU-, not HBM-bound)
    make_entry_single<14, 5, kLazy | (kOptPrio << 1), 4>(124),
    make_entry_single<15, 5, kLazy | (kOptPrio << 1), 4>(125),
    // A/B: n = 4096 as 128 threads x 32 coefficients (+1 % over id 93); n = 1024 with the parked product (129) and at six waves per SIMD (149)
    make_entry_single<12, 5, kLazy, 4>(127),
    make_entry_single<12, 5, kLazy | (kOptPrio << 1), 4>(147),
    make_entry_single<12, 5, kLazy | ((kOptPrio | kOptPrioBarrier) << 1), 4>(148),
    make_entry_single<10, 4, kLazy, 5>(129),
    make_entry_single<10, 4, kLazy, 6>(149),
#endif
};
}  // namespace AGX_TU

rb_span rb_entries_r5() { return rb_span{AGX_TU::kEntries, sizeof(AGX_TU::kEntries) / sizeof(AGX_TU::kEntries[0])}; }

}  // namespace agx
