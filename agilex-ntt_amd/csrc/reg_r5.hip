// reg_r5.hip -- one group of the kernel registry (rb_registry.hpp); ids are stable handles for tests and A/B runs
// (AGX_VARIANT_REGBLOCK_BASE + id), not indices.
//
// Whole-frame kernels for n = 16384 (512 threads) and n = 32768 (1024 threads): R = 5, every thread keeps 32 coefficients,
// three passes (5 + 5 + 4 / 5 + 5 + 5 stages), exchanges through the split-word image (4n bytes of LDS: 68 KiB -> two
// workgroups per CU at n = 16384, 136 KiB at n = 32768), twiddles streamed two table entries at a time, butterflies pinned in
// program order (rb_kernels.hpp: kOptStreamTw, kOptPinBf) -- together 108-121 VGPRs and no scratch, where the same arithmetic
// scheduled freely wants 205.  One frame in registers at a time: the fused product parks NTT(first) thread-privately in c's frame.
// Measured against the kernels they replace (profiles/r03_*): n = 32768 forward 34-37 % of 8 TB/s (pair / fused-split kernels:
// 29 / 27 %), inverse 32-36 % (30 %), product in ONE launch 16.6 % of its 24n bytes (three launches: 11.8 %); n = 16384 forward
// 41 % (R = 4, 1024 threads: 37 %), product 19.6 % (16.9 %).
#define AGX_TU tu_r5
#include "rb_kernels.hpp"

namespace agx {
namespace AGX_TU {
constexpr int kLazy = 1 | ((kOptPad | kOptSelect | kOptLazy16 | kOptLazyInv | kOptNtLoad | kOptNtStore | kOptEstReduce | kOptSplitWord | kOptStreamTw | kOptPinBf) << 1);   // q <= 2^60
constexpr int kFast = 1 | ((kOptPad | kOptSelect | kOptNtLoad | kOptNtStore | kOptSplitWord | kOptStreamTw | kOptPinBf) << 1);                                            // q <= 2^61
constexpr int kExact = 0 | ((kOptPad | kOptNtLoad | kOptNtStore | kOptSplitWord | kOptStreamTw | kOptPinBf) << 1);                                                        // q < 2^62, reference op sequence
const rb_entry kEntries[] = {
    // defaults: forward one workgroup per frame, inverse by the ticket-drawing loop kernel (+3 % at n = 16384, +9 % at 8,192 frames of n = 32768)
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
#endif
};
}  // namespace AGX_TU

rb_span rb_entries_r5() { return rb_span{AGX_TU::kEntries, sizeof(AGX_TU::kEntries) / sizeof(AGX_TU::kEntries[0])}; }

}  // namespace agx
