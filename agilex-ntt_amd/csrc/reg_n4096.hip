// reg_n4096.hip -- one group of the kernel registry (rb_registry.hpp); ids are stable handles for tests
// and A/B runs (AGX_VARIANT_REGBLOCK_BASE + id), not indices.
#define AGX_TU tu_n4096
#include "rb_kernels.hpp"

namespace agx {
namespace AGX_TU {
// n = 4096 defaults: wave priority raised from launch until the frame's one all-wave barrier has been passed (+2 %);
// 93 = 16q-lazy with the tail-free subtract schedule and quotient-estimate final reduction (q <= 2^60), 92 = fast (q <= 2^61), 91 = exact (q < 2^62)
const rb_entry kEntries[] = {
    with_fwd_companion(make_entry2<12, 3, 1, 1 | ((kOptPad | kOptSelect | kOptLazy16 | kOptTwAhead | kOptPrio | kOptPrioBarrier | kOptScalarBase | kOptLazyInv | kOptTwAheadInv | kOptNtLoad | kOptNtStore | kOptEstReduce) << 1), 8>(93), 159, 4096),      // forward launches of >= 4,096 frames go to the streamed 128-thread kernel of reg_s4096.hip (+2.3 %); smaller ones stay here (512 threads per frame: 12.6 vs 18.7 us for BASELINE configs[1]'s batch-1 forward + inverse pair)   // 90 + tail-free subtract schedule and quotient-estimate final reduction
    make_entry2<12, 3, 1, 0 | ((kOptPad | kOptPrio | kOptPrioBarrier | kOptScalarBase) << 1), 8>(91),
    make_entry2<12, 3, 1, 1 | ((kOptPad | kOptSelect | kOptTwAhead | kOptPrio | kOptPrioBarrier | kOptScalarBase) << 1), 8>(92),
#ifdef AGX_DIAG
    // A/B entries (lib/libagxntt_diag.so only): no default and no call-shape selector reaches them
    make_entry2<12, 3, 1, 1 | ((kOptPad | kOptSelect | kOptLazy16 | kOptTwAhead | kOptPrio | kOptPrioBarrier | kOptScalarBase | kOptLazyInv | kOptTwAheadInv | kOptNtLoad | kOptNtStore) << 1), 8>(90),
    make_entry2<12, 3, 1, 1 | ((kOptPad | kOptSelect | kOptLazy16 | kOptTwAhead | kOptPrio | kOptPrioBarrier | kOptScalarBase | kOptLazyInv | kOptTwAheadInv | kOptNtLoad | kOptNtStore | kOptEstReduce | kOptMulLoCross) << 1), 8>(87),     // A/B: 93 with the cross products as 32-bit multiplies (-3 % energy per butterfly in tools/microbench pwr)
    make_entry2<12, 4, 1, 1 | ((kOptPad | kOptSelect | kOptLazy16 | kOptTwAhead | kOptPrio | kOptPrioBarrier | kOptScalarBase) << 1), 4>(66),   // A/B: R = 4 (three passes, 4-wave workgroups) at 4 waves/SIMD, within 1.5 % of id 90
    // A/B (forward only): R = 4, 256-thread workgroups, split-word exchanges through a 17 KiB image -> up to 8 workgroups per CU;
    // 89 / 86: register budget for 5 / 6 workgroups per CU (no look-ahead twiddle arrays: at R = 4 they alone are 64 VGPRs); measured 0.281 / 0.288 ms
    // against 0.274 for the default and 0.284 for id 66 (full image, 4 per CU); 7 / 8 per CU spill 64 / 100 B per lane: 0.317 / 0.351 ms
    make_entry_fwd_only<12, 4, 1, 1 | ((kOptPad | kOptSelect | kOptLazy16 | kOptScalarBase | kOptNtLoad | kOptNtStore | kOptEstReduce | kOptSplitWord) << 1), 6>(86),
    make_entry_fwd_only<12, 4, 1, 1 | ((kOptPad | kOptSelect | kOptLazy16 | kOptScalarBase | kOptNtLoad | kOptNtStore | kOptEstReduce | kOptSplitWord) << 1), 5>(89),
    // A/B (inverse): 93 with the inverse's first-pass twiddles requested ahead of the frame's LDS staging (101: first stage's, 102: all seven)
    make_entry2<12, 3, 1, 1 | ((kOptPad | kOptSelect | kOptLazy16 | kOptTwAhead | kOptPrio | kOptPrioBarrier | kOptScalarBase | kOptLazyInv | kOptTwAheadInv | kOptNtLoad | kOptNtStore | kOptEstReduce | kOptInvTwFirst) << 1), 8>(101),
    make_entry2<12, 3, 1, 1 | ((kOptPad | kOptSelect | kOptLazy16 | kOptTwAhead | kOptPrio | kOptPrioBarrier | kOptScalarBase | kOptLazyInv | kOptTwAheadInv | kOptNtLoad | kOptNtStore | kOptEstReduce | kOptInvTwFirst | kOptInvTwFirstAll) << 1), 8>(102),
    // A/B (inverse): resident grid drawing frames from a ticket counter (the n = 16384 inverse's form) / fixed stride
    make_entry_dloop<12, 3, 1 | ((kOptPad | kOptSelect | kOptLazy16 | kOptTwAhead | kOptPrio | kOptPrioBarrier | kOptScalarBase | kOptLazyInv | kOptTwAheadInv | kOptNtLoad | kOptNtStore | kOptEstReduce) << 1), 8, false, true>(106),
    make_entry_loop<12, 3, 1 | ((kOptPad | kOptSelect | kOptLazy16 | kOptTwAhead | kOptPrio | kOptPrioBarrier | kOptScalarBase | kOptLazyInv | kOptTwAheadInv | kOptNtLoad | kOptNtStore | kOptEstReduce) << 1), 8, false>(107),
    // A/B (inverse): priority policies
    make_entry2<12, 3, 1, 1 | ((kOptPad | kOptSelect | kOptLazy16 | kOptTwAhead | kOptPrio | kOptPrioBarrier | kOptScalarBase | kOptLazyInv | kOptTwAheadInv | kOptNtLoad | kOptNtStore | kOptEstReduce | kOptInvPrioTail) << 1), 8>(103),
    make_entry2<12, 3, 1, 1 | ((kOptPad | kOptSelect | kOptLazy16 | kOptTwAhead | kOptPrio | kOptPrioBarrier | kOptScalarBase | kOptLazyInv | kOptTwAheadInv | kOptNtLoad | kOptNtStore | kOptEstReduce | kOptInvPrioAsc) << 1), 8>(104),
    make_entry2<12, 3, 1, 1 | ((kOptPad | kOptSelect | kOptLazy16 | kOptTwAhead | kOptPrio | kOptPrioBarrier | kOptScalarBase | kOptLazyInv | kOptTwAheadInv | kOptNtLoad | kOptNtStore | kOptEstReduce | kOptInvPrioDesc) << 1), 8>(105),
#endif
};
}  // namespace AGX_TU

rb_span rb_entries_n4096() { return rb_span{AGX_TU::kEntries, sizeof(AGX_TU::kEntries) / sizeof(AGX_TU::kEntries[0])}; }

}  // namespace agx
