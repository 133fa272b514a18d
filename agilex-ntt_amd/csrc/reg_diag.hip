// reg_diag.hip -- diagnostics and A/B kernels, built only into lib/libagxntt_diag.so (make diag, -DAGX_DIAG):
// the trace twin of the default n=4096 kernel (70: s_memtime stamps at 12 phase boundaries, tools/timeline.py),
// the streaming kernels (83/84: resident workgroups drawing frames from a ticket counter; their ticket pair belongs
// to the plan, so ONE stream per plan), and with -DAGX_TIMING_ABLATIONS the timing-only ablations (67-72, WRONG results).
// None of this is in the product library.
#define AGX_TU tu_diag
#include "rb_kernels.hpp"

namespace agx {
namespace AGX_TU {
const rb_entry kEntries[] = {
    make_entry2<12, 3, 1, 1 | ((kOptPad | kOptSelect | kOptLazy16 | kOptTwAhead | kOptPrio | kOptPrioBarrier | kOptScalarBase | kOptLazyInv | kOptTwAheadInv | kOptNtLoad | kOptNtStore | kOptTrace) << 1), 8>(70),   // diagnostics only: id 90 + stamps (forward and inverse)
    make_entry2<12, 3, 1, 1 | ((kOptPad | kOptSelect | kOptLazy16 | kOptTwAhead | kOptPrio | kOptPrioBarrier | kOptScalarBase | kOptLazyInv | kOptTwAheadInv | kOptNtLoad | kOptNtStore | kOptTrace | kOptInvTwFirst) << 1), 8>(73),   // + the inverse's first-stage twiddles ahead of the staging
    make_entry2<12, 3, 1, 1 | ((kOptPad | kOptSelect | kOptLazy16 | kOptTwAhead | kOptPrio | kOptPrioBarrier | kOptScalarBase | kOptLazyInv | kOptTwAheadInv | kOptNtLoad | kOptNtStore | kOptTrace | kOptInvTwFirst | kOptInvTwFirstAll) << 1), 8>(74),
    make_entry_stream<12, 3, 1 | ((kOptPad | kOptSelect | kOptLazy16 | kOptTwAhead | kOptNtLoad | kOptNtStore) << 1), 6>(83),   // A/B only: one stream per plan
    make_entry_stream<12, 3, 1 | ((kOptPad | kOptSelect | kOptLazy16 | kOptTwAhead | kOptNtLoad | kOptNtStore) << 1), 8, 1>(84),
#ifdef AGX_TIMING_ABLATIONS
    // timing only, WRONG RESULTS (make EXTRA=-DAGX_TIMING_ABLATIONS): the default kernel without per-lane twiddle
    // traffic (67), with L2-resident frames (68: loads and stores, 71: loads only, 72: stores only), with neither (69)
    make_entry2<12, 3, 1, 1 | ((kOptPad | kOptSelect | kOptLazy16 | kOptTwAhead | kOptPrio | kOptPrioBarrier | kOptScalarBase | kOptLazyInv | kOptTwAheadInv | kOptAblateTw) << 1), 8>(67),
    make_entry2<12, 3, 1, 1 | ((kOptPad | kOptSelect | kOptLazy16 | kOptTwAhead | kOptPrio | kOptPrioBarrier | kOptScalarBase | kOptLazyInv | kOptTwAheadInv | kOptAblateHbm) << 1), 8>(68),
    make_entry2<12, 3, 1, 1 | ((kOptPad | kOptSelect | kOptLazy16 | kOptTwAhead | kOptPrio | kOptPrioBarrier | kOptScalarBase | kOptLazyInv | kOptTwAheadInv | kOptAblateTw | kOptAblateHbm) << 1), 8>(69),
    make_entry2<12, 3, 1, 1 | ((kOptPad | kOptSelect | kOptLazy16 | kOptTwAhead | kOptPrio | kOptPrioBarrier | kOptScalarBase | kOptLazyInv | kOptTwAheadInv | kOptAblateHbm | kOptAblateLdOnly) << 1), 8>(71),
    make_entry2<12, 3, 1, 1 | ((kOptPad | kOptSelect | kOptLazy16 | kOptTwAhead | kOptPrio | kOptPrioBarrier | kOptScalarBase | kOptLazyInv | kOptTwAheadInv | kOptAblateHbm | kOptAblateStOnly) << 1), 8>(72),
#endif
};
}  // namespace AGX_TU

rb_span rb_entries_diag() { return rb_span{AGX_TU::kEntries, sizeof(AGX_TU::kEntries) / sizeof(AGX_TU::kEntries[0])}; }

hipError_t regblock_set_trace(uint64_t* buf, uint64_t waves) {
    hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(AGX_TU::g_trace_buf), &buf, sizeof(buf));
    if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(AGX_TU::g_trace_waves), &waves, sizeof(waves));
    return e;
}

}  // namespace agx
