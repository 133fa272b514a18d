// reg_diag.hip -- lib/libagxntt_diag.so only (make diag, -DAGX_DIAG): the trace twin of the n = 4096 default.  tools/timeline.py
// selects it (AGX_VARIANT_REGBLOCK_BASE + 70) and reads the s_memtime stamps every wave leaves at twelve phase boundaries
// (agx_ntt_debug_set_trace_buffer, tools/agx_ntt_diag.h); forward and inverse.  The streaming / ablation / priority-policy A/B kernels
// that used to live here were measured, recorded (profiles/r01g_*, r01j_*, r03_*) and removed in round 4.
#define AGX_TU tu_diag
#include "rb_kernels.hpp"

namespace agx {
namespace AGX_TU {
const rb_entry kEntries[] = {
    make_entry2<12, 3, 1, 1 | ((kOptPad | kOptSelect | kOptLazy16 | kOptTwAhead | kOptPrio | kOptPrioBarrier | kOptScalarBase | kOptLazyInv | kOptTwAheadInv | kOptNtLoad | kOptNtStore | kOptTrace) << 1), 8>(70),
};
}  // namespace AGX_TU

rb_span rb_entries_diag() { return rb_span{AGX_TU::kEntries, sizeof(AGX_TU::kEntries) / sizeof(AGX_TU::kEntries[0])}; }

hipError_t regblock_set_trace(uint64_t* buf, uint64_t waves) {
    hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(AGX_TU::g_trace_buf), &buf, sizeof(buf));
    if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(AGX_TU::g_trace_waves), &waves, sizeof(waves));
    return e;
}

}  // namespace agx
