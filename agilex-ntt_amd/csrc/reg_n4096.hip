// reg_n4096.hip -- one group of the kernel registry (rb_registry.hpp); ids are stable handles for tests
// and A/B runs (AGX_VARIANT_REGBLOCK_BASE + id), not indices.
#define AGX_TU tu_n4096
#include "rb_kernels.hpp"

namespace agx {
namespace AGX_TU {
// n = 4096 defaults: R = 3, 512 threads per frame, 8 waves/SIMD, one s_barrier per frame; wave priority raised from launch until that
// barrier has been passed (+2 %); look-ahead twiddle fetches; non-temporal frame loads / stores.
// 93 = 16q-lazy with the tail-free subtract schedule and quotient-estimate final reduction (q <= 2^60), 92 = fast (q <= 2^61), 91 = exact (q < 2^62)
const rb_entry kEntries[] = {
    // forward launches of >= 4,096 frames go to the streamed 128-thread kernel of reg_s4096.hip (id 159: +2.3 %); smaller ones stay here
    // (512 threads per frame: 12.6 vs 18.7 us for BASELINE configs[1]'s batch-1 forward + inverse pair)
    with_fwd_companion(make_entry2<12, 3, 1, 1 | ((kOptPad | kOptSelect | kOptLazy16 | kOptTwAhead | kOptPrio | kOptPrioBarrier | kOptScalarBase | kOptLazyInv | kOptTwAheadInv | kOptNtLoad | kOptNtStore | kOptEstReduce) << 1), 8>(93), 159, 4096),
    make_entry2<12, 3, 1, 0 | ((kOptPad | kOptPrio | kOptPrioBarrier | kOptScalarBase) << 1), 8>(91),
    make_entry2<12, 3, 1, 1 | ((kOptPad | kOptSelect | kOptTwAhead | kOptPrio | kOptPrioBarrier | kOptScalarBase) << 1), 8>(92),
};
}  // namespace AGX_TU

rb_span rb_entries_n4096() { return rb_span{AGX_TU::kEntries, sizeof(AGX_TU::kEntries) / sizeof(AGX_TU::kEntries[0])}; }

}  // namespace agx
