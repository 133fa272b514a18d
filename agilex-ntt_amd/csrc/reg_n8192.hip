// reg_n8192.hip -- one group of the kernel registry (rb_registry.hpp); ids are stable handles for tests
// and A/B runs (AGX_VARIANT_REGBLOCK_BASE + id), not indices.
#define AGX_TU tu_n8192
#include "rb_kernels.hpp"

namespace agx {
namespace AGX_TU {
// n = 8192: one frame per 1024-thread workgroup, 8 waves/SIMD
const rb_entry kEntries[] = {
    make_entry2<13, 3, 1, 0 | (kOptPad << 1), 8>(34),
    make_entry2<13, 3, 1, 1 | ((kOptPad | kOptSelect) << 1), 8>(33),
    make_entry2<13, 3, 1, 1 | ((kOptPad | kOptSelect | kOptLazy16 | kOptTwAhead | kOptLazyInv | kOptNtLoad | kOptNtStore) << 1), 8>(64),
    make_entry2<13, 3, 1, 1 | ((kOptPad | kOptSelect | kOptLazy16) << 1), 8>(42),
    make_entry2<13, 4, 1, 1 | ((kOptPad | kOptSelect | kOptLazy16 | kOptTwAhead | kOptLazyInv | kOptNtLoad | kOptNtStore) << 1), 4>(65),   // A/B: R = 4, 512-thread workgroups (8 waves), 4 waves/SIMD
};
}  // namespace AGX_TU

rb_span rb_entries_n8192() { return rb_span{AGX_TU::kEntries, sizeof(AGX_TU::kEntries) / sizeof(AGX_TU::kEntries[0])}; }

}  // namespace agx
