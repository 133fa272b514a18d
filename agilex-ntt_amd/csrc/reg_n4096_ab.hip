// reg_n4096_ab.hip -- one group of the kernel registry (rb_registry.hpp); ids are stable handles for tests
// and A/B runs (AGX_VARIANT_REGBLOCK_BASE + id), not indices.
#define AGX_TU tu_n4096_ab
#include "rb_kernels.hpp"

namespace agx {
namespace AGX_TU {
// n = 4096 second generation as it was tuned step by step (A/B history, each still selectable and tested):
// 12/13 XOR-swizzled image + sign-mask csub; 28/27 padded image (+ select csub); 39 16q-lazy; 50 + look-ahead twiddles
const rb_entry kEntries[] = {
    make_entry2<12, 3, 1, 1, 8>(12),
    make_entry2<12, 3, 1, 0, 8>(13),
    make_entry2<12, 3, 1, 0 | (kOptPad << 1), 8>(28),
    make_entry2<12, 3, 1, 1 | ((kOptPad | kOptSelect) << 1), 8>(27),
    make_entry2<12, 3, 1, 1 | ((kOptPad | kOptSelect | kOptLazy16) << 1), 8>(39),
    make_entry2<12, 3, 1, 1 | ((kOptPad | kOptSelect | kOptLazy16 | kOptTwAhead) << 1), 8>(50),
};
}  // namespace AGX_TU

rb_span rb_entries_n4096_ab() { return rb_span{AGX_TU::kEntries, sizeof(AGX_TU::kEntries) / sizeof(AGX_TU::kEntries[0])}; }

}  // namespace agx
