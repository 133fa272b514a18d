// reg_n1024.hip -- one group of the kernel registry (rb_registry.hpp); ids are stable handles for tests
// and A/B runs (AGX_VARIANT_REGBLOCK_BASE + id), not indices.
#define AGX_TU tu_n1024
#include "rb_kernels.hpp"

namespace agx {
namespace AGX_TU {
// n = 1024: four frames per 512-thread workgroup
const rb_entry kEntries[] = {
    make_entry2<10, 3, 4, 0 | (kOptPad << 1), 8>(30),
    make_entry2<10, 3, 4, 1 | ((kOptPad | kOptSelect) << 1), 8>(29),
    make_entry2<10, 3, 1, 1 | ((kOptPad | kOptSelect | kOptLazy16 | kOptTwAhead | kOptLazyInv | kOptNtLoad | kOptNtStore) << 1), 8>(63),   // A/B: one frame per 128-thread workgroup
    make_entry2<10, 3, 4, 1 | ((kOptPad | kOptSelect | kOptLazy16) << 1), 8>(40),
    make_entry2<10, 3, 4, 1 | ((kOptPad | kOptSelect | kOptLazy16 | kOptTwAhead | kOptLazyInv | kOptNtLoad | kOptNtStore) << 1), 8>(61),
    make_entry2<10, 3, 2, 1 | ((kOptPad | kOptSelect | kOptLazy16 | kOptTwAhead | kOptLazyInv | kOptNtLoad | kOptNtStore) << 1), 8>(62),   // A/B: two frames per 256-thread workgroup
};
}  // namespace AGX_TU

rb_span rb_entries_n1024() { return rb_span{AGX_TU::kEntries, sizeof(AGX_TU::kEntries) / sizeof(AGX_TU::kEntries[0])}; }

}  // namespace agx
