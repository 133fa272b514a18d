// reg_n2048.hip -- one group of the kernel registry (rb_registry.hpp); ids are stable handles for tests
// and A/B runs (AGX_VARIANT_REGBLOCK_BASE + id), not indices.
#define AGX_TU tu_n2048
#include "rb_kernels.hpp"

namespace agx {
namespace AGX_TU {
// n = 2048: two frames per 512-thread workgroup
const rb_entry kEntries[] = {
    make_entry2<11, 3, 2, 0 | (kOptPad << 1), 8>(32),
    make_entry2<11, 3, 2, 1 | ((kOptPad | kOptSelect) << 1), 8>(31),
    make_entry2<11, 3, 1, 1 | ((kOptPad | kOptSelect | kOptLazy16 | kOptLazyInv | kOptNtLoad | kOptNtStore) << 1), 8>(59),   // A/B: one frame per 256-thread workgroup (8 workgroups per CU)
    // A/B at n = 4096: one 256-thread workgroup per frame, its two 2048-halves in turn (4-wave barrier, 17 KiB image)
    make_entry_pair<11, 3, 1 | ((kOptPad | kOptSelect | kOptLazy16 | kOptTwAhead | kOptNtLoad | kOptNtStore) << 1), 6>(94),
    make_entry_pair<11, 3, 1 | ((kOptPad | kOptSelect | kOptLazy16 | kOptTwAhead | kOptNtLoad | kOptNtStore) << 1), 8>(95),
    make_entry_pair<11, 3, 1 | ((kOptPad | kOptSelect | kOptLazy16 | kOptNtLoad | kOptNtStore) << 1), 8>(96),
    // A/B at n = 4096, out of place only: two 256-thread workgroups per frame, each recomputing the leading stage for its 2048-half
    make_entry_split<11, 3, 1, 1 | ((kOptPad | kOptSelect | kOptLazy16 | kOptTwAhead | kOptNtLoad | kOptNtStore) << 1), 8, 1>(97),
    make_entry_split<11, 3, 1, 1 | ((kOptPad | kOptSelect | kOptLazy16 | kOptTwAhead | kOptNtStore) << 1), 8, 1>(98),
    make_entry2<11, 3, 2, 1 | ((kOptPad | kOptSelect | kOptLazy16 | kOptLazyInv | kOptNtLoad | kOptNtStore) << 1), 8>(41),
};
}  // namespace AGX_TU

rb_span rb_entries_n2048() { return rb_span{AGX_TU::kEntries, sizeof(AGX_TU::kEntries) / sizeof(AGX_TU::kEntries[0])}; }

}  // namespace agx
