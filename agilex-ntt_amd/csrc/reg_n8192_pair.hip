// reg_n8192_pair.hip -- one group of the kernel registry (rb_registry.hpp); ids are stable handles for tests
// and A/B runs (AGX_VARIANT_REGBLOCK_BASE + id), not indices.
#define AGX_TU tu_n8192_pair
#include "rb_kernels.hpp"

namespace agx {
namespace AGX_TU {
// n = 16384: one workgroup per frame, its two 8192-halves in turn (in-place safe, no redundant work)
const rb_entry kEntries[] = {
    make_entry_pair<13, 3, 0 | (kOptPad << 1), 8>(51),
    make_entry_pair<13, 3, 1 | ((kOptPad | kOptSelect) << 1), 8>(52),
    make_entry_pair<13, 3, 1 | ((kOptPad | kOptSelect | kOptLazy16 | kOptNtLoad | kOptNtStore) << 1), 8>(53),
};
}  // namespace AGX_TU

rb_span rb_entries_n8192_pair() { return rb_span{AGX_TU::kEntries, sizeof(AGX_TU::kEntries) / sizeof(AGX_TU::kEntries[0])}; }

}  // namespace agx
