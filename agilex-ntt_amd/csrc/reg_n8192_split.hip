// reg_n8192_split.hip -- one group of the kernel registry (rb_registry.hpp); ids are stable handles for tests
// and A/B runs (AGX_VARIANT_REGBLOCK_BASE + id), not indices.
#define AGX_TU tu_n8192_split
#include "rb_kernels.hpp"

namespace agx {
namespace AGX_TU {
// n = 16384 / 32768 as 2 / 4 resident blocks of 8192 (8 waves/SIMD) with fused leading stages (out of place only)
const rb_entry kEntries[] = {
    make_entry_split<13, 3, 1, 0 | (kOptPad << 1), 8, 1>(44),
    make_entry_split<13, 3, 1, 1 | ((kOptPad | kOptSelect) << 1), 8, 1>(45),
    make_entry_split<13, 3, 1, 1 | ((kOptPad | kOptSelect | kOptLazy16 | kOptNtLoad | kOptNtStore) << 1), 8, 1>(46),
    make_entry_split<13, 3, 1, 0 | (kOptPad << 1), 8, 2>(47),
    make_entry_split<13, 3, 1, 1 | ((kOptPad | kOptSelect) << 1), 8, 2>(48),
    make_entry_split<13, 3, 1, 1 | ((kOptPad | kOptSelect | kOptLazy16 | kOptNtLoad | kOptNtStore) << 1), 8, 2>(49),
};
}  // namespace AGX_TU

rb_span rb_entries_n8192_split() { return rb_span{AGX_TU::kEntries, sizeof(AGX_TU::kEntries) / sizeof(AGX_TU::kEntries[0])}; }

}  // namespace agx
