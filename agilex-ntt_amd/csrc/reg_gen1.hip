// reg_gen1.hip -- one group of the kernel registry (rb_registry.hpp); ids are stable handles for tests
// and A/B runs (AGX_VARIANT_REGBLOCK_BASE + id), not indices.
#define AGX_TU tu_gen1
#include "rb_kernels.hpp"

namespace agx {
namespace AGX_TU {
// first generation (portable butterfly, exact arithmetic, R = 4): the measured starting point at n = 4096
const rb_entry kEntries[] = {
    make_entry<12, 4, 1, false, 1>(2),
};
}  // namespace AGX_TU

rb_span rb_entries_gen1() { return rb_span{AGX_TU::kEntries, sizeof(AGX_TU::kEntries) / sizeof(AGX_TU::kEntries[0])}; }

}  // namespace agx
