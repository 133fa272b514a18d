// reg_s2048.hip -- one size of the streamed single-frame kernels (the family is described at the top of reg_s1024.hip); a group of the kernel
// registry (rb_registry.hpp): ids are stable handles for tests and A/B runs (AGX_VARIANT_REGBLOCK_BASE + id), not indices.
#define AGX_TU tu_s2048
#include "rb_kernels.hpp"
#include "rb_stream_opts.hpp"

namespace agx {
namespace AGX_TU {
const rb_entry kEntries[] = {
    // n = 2048: one wave per frame, R = 5
    make_entry_single<11, 5, kLazy, 4>(153),
    make_entry_single<11, 5, kFast, 4>(154),
    make_entry_single<11, 5, kExact, 4>(155),
};
}  // namespace AGX_TU

rb_span rb_entries_s2048() { return rb_span{AGX_TU::kEntries, sizeof(AGX_TU::kEntries) / sizeof(AGX_TU::kEntries[0])}; }

}  // namespace agx
