// reg_s8192.hip -- one size of the streamed single-frame kernels (the family is described at the top of reg_s1024.hip); a group of the kernel
// registry (rb_registry.hpp): ids are stable handles for tests and A/B runs (AGX_VARIANT_REGBLOCK_BASE + id), not indices.
#define AGX_TU tu_s8192
#include "rb_kernels.hpp"
#include "rb_stream_opts.hpp"

namespace agx {
namespace AGX_TU {
const rb_entry kEntries[] = {
    // n = 8192: 256 threads per frame, four workgroups per CU
    make_entry_single<13, 5, kLazy, 4>(156),
    make_entry_single<13, 5, kFast, 4>(157),
    make_entry_single<13, 5, kExact, 4>(158),
};
}  // namespace AGX_TU

rb_span rb_entries_s8192() { return rb_span{AGX_TU::kEntries, sizeof(AGX_TU::kEntries) / sizeof(AGX_TU::kEntries[0])}; }

}  // namespace agx
