// reg_s16384.hip -- one size of the streamed single-frame kernels (the family is described at the top of reg_s1024.hip); a group of the kernel
// registry (rb_registry.hpp): ids are stable handles for tests and A/B runs (AGX_VARIANT_REGBLOCK_BASE + id), not indices.
#define AGX_TU tu_s16384
#include "rb_kernels.hpp"
#include "rb_stream_opts.hpp"

namespace agx {
namespace AGX_TU {
const rb_entry kEntries[] = {
    // n = 16384: 512 threads, 68 KiB image, two workgroups per CU; inverse by the ticket-drawing loop kernel (+3 % at 8,192 frames);
    // forward calls go to id 164: R = 4 streamed one table entry at a time, 1024 threads, 63 VGPRs -> 8 waves/SIMD (+2.3 ... +3.8 %)
    with_fwd_companion(make_entry_single_invloop<14, 5, kLazy, 4>(117), 164),
    make_entry_single_fwd<14, 4, kLazy | (kOptStreamCh1 << 1), 8>(164),
    make_entry_single<14, 5, kFast, 4>(120),
    make_entry_single<14, 5, kExact, 4>(122),
#ifdef AGX_DIAG
    // A/B twins: the inverse one workgroup per frame too (115); all three transforms in the R = 4 / 8 waves shape (160)
    make_entry_single<14, 5, kLazy, 4>(115),
    make_entry_single<14, 4, kLazy | (kOptStreamCh1 << 1), 8>(160),
#endif
};
}  // namespace AGX_TU

rb_span rb_entries_s16384() { return rb_span{AGX_TU::kEntries, sizeof(AGX_TU::kEntries) / sizeof(AGX_TU::kEntries[0])}; }

}  // namespace agx
