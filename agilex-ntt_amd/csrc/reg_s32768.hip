// reg_s32768.hip -- one size of the streamed single-frame kernels (the family is described at the top of reg_s1024.hip); a group of the kernel
// registry (rb_registry.hpp): ids are stable handles for tests and A/B runs (AGX_VARIANT_REGBLOCK_BASE + id), not indices.
#define AGX_TU tu_s32768
#include "rb_kernels.hpp"
#include "rb_stream_opts.hpp"

namespace agx {
namespace AGX_TU {
const rb_entry kEntries[] = {
    // n = 32768: 1024 threads, the whole frame resident (136 KiB image); inverse by the ticket-drawing loop kernel (+9 % at 8,192 frames)
    make_entry_single_invloop<15, 5, kLazy, 4>(119),
    make_entry_single<15, 5, kFast, 4>(121),
    make_entry_single<15, 5, kExact, 4>(123),
#ifdef AGX_DIAG
    make_entry_single<15, 5, kLazy, 4>(114),      // A/B twin: the inverse one workgroup per frame too
    // (512 threads x 64 coefficients -- R = 6, 205 VGPRs, 2 waves/SIMD, 8-wave barriers -- measured in round 4 and deleted: 31.1 vs 35.6 % of 8 TB/s,
    // profiles/r04_sweep_32768_r6.txt)
#endif
};
}  // namespace AGX_TU

rb_span rb_entries_s32768() { return rb_span{AGX_TU::kEntries, sizeof(AGX_TU::kEntries) / sizeof(AGX_TU::kEntries[0])}; }

}  // namespace agx
