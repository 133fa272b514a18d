// reg_s16384.hip -- one size of the streamed single-frame kernels (the family is described at the top of reg_s1024.hip); a group of the kernel
// registry (rb_registry.hpp): ids are stable handles for tests and A/B runs (AGX_VARIANT_REGBLOCK_BASE + id), not indices.
#define AGX_TU tu_s16384
#include "rb_kernels.hpp"
#include "rb_stream_opts.hpp"

namespace agx {
namespace AGX_TU {
const rb_entry kEntries[] = {
    // n = 16384: 512 threads, two workgroups per CU; forward one workgroup per frame, inverse by the ticket-drawing loop kernel (+3 %)
    with_fwd_companion(make_entry_single_dloop<14, 5, kLazy, 4, false, true>(117), 164),
    // FORWARD ONLY, the forward companion of 117: R = 4 (16 coefficients per thread) with one table entry per chunk fits 63 VGPRs, so two
    // 1024-thread workgroups per CU run at 8 waves/SIMD: +2.3 % forward (42.75 vs 41.79 % of 8 TB/s); its inverse and product only equal 117's
    // (A/B twin with all three transforms: id 160)
    make_entry_single_fwd<14, 4, kLazy | (kOptStreamCh1 << 1), 8>(164),
    make_entry_single<14, 5, kFast, 4>(120),
    make_entry_single<14, 5, kExact, 4>(122),
#ifdef AGX_DIAG
    // A/B: inverse one workgroup per frame too (115); forward by the loop kernel as well (118: 24 B of scratch, -3 %); 124 frame loads at raised priority (nothing)
    make_entry_single<14, 5, kLazy, 4>(115),
    make_entry_single_dloop<14, 5, kLazy, 4, true, true>(118),
    make_entry_single<14, 5, kLazy | (kOptPrio << 1), 4>(124),
    // A/B: forward by the ticket loop with one table entry per chunk (120 VGPRs, no scratch): still -4 % (n = 16384) / -3..-5 % (n = 32768) against one
    // workgroup per frame -- the hand-over barrier and the thinner twiddle prefetch cost more than the overlapped store tail wins
    make_entry_single_dloop<14, 5, kLazy | (kOptStreamCh1 << 1), 4, true, true>(145),
    // A/B: R = 4 (16 coefficients per thread) streamed with one table entry per chunk: 60-64 VGPRs, no scratch -> 8 waves/SIMD (its inverse by the
    // ticket loop needs 60 B of scratch at 64 VGPRs and falls to 37.5 % against 40.4 % for id 117's: not registered)
    make_entry_single<14, 4, kLazy | (kOptStreamCh1 << 1), 8>(160),
#endif
};
}  // namespace AGX_TU

rb_span rb_entries_s16384() { return rb_span{AGX_TU::kEntries, sizeof(AGX_TU::kEntries) / sizeof(AGX_TU::kEntries[0])}; }

}  // namespace agx
