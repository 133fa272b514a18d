// reg_s4096.hip -- one size of the streamed single-frame kernels (the family is described at the top of reg_s1024.hip); a group of the kernel
// registry (rb_registry.hpp): ids are stable handles for tests and A/B runs (AGX_VARIANT_REGBLOCK_BASE + id), not indices.
#define AGX_TU tu_s4096
#include "rb_kernels.hpp"
#include "rb_stream_opts.hpp"

namespace agx {
namespace AGX_TU {
const rb_entry kEntries[] = {
    // n = 4096, FORWARD ONLY: 128 threads x 32 coefficients, frame loads at raised priority; the forward companion of the R = 3 default (id 93):
    // 59.5 vs 58.2 M NTT/s, 18.5 vs 18.9 uJ per NTT at the same 1400 W (profiles/r03_energy_ab.txt); its inverse (-1 %) and parked
    // product (-4 %) lose to id 93's, so only the forward kernel ships (A/B twin with all three transforms: id 147)
    make_entry_single_fwd<12, 5, kLazy | (kOptPrio << 1), 4>(159),
#ifdef AGX_DIAG
    // A/B twins kept in lib/libagxntt_diag.so: all three transforms in this shape with the load priority (147: its inverse -1 %, its parked
    // product -4 % against id 93's), and R = 4 streamed one table entry at a time at 8 waves/SIMD (161: equal to the default within 1 %)
    make_entry_single<12, 5, kLazy | (kOptPrio << 1), 4>(147),
    make_entry_single<12, 4, kLazy | (kOptStreamCh1 << 1), 8>(161),
#endif
};
}  // namespace AGX_TU

rb_span rb_entries_s4096() { return rb_span{AGX_TU::kEntries, sizeof(AGX_TU::kEntries) / sizeof(AGX_TU::kEntries[0])}; }

}  // namespace agx
