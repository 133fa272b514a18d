// reg_n16384.hip -- one group of the kernel registry (rb_registry.hpp); ids are stable handles for tests
// and A/B runs (AGX_VARIANT_REGBLOCK_BASE + id), not indices.
#define AGX_TU tu_n16384
#include "rb_kernels.hpp"

namespace agx {
namespace AGX_TU {
// n = 16384: R = 4, one 1024-thread workgroup per frame (4 waves/SIMD); inverse at n = 32768 by the one-launch pair kernel;
// 54-56: n = 32768 forward, 1024 threads hold the frame (32 coefficients each), two 16384-halves in turn
const rb_entry kEntries[] = {
    make_entry2_invpair<14, 4, 0 | (kOptPad << 1), 4>(36),
    make_entry2_invpair<14, 4, 1 | ((kOptPad | kOptSelect) << 1), 4>(35),
    // 43 (default): forward one workgroup per frame; inverse by the dynamic loop kernel (a resident grid drawing frames from a
    // ticket counter: its long direct-store tail overlaps the next frame's loads; +10.6 % at config 4's 65,536 frames).
    // A/B: 37 forward by the dynamic loop kernel too (-2.8 %: the hand-over barrier costs the forward more than the overlap wins) and the
    // fused product in ONE launch (polymul_rb2_park: NTT(first) parked in c's frame; no caller scratch): 18.71 vs 18.48 ms for the
    // three-launch product at config 4's slice, 2.48 vs 2.44 ms at 8,192 products -- the product is bound by its three transforms,
    // 57 both by the fixed-stride loop kernels (forward -4 %, inverse +7.6 %), 58 neither (the round-1 default)
    make_entry_dloop<14, 4, 1 | ((kOptPad | kOptSelect | kOptLazy16 | kOptLazyInv | kOptTwAhead | kOptNtLoad | kOptNtStore) << 1), 4, false, true>(43),
    make_entry_dloop<14, 4, 1 | ((kOptPad | kOptSelect | kOptLazy16 | kOptLazyInv | kOptTwAhead | kOptNtLoad | kOptNtStore) << 1), 4, true, true, true>(37),   // + fused product in one launch (polymul_rb2_park)
    make_entry_loop<14, 4, 1 | ((kOptPad | kOptSelect | kOptLazy16 | kOptLazyInv | kOptTwAhead | kOptNtLoad | kOptNtStore) << 1), 4>(57),
    make_entry2_invpair<14, 4, 1 | ((kOptPad | kOptSelect | kOptLazy16 | kOptLazyInv | kOptTwAhead | kOptNtLoad | kOptNtStore) << 1), 4>(58),
    make_entry_pair<14, 4, 1 | ((kOptPad | kOptSelect | kOptLazy16 | kOptNtLoad | kOptNtStore) << 1), 4>(54),
    make_entry_pair<14, 4, 1 | ((kOptPad | kOptSelect) << 1), 4>(55),
    make_entry_pair<14, 4, 0 | (kOptPad << 1), 4>(56),
};
}  // namespace AGX_TU

rb_span rb_entries_n16384() { return rb_span{AGX_TU::kEntries, sizeof(AGX_TU::kEntries) / sizeof(AGX_TU::kEntries[0])}; }

}  // namespace agx
