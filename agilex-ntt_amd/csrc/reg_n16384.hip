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
    // 43 (default): one workgroup per frame forward; inverse by the loop kernel (+9 % at 32,768 frames: its long direct-store
    // tail overlaps the next frame's loads); 57: forward by the loop kernel too (A/B: -4 %, static partitioning costs more
    // than the overlap wins); 58: neither (the round-1 default)
    make_entry_loop<14, 4, 1 | ((kOptPad | kOptSelect | kOptLazy16 | kOptLazyInv | kOptTwAhead | kOptNtLoad | kOptNtStore) << 1), 4, false>(43),
    make_entry2_invpair<14, 4, 1 | ((kOptPad | kOptSelect | kOptLazy16 | kOptLazyInv | kOptTwAhead | kOptNtLoad | kOptNtStore) << 1), 4>(58),
    make_entry_loop<14, 4, 1 | ((kOptPad | kOptSelect | kOptLazy16 | kOptLazyInv | kOptTwAhead | kOptNtLoad | kOptNtStore) << 1), 4>(57),   // resident grid walking over the frames
    make_entry_pair<14, 4, 1 | ((kOptPad | kOptSelect | kOptLazy16 | kOptNtLoad | kOptNtStore) << 1), 4>(54),
    make_entry_pair<14, 4, 1 | ((kOptPad | kOptSelect) << 1), 4>(55),
    make_entry_pair<14, 4, 0 | (kOptPad << 1), 4>(56),
};
}  // namespace AGX_TU

rb_span rb_entries_n16384() { return rb_span{AGX_TU::kEntries, sizeof(AGX_TU::kEntries) / sizeof(AGX_TU::kEntries[0])}; }

}  // namespace agx
