// reg_wp32.hip -- the wave-packed kernels of the small sizes, n = 2 ... 512, 32-bit arithmetic for plans whose every modulus is below
// 2^31 (wp_kernels.hpp over rb32_kernels.hpp); a group of the kernel registry (rb_registry.hpp).
#define AGX_TU tu_wp32
#include "rb_kernels.hpp"
#include <cstring>
#include "rb32_kernels.hpp"
#define AGX_WP_Q32
#include "wp_kernels.hpp"

namespace agx {
namespace AGX_TU {
const rb_entry kEntries[] = {
    // tier 2 (every q < 2^30); one wave per workgroup (reg_wp.hip)
    make_entry_wp32<5, 3, 1, 2, 8>(230),      // n = 32: 8 x 4
    make_entry_wp32<6, 3, 1, 2, 8>(231),      // n = 64: 8 x 8
    make_entry_wp32<7, 4, 1, 2, 8>(232),      // n = 128: 16 x 8
    make_entry_wp32<8, 4, 1, 2, 8>(233),      // n = 256: 16 x 16
    make_entry_wp32<9, 4, 1, 2, 8>(234),      // n = 512: 16 x 32 (three passes; 32 x 16 spills in the product: 40 vs 55 % of its 24n bytes)
    // tier 1 (every q < 2^31)
    make_entry_wp32<5, 3, 1, 1, 8>(240),
    make_entry_wp32<6, 3, 1, 1, 8>(241),
    make_entry_wp32<7, 4, 1, 1, 8>(242),
    make_entry_wp32<8, 4, 1, 1, 8>(243),
    make_entry_wp32<9, 4, 1, 1, 8>(244),
    // n = 2 ... 16: one lane per frame (reg_wp.hip); tier 2, then tier 1
    make_entry_wp32<1, 1, 1, 2, 8>(260), make_entry_wp32<2, 2, 1, 2, 8>(261), make_entry_wp32<3, 3, 1, 2, 8>(262), make_entry_wp32<4, 4, 1, 2, 8>(263),
    make_entry_wp32<1, 1, 1, 1, 8>(264), make_entry_wp32<2, 2, 1, 1, 8>(265), make_entry_wp32<3, 3, 1, 1, 8>(266), make_entry_wp32<4, 4, 1, 1, 8>(267),
#ifdef AGX_DIAG
    // A/B shapes (profiles/r04_small_sizes_sweeps.txt; 237-239 measured there and deleted)
    make_entry_wp32<5, 5, 4, 2, 4>(235),      // n = 32: ONE LANE per frame, the whole transform in 32 registers, every twiddle a scalar: -7 %
    make_entry_wp32<9, 5, 4, 2, 4>(236),      // n = 512: 32 x 16
#endif
};
}  // namespace AGX_TU

rb_span rb_entries_wp32() { return rb_span{AGX_TU::kEntries, sizeof(AGX_TU::kEntries) / sizeof(AGX_TU::kEntries[0])}; }

}  // namespace agx
