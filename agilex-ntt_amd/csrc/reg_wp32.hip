// reg_wp32.hip -- the wave-packed kernels of the small sizes, n = 32 ... 512, 32-bit arithmetic for plans whose every modulus is below
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
    // tier 2 (every q < 2^30)
    make_entry_wp32<5, 5, 4, 2, 4>(230),      // n = 32: ONE LANE per frame, the whole transform in 32 registers, every twiddle a scalar
    make_entry_wp32<6, 3, 4, 2, 8>(231),      // n = 64: 8 x 8
    make_entry_wp32<7, 4, 4, 2, 8>(232),      // n = 128: 16 x 8
    make_entry_wp32<8, 4, 4, 2, 8>(233),      // n = 256: 16 x 16
    make_entry_wp32<9, 5, 4, 2, 4>(234),      // n = 512: 32 x 16
    // tier 1 (every q < 2^31)
    make_entry_wp32<5, 5, 4, 1, 4>(240),
    make_entry_wp32<6, 3, 4, 1, 8>(241),
    make_entry_wp32<7, 4, 4, 1, 8>(242),
    make_entry_wp32<8, 4, 4, 1, 8>(243),
    make_entry_wp32<9, 5, 4, 1, 4>(244),
#ifdef AGX_DIAG
    // A/B shapes
    make_entry_wp32<5, 3, 4, 2, 8>(235),      // n = 32: 8 x 4
    make_entry_wp32<9, 4, 4, 2, 8>(236),      // n = 512: 16 x 32 (three passes)
    make_entry_wp32<7, 5, 4, 2, 4>(237),      // n = 128: 32 x 4
    make_entry_wp32<8, 5, 4, 2, 4>(238),      // n = 256: 32 x 8
    make_entry_wp32<6, 4, 4, 2, 8>(239),      // n = 64: 16 x 4
#endif
};
}  // namespace AGX_TU

rb_span rb_entries_wp32() { return rb_span{AGX_TU::kEntries, sizeof(AGX_TU::kEntries) / sizeof(AGX_TU::kEntries[0])}; }

}  // namespace agx
