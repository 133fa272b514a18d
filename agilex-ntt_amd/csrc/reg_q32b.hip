// reg_q32b.hip (tier 1 of the narrow-modulus kernels) -- one group of the kernel registry (rb_registry.hpp); ids are stable handles for tests and A/B runs
// (AGX_VARIANT_REGBLOCK_BASE + id), not indices.
//
// Narrow-modulus kernels (rb32_kernels.hpp): 32-bit Shoup / Harvey butterflies for plans whose every modulus is below 2^30
// (tier 2: ids 130-135) or below 2^31 (tier 1: 136-141), n = 1024 ... 32768, forward + inverse + one-launch product each.
// R = 4 (16 coefficients per thread, one VGPR each) up to n = 16384; n = 32768 takes R = 5 with 1024 threads.
#define AGX_TU tu_q32b
#include "rb_kernels.hpp"
#include <cstring>
#include "rb32_kernels.hpp"

namespace agx {
namespace AGX_TU {
const rb_entry kEntries[] = {
    make_entry_q32<10, 4, 1, 1, 8>(136),
    make_entry_q32<11, 4, 1, 1, 8>(137),
    make_entry_q32<12, 4, 1, 1, 8>(138),
    make_entry_q32<13, 4, 1, 1, 8>(139),
    make_entry_q32<14, 4, 1, 1, 8>(140),
    make_entry_q32<15, 5, 1, 1, 4>(141),
};
}  // namespace AGX_TU

rb_span rb_entries_q32b() { return rb_span{AGX_TU::kEntries, sizeof(AGX_TU::kEntries) / sizeof(AGX_TU::kEntries[0])}; }

}  // namespace agx
