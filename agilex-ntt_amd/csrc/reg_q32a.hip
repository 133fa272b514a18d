// reg_q32a.hip (tier 2 of the narrow-modulus kernels) -- one group of the kernel registry (rb_registry.hpp); ids are stable handles for tests and A/B runs
// (AGX_VARIANT_REGBLOCK_BASE + id), not indices.
//
// Narrow-modulus kernels (rb32_kernels.hpp): 32-bit Shoup / Harvey butterflies for plans whose every modulus is below 2^30
// (tier 2: ids 130-135) or below 2^31 (tier 1: 136-141), n = 1024 ... 32768, forward + inverse + one-launch product each.
// R = 4 (16 coefficients per thread, one VGPR each) up to n = 16384; n = 32768 takes R = 5 with 1024 threads.
#define AGX_TU tu_q32a
#include "rb_kernels.hpp"
#include <cstring>
#include "rb32_kernels.hpp"

namespace agx {
namespace AGX_TU {
const rb_entry kEntries[] = {
    // one frame per workgroup at every size: at n = 1024 (one wave per frame: no workgroup barrier at all) and n = 2048 a workgroup used to hold
    // four / two frames; a frame that retires alone frees its slot at once: +5 % forward, +1 ... 2 % inverse and product (profiles/r04_q32_one_frame_per_workgroup.txt)
    make_entry_q32<10, 4, 1, 2, 8>(130),
    make_entry_q32<11, 4, 1, 2, 8>(131),
    make_entry_q32<12, 4, 1, 2, 8>(132),
    make_entry_q32<13, 4, 1, 2, 8>(133),
    make_entry_q32<14, 4, 1, 2, 8>(134),     // 68 KiB image: two 1024-thread workgroups per CU
    make_entry_q32<15, 5, 1, 2, 4>(135),
};
}  // namespace AGX_TU

rb_span rb_entries_q32a() { return rb_span{AGX_TU::kEntries, sizeof(AGX_TU::kEntries) / sizeof(AGX_TU::kEntries[0])}; }

}  // namespace agx
