// reg_s1024.hip -- one size of the streamed single-frame kernels; a group of the kernel registry (rb_registry.hpp); ids are stable handles for tests and A/B runs
// (AGX_VARIANT_REGBLOCK_BASE + id), not indices.
//
// The streamed single-frame kernels (rb_kernels.hpp: kOptStreamTw, kOptPinBf, kOptSplitWord): every thread keeps 32 coefficients
// (R = 5; 16 at n = 1024), three passes, exchanges through the split-word image (4n bytes of LDS), twiddles streamed two table
// entries at a time, butterflies pinned in program order -- 108-126 VGPRs and no scratch, where the same arithmetic scheduled freely
// wants 205.  They are the defaults of every size but n = 4096:
//   n = 1024 (R = 4) / 2048: ONE WAVE per frame -- no workgroup barrier anywhere in the transform;  +5 / +10 % forward, +11 / +8 % inverse
//                            over the 128- / 256-thread R = 3 kernels at 8 waves/SIMD (the fewer, fatter waves win: 5 / 4 waves per SIMD)
//   n = 8192:  256 threads per frame, four workgroups per CU;                                         +7 % forward, +5 % inverse, +7 % product
//   n = 16384: 512 threads, 68 KiB image, two workgroups per CU;                                      forward 37 -> 41 % of 8 TB/s, product 16.9 -> 19.6 %
//   n = 32768: 1024 threads, 136 KiB image, the WHOLE frame resident (16n bytes of traffic);          forward 27-29 -> 34-37 %, inverse 30 -> 32-36 %,
//                                                                                                      product in ONE launch 11.8 -> 16.6 % of its 24n bytes
//   (n = 4096 as 128 threads x 32 coefficients, id 127: +1 % over the R = 3 default at the same 1400 W -- not adopted)
// One frame in registers at a time: the fused product parks NTT(first) thread-privately in c's frame (polymul_rb2_park); n = 1024,
// where two frames of 16 coefficients fit, keeps both in registers (polymul_rb2).
#define AGX_TU tu_s1024
#include "rb_kernels.hpp"
#include "rb_stream_opts.hpp"

namespace agx {
namespace AGX_TU {
const rb_entry kEntries[] = {
    // n = 1024: one wave per frame, R = 4, five waves per SIMD; product with both frames in registers
    make_entry_single_mul2<10, 4, kLazy, 5, 4>(150),
    make_entry_single_mul2<10, 4, kFast, 5, 4>(151),
    make_entry_single_mul2<10, 4, kExact, 5, 4>(152),
};
}  // namespace AGX_TU

rb_span rb_entries_s1024() { return rb_span{AGX_TU::kEntries, sizeof(AGX_TU::kEntries) / sizeof(AGX_TU::kEntries[0])}; }

}  // namespace agx
