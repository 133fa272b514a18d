// reg_wp.hip -- the wave-packed kernels of the small sizes, n = 2 ... 512, 64-bit arithmetic (wp_kernels.hpp); a group of the kernel
// registry (rb_registry.hpp): ids are stable handles for tests and A/B runs (AGX_VARIANT_REGBLOCK_BASE + id), not indices.
#define AGX_TU tu_wp
#include "rb_kernels.hpp"
#include "rb_stream_opts.hpp"
#include "wp_kernels.hpp"

namespace agx {
namespace AGX_TU {
// streamed twiddles + pinned butterflies as in the single-frame kernels of the large sizes (rb_stream_opts.hpp), full 64-bit image
constexpr int kWpLazy = 1 | ((kOptPad | kOptSelect | kOptLazy16 | kOptLazyInv | kOptNtLoad | kOptNtStore | kOptEstReduce | kOptStreamTw | kOptPinBf) << 1);   // q <= 2^60
constexpr int kWpFast = 1 | ((kOptPad | kOptSelect | kOptNtLoad | kOptNtStore | kOptStreamTw | kOptPinBf) << 1);                                            // q <= 2^61
constexpr int kWpExact = 0 | ((kOptPad | kOptNtLoad | kOptNtStore | kOptStreamTw | kOptPinBf) << 1);                                                        // q < 2^62, reference op sequence
// One wave per workgroup everywhere: nothing is shared between the waves of a group but the LDS allocation, and a wave that retires alone frees
// its slot at once (four waves per workgroup: -1 ... -5 %, profiles/r04_small_sizes_sweeps.txt).
const rb_entry kEntries[] = {
    // n = 32: 8 coefficients per lane, 4 lanes per frame, 16 frames per wave
    make_entry_wp<5, 3, 1, kWpLazy, 8>(200), make_entry_wp<5, 3, 1, kWpFast, 8>(201), make_entry_wp<5, 3, 1, kWpExact, 8>(202),
    // n = 64: 8 x 8
    make_entry_wp<6, 3, 1, kWpLazy, 8>(203), make_entry_wp<6, 3, 1, kWpFast, 8>(204), make_entry_wp<6, 3, 1, kWpExact, 8>(205),
    // n = 128: 16 coefficients per lane, 8 lanes per frame
    make_entry_wp<7, 4, 1, kWpLazy, 5>(206), make_entry_wp<7, 4, 1, kWpFast, 5>(207), make_entry_wp<7, 4, 1, kWpExact, 5>(208),
    // n = 256: 16 x 16
    make_entry_wp<8, 4, 1, kWpLazy, 5>(209), make_entry_wp<8, 4, 1, kWpFast, 5>(210), make_entry_wp<8, 4, 1, kWpExact, 5>(211),
    // n = 512: 16 x 32
    make_entry_wp<9, 4, 1, kWpLazy, 5>(212), make_entry_wp<9, 4, 1, kWpFast, 5>(213), make_entry_wp<9, 4, 1, kWpExact, 5>(214),
    // n = 2 ... 16 (below the reference's size table): ONE LANE per frame -- the whole transform in a lane's registers, every twiddle a scalar, no
    // exchange at all; 64 frames per wave.  Two forms only (fast for q <= 2^61, else exact): 0.03-0.13 butterflies per byte, nothing to gain from laziness.
    make_entry_wp<1, 1, 1, kWpFast, 8>(250), make_entry_wp<1, 1, 1, kWpExact, 8>(251),
    make_entry_wp<2, 2, 1, kWpFast, 8>(252), make_entry_wp<2, 2, 1, kWpExact, 8>(253),
    make_entry_wp<3, 3, 1, kWpFast, 8>(254), make_entry_wp<3, 3, 1, kWpExact, 8>(255),
    make_entry_wp<4, 4, 1, kWpFast, 5>(256), make_entry_wp<4, 4, 1, kWpExact, 5>(257),
#ifdef AGX_DIAG
    // A/B shapes kept in lib/libagxntt_diag.so (measured in profiles/r04_small_sizes_sweeps.txt; 216-219, 223, 225, 226 measured there and deleted)
    make_entry_wp<5, 5, 4, kWpLazy | (kOptSplitWord << 1), 4>(215),      // n = 32: ONE LANE per frame, every twiddle a scalar, split-word image: -5 % forward
    make_entry_wp<9, 5, 1, kWpLazy | (kOptSplitWord << 1), 4>(220),      // n = 512: 32 x 16 (two passes): +1 ... +2 % forward / inverse, -5 % product
    make_entry_single_mul2<9, 3, kLazy, 8, 5>(221),                      // n = 512: one wave per frame, 8 x 64 (the large sizes' kernel shape): equal
    make_entry_wp<8, 4, 4, kWpLazy, 5>(222),                             // n = 256: four waves per workgroup
    make_entry_wp<5, 3, 4, kWpLazy, 8>(224),                             // n = 32: four waves per workgroup: -5 % forward, -7 % inverse and product
#endif
};
}  // namespace AGX_TU

rb_span rb_entries_wp() { return rb_span{AGX_TU::kEntries, sizeof(AGX_TU::kEntries) / sizeof(AGX_TU::kEntries[0])}; }

}  // namespace agx
