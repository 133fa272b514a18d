// reg_r5.hip -- one group of the kernel registry (rb_registry.hpp): R = 5 (32 coefficients per thread) whole-frame kernels for
// n = 32768 (1024 threads) and n = 16384 (512 threads), exchanges through the split-word image (4n bytes of LDS).
#define AGX_TU tu_r5
#include "rb_kernels.hpp"
namespace agx {
namespace AGX_TU {
const rb_entry kEntries[] = {
    make_entry_fwd_only<15, 5, 1, 1 | ((kOptPad | kOptSelect | kOptLazy16 | kOptNtLoad | kOptNtStore | kOptEstReduce | kOptSplitWord) << 1), 4>(110),
    make_entry_fwd_only<14, 5, 1, 1 | ((kOptPad | kOptSelect | kOptLazy16 | kOptNtLoad | kOptNtStore | kOptEstReduce | kOptSplitWord) << 1), 4>(111),
    make_entry_fwd_only<15, 5, 1, 1 | ((kOptPad | kOptSelect | kOptLazy16 | kOptNtLoad | kOptNtStore | kOptEstReduce | kOptSplitWord | kOptStreamTw | kOptPinBf) << 1), 4>(112),
    make_entry_fwd_only<14, 5, 1, 1 | ((kOptPad | kOptSelect | kOptLazy16 | kOptNtLoad | kOptNtStore | kOptEstReduce | kOptSplitWord | kOptStreamTw | kOptPinBf) << 1), 4>(113),
    // forward + inverse + parked one-launch product
    make_entry_single<15, 5, 1 | ((kOptPad | kOptSelect | kOptLazy16 | kOptLazyInv | kOptNtLoad | kOptNtStore | kOptEstReduce | kOptSplitWord | kOptStreamTw | kOptPinBf) << 1), 4>(114),
    make_entry_single<14, 5, 1 | ((kOptPad | kOptSelect | kOptLazy16 | kOptLazyInv | kOptNtLoad | kOptNtStore | kOptEstReduce | kOptSplitWord | kOptStreamTw | kOptPinBf) << 1), 4>(115),
    make_entry_single_dloop<15, 5, 1 | ((kOptPad | kOptSelect | kOptLazy16 | kOptLazyInv | kOptNtLoad | kOptNtStore | kOptEstReduce | kOptSplitWord | kOptStreamTw | kOptPinBf) << 1), 4, true, true>(116),
    make_entry_single_dloop<14, 5, 1 | ((kOptPad | kOptSelect | kOptLazy16 | kOptLazyInv | kOptNtLoad | kOptNtStore | kOptEstReduce | kOptSplitWord | kOptStreamTw | kOptPinBf) << 1), 4, false, true>(117),
    make_entry_single_dloop<14, 5, 1 | ((kOptPad | kOptSelect | kOptLazy16 | kOptLazyInv | kOptNtLoad | kOptNtStore | kOptEstReduce | kOptSplitWord | kOptStreamTw | kOptPinBf) << 1), 4, true, true>(118),
};
}  // namespace AGX_TU

rb_span rb_entries_r5() { return rb_span{AGX_TU::kEntries, sizeof(AGX_TU::kEntries) / sizeof(AGX_TU::kEntries[0])}; }

}  // namespace agx
