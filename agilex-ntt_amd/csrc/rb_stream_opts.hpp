// rb_stream_opts.hpp -- option sets of the streamed single-frame kernels, shared by the reg_s<n>.hip registry groups
#pragma once
namespace agx {
namespace AGX_TU {
constexpr int kLazy = 1 | ((kOptPad | kOptSelect | kOptLazy16 | kOptLazyInv | kOptNtLoad | kOptNtStore | kOptEstReduce | kOptSplitWord | kOptStreamTw | kOptPinBf) << 1);   // q <= 2^60
constexpr int kFast = 1 | ((kOptPad | kOptSelect | kOptNtLoad | kOptNtStore | kOptSplitWord | kOptStreamTw | kOptPinBf) << 1);                                            // q <= 2^61
constexpr int kExact = 0 | ((kOptPad | kOptNtLoad | kOptNtStore | kOptSplitWord | kOptStreamTw | kOptPinBf) << 1);                                                        // q < 2^62, reference op sequence
}  // namespace AGX_TU
}  // namespace agx
