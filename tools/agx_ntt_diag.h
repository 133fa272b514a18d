/*
 * agx_ntt_diag.h -- the one extra entry point of lib/libagxntt_diag.so (`make -C agilex-ntt_amd diag`), the
 * product library plus the diagnostics / A/B kernels of csrc/reg_diag.hip.  Not part of the drop-in boundary
 * (include/agx_ntt.h) and absent from lib/libagxntt.so.  No reference counterpart.
 */
#ifndef AGX_NTT_DIAG_H
#define AGX_NTT_DIAG_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
/* tools/timeline.py: the registry's trace kernel (AGX_VARIANT_REGBLOCK_BASE + 70) writes 16 u64 per wave
 * (12 s_memtime phase stamps, HW_ID, XCC_ID) into this device buffer; NULL/0 turns it off. */
__attribute__((visibility("default"))) int agx_ntt_debug_set_trace_buffer(void* d_buf, uint64_t bytes);
#ifdef __cplusplus
}
#endif
#endif
