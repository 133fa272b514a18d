/*
 * agx_ntt.h -- C ABI of the MI355X-native batched negacyclic NTT engine.
 *
 * This is the drop-in boundary for the forward-NTT path of joekurina/Agilex-NTT
 * (reference paths are relative to the reference tree).  Everything is plain
 * pointers and sizes: no SYCL, HIP or torch types appear in a signature
 * (streams are passed as `void*` holding a hipStream_t; NULL = default stream).
 *
 * Data contract (verified against the reference arithmetic):
 *   one "frame" = one length-n polynomial under one modulus q, n a power of two,
 *   q prime, q = 1 (mod 2n), q < 2^62, coefficients uint64_t.
 *   forward : out[bitrev(k)] = sum_j x[j] * psi^((2k+1) j) mod q, fully reduced to [0,q)
 *             (natural order in, bit-reversed order out; inputs may lie in [0,4q)).
 *   inverse : exact inverse of forward (bit-reversed in, natural out, [0,q)).
 *   tables  : twiddle[j] = psi^bitrev(j) mod q,  precon[j] = floor(twiddle[j] * 2^64 / q),
 *             j = 0..n-1 (index 0 unused by the transform), as the reference expects
 *             its caller to supply them (include/kernel/ntt.h:35-41).
 *
 * Every function returns an agx_status (0 = success) and never throws.
 */
#ifndef AGX_NTT_H
#define AGX_NTT_H

#include <stddef.h>
#include <stdint.h>

/* the library is built with -fvisibility=hidden: only the entry points declared here are exported */
#if defined(__GNUC__) || defined(__clang__)
#define AGX_API __attribute__((visibility("default")))
#else
#define AGX_API
#endif

#ifdef __cplusplus
extern "C" {
#endif

typedef enum agx_status {
    AGX_OK = 0,
    AGX_ERR_NULL_POINTER = 1,   /* a required pointer argument is NULL                      */
    AGX_ERR_BAD_SIZE = 2,       /* n is not a power of two in [AGX_NTT_MIN_N, AGX_NTT_MAX_N] */
    AGX_ERR_BAD_MODULUS = 3,    /* q >= 2^62, q even, q != 1 (mod 2n) or (plan_create_auto) q composite */
    AGX_ERR_BAD_ROOT = 4,       /* psi is not a primitive 2n-th root of unity mod q           */
    AGX_ERR_BAD_ARGUMENT = 5,   /* zero primes, negative stride, overlapping in/out, ...      */
    AGX_ERR_NO_DEVICE = 6,      /* no usable HIP device                                       */
    AGX_ERR_HIP = 7,            /* a HIP runtime call failed (agx_ntt_last_hip_error)         */
    AGX_ERR_ALLOC = 8,          /* host or device allocation failed                           */
    AGX_ERR_NO_INVERSE = 9      /* plan was created without inverse tables                    */
} agx_status;

#define AGX_NTT_MIN_N 2u
#define AGX_NTT_MAX_N 32768u  /* the reference's largest size (include/kernel/ntt.h:19-20) */

/* kernel variants (agx_ntt_plan_set_variant); AUTO picks the tuned kernel for n */
#define AGX_VARIANT_AUTO 0
#define AGX_VARIANT_LDS_RADIX2 1 /* one stage per barrier, LDS resident: mirrors the reference's op sequence */
#define AGX_VARIANT_REGBLOCK 2   /* register-blocked radix-2^R passes, tuned configuration for n */
#define AGX_VARIANT_REGBLOCK_BASE 256 /* + k: k-th entry of the kernel registry (A/B measurements only) */

AGX_API const char* agx_ntt_strerror(int status);
AGX_API int agx_ntt_last_hip_error(void);     /* hipError_t of the last AGX_ERR_HIP on this thread */
AGX_API int agx_ntt_device_count(int* count); /* AGX_OK and *count = 0 when there is no GPU */

/* ------------------------------------------------------------------------- */
/* (1) One-shot host-pointer forward NTT.                                     */
/* Replaces the reference's three calls taken together:                        */
/*   ntt_input_kernel(inData, inData2, modulus, twiddleFactors,                */
/*                    barrettTwiddleFactors, numFrames, q)  include/kernel/ntt.h:35-41 */
/*   fwd_ntt_kernel<0>(q)                                   include/kernel/ntt.h:32-33 */
/*   ntt_output_kernel(outData, numFrames, q)               include/kernel/ntt.h:43-45 */
/* Reads only in[b*n + 0 .. n/2) and in2[b*n + n/2 .. n) (src/kernel/ntt.cpp:584-590; */
/* in2 may alias in), writes out[b*n + p] (src/kernel/ntt.cpp:628-633).  n is a */
/* runtime argument here (the reference fixes it at compile time,              */
/* include/kernel/ntt.h:7-23).  Synchronous; all pointers are host memory.     */
/* ------------------------------------------------------------------------- */
/* With AGX_NTT_DEVICES=0,1,2,3 in the environment the frames are dealt to those devices in contiguous blocks (a group, section 5) instead */
/* of running on the current one: the reference's NUM_NTT_COMPUTE_UNITS replication (src/kernel/ntt.cpp:8-12, 526-536) for a caller that  */
/* cannot change its code; a device id that does not exist returns AGX_ERR_BAD_ARGUMENT.                                                  */
AGX_API int agx_ntt_forward_host(const uint64_t* in, const uint64_t* in2, const uint64_t* modulus,
                         const uint64_t* twiddles, const uint64_t* precons, uint64_t* out,
                         uint32_t n, uint32_t num_frames);

/* The same path for repeated calls and large inputs: tables stay on the device in `plan` (one modulus) */
/* and host frames stream through pinned staging buffers with upload, transform and download          */
/* overlapped on three HIP streams (the reference streams frames through its input/output kernels,    */
/* src/kernel/ntt.cpp:508-640).  agx_ntt_forward_host is this call on a plan it builds from the caller's */
/* tables and keeps for the next call with the same (n, modulus, tables).                              */
struct agx_ntt_plan;
AGX_API int agx_ntt_forward_host_stream(const struct agx_ntt_plan* plan, const uint64_t* in, const uint64_t* in2,
                                uint64_t* out, uint64_t num_frames);
/* the inverse transform through the same pipeline (bit-reversed order in, natural order out, one modulus; the reference ships no */
/* inverse path -- SURVEY F2 -- so there is no operand pairing to mirror)                                                         */
AGX_API int agx_ntt_inverse_host_stream(const struct agx_ntt_plan* plan, const uint64_t* in, uint64_t* out, uint64_t num_frames);

/* Releases what the library keeps between calls: the one-shot plan of every device (agx_ntt_forward_host keeps the plan of its */
/* last call per device) and the per-device pool of pinned / device staging buffers of the streaming path (192 MiB pinned +     */
/* 96 MiB device memory per device that has used it).  Call before hipDeviceReset and at shutdown; later calls rebuild on demand. */
/* (The reference holds its buffers in SYCL RAII objects of main(), src/main.cpp:32-37; this is the explicit counterpart.)       */
AGX_API int agx_ntt_release_caches(void);

/* ------------------------------------------------------------------------- */
/* (2) Plans: device-resident tables for num_primes moduli of one size n.      */
/* A plan is immutable after creation and may be shared between host threads   */
/* and streams (kernels that hand out frames through a counter keep one counter */
/* pair per stream, for up to 64 distinct streams per plan; launches on further */
/* streams -- and on hipStreamPerThread, one handle that names a different      */
/* stream in every host thread -- take a stateless kernel form: slower by a few */
/* per cent, never wrong);                                                       */
/* it belongs to the HIP device that was current when it was created: calls    */
/* that take it return AGX_ERR_BAD_ARGUMENT while another device is current.   */
/* ------------------------------------------------------------------------- */
typedef struct agx_ntt_plan agx_ntt_plan;

/* caller-supplied tables, laid out [num_primes][n] (one reference launch per prime,
 * src/kernel/ntt.cpp:143-144,569).  inv_* may both be NULL (forward-only plan).
 * n_inv[p] = n^-1 mod q_p is derived internally. */
AGX_API int agx_ntt_plan_create(agx_ntt_plan** plan, uint32_t n, uint32_t num_primes, const uint64_t* moduli,
                        const uint64_t* twiddles, const uint64_t* precons,
                        const uint64_t* inv_twiddles, const uint64_t* inv_precons);

/* tables generated by the library; psi == NULL -> smallest primitive 2n-th root per prime */
AGX_API int agx_ntt_plan_create_auto(agx_ntt_plan** plan, uint32_t n, uint32_t num_primes,
                             const uint64_t* moduli, const uint64_t* psi);
AGX_API int agx_ntt_plan_destroy(agx_ntt_plan* plan);
/* testing / benchmarking only.  NOT thread-safe: it rebuilds the plan's pass tables, so no launch that uses the plan may be in flight */
/* or be issued from another thread while it runs (it synchronises the device before it frees the old tables)                         */
AGX_API int agx_ntt_plan_set_variant(agx_ntt_plan* plan, int variant);
AGX_API int agx_ntt_plan_info(const agx_ntt_plan* plan, uint32_t* n, uint32_t* num_primes, int* device, int* has_inverse);
AGX_API int agx_ntt_plan_get_modulus(const agx_ntt_plan* plan, uint32_t prime_index, uint64_t* q, uint64_t* psi);

/* ------------------------------------------------------------------------- */
/* (3) Device-pointer batched transforms.  Frame (p, b) starts at              */
/* base + p*prime_stride + b*poly_stride (strides in uint64_t elements);       */
/* the dense forms use the [prime][batch][n] layout (prime_stride = batch*n,   */
/* poly_stride = n).  In place (d_out == d_in) is allowed; an output whose     */
/* frames touch the input's frames without being the same frames (d_out =      */
/* d_in + n/2, c = a + 8 ...) returns AGX_ERR_BAD_ARGUMENT: workgroups run in  */
/* any order, so such a call would corrupt its own inputs.  Asynchronous on    */
/* `stream`; nothing is allocated or synchronised inside, so the calls can be  */
/* captured into a hipGraph (kernels that hand out frames through a counter    */
/* switch to a stateless form while the stream is capturing).                   */
/* ------------------------------------------------------------------------- */
AGX_API int agx_ntt_forward(const agx_ntt_plan* plan, const uint64_t* d_in, uint64_t* d_out, uint64_t batch, void* stream);
AGX_API int agx_ntt_inverse(const agx_ntt_plan* plan, const uint64_t* d_in, uint64_t* d_out, uint64_t batch, void* stream);
/* forward with HEXL-style lazy outputs: results are congruent to the transform and lie in [0,4q)
 * (the last two conditional subtracts of src/kernel/ntt.cpp:377-394 are skipped where that saves
 * work); agx_ntt_inverse, agx_ntt_pointwise and agx_ntt_forward all accept such values as inputs */
AGX_API int agx_ntt_forward_lazy(const agx_ntt_plan* plan, const uint64_t* d_in, uint64_t* d_out, uint64_t batch, void* stream);
AGX_API int agx_ntt_forward_strided(const agx_ntt_plan* plan, const uint64_t* d_in, uint64_t* d_out, uint64_t batch,
                            int64_t prime_stride, int64_t poly_stride, void* stream);
AGX_API int agx_ntt_inverse_strided(const agx_ntt_plan* plan, const uint64_t* d_in, uint64_t* d_out, uint64_t batch,
                            int64_t prime_stride, int64_t poly_stride, void* stream);

/* c = a o b (coefficient-wise product mod q_p), dense [prime][batch][n] layout; c may alias a or b */
AGX_API int agx_ntt_pointwise(const agx_ntt_plan* plan, const uint64_t* d_a, const uint64_t* d_b, uint64_t* d_c,
                      uint64_t batch, void* stream);
/* c = a * b in Z_q[X]/(X^n + 1) = INTT(NTT(a) o NTT(b)); dense layout; c may alias a, b or both (squaring in place: a == b == c).
 * d_scratch: num_primes*batch*n elements of device memory owned by the caller, disjoint from a, b, c.
 * Every size has a one-launch fused kernel, so d_scratch may be NULL; it is only used by plans forced onto the radix-2
 * kernels (AGX_VARIANT_LDS_RADIX2: three launches), where a NULL scratch returns AGX_ERR_NULL_POINTER. */
AGX_API int agx_ntt_polymul(const agx_ntt_plan* plan, const uint64_t* d_a, const uint64_t* d_b, uint64_t* d_c,
                    uint64_t* d_scratch, uint64_t batch, void* stream);

/* synthetic coefficients generated on the device: frame (p,b) element i =
 * splitmix64(seed, p, first_poly + b, i) mod q_p, a pure function of its indices (bench / tests) */
AGX_API int agx_ntt_fill_synthetic(const agx_ntt_plan* plan, uint64_t* d_out, uint64_t batch, uint64_t first_poly,
                           uint64_t seed, void* stream);

/* ------------------------------------------------------------------------- */
/* (4) Host math the reference leaves to its caller (src/main.cpp:49-55 ships   */
/* placeholders only).                                                          */
/* ------------------------------------------------------------------------- */
/* the `count` largest primes q < 2^bits with q = 1 (mod 2n), descending */
AGX_API int agx_ntt_find_primes(uint32_t bits, uint32_t n, uint32_t count, uint64_t* primes_out);
AGX_API int agx_ntt_min_root(uint64_t q, uint32_t n, uint64_t* psi_out);
AGX_API int agx_ntt_make_tables(uint64_t q, uint64_t psi, uint32_t n, uint64_t* twiddles, uint64_t* precons);
AGX_API int agx_ntt_make_inverse_tables(uint64_t q, uint64_t psi, uint32_t n, uint64_t* inv_twiddles, uint64_t* inv_precons);

/* ------------------------------------------------------------------------- */
/* (5) Groups: the same calls over several GPUs of one node.                    */
/* The reference deals the frames of a call to its replicated compute units     */
/* inside the call: unit i gets floor(F / C) + [i < F mod C] frames             */
/* (src/kernel/ntt.cpp:526-536), frame b goes to unit b % C (:579-582) and is   */
/* collected from it (:622-625); units never exchange data.  A group is that    */
/* scheme over whole devices: one SHARD per entry of `devices` (a device may be */
/* listed more than once), each with its own plan (tables prepared once on the  */
/* host), stream, staging buffers and host thread.  Frames are dealt in          */
/* CONTIGUOUS blocks of the reference's minibatch sizes (agx_ntt_shard_range).   */
/* No collective, no peer access: nothing is exchanged between shards.           */
/* A bad device id returns AGX_ERR_BAD_ARGUMENT; a call's status is that of the  */
/* lowest-numbered failing shard.  One call at a time per group (calls from      */
/* several host threads serialise).                                               */
/* ------------------------------------------------------------------------- */
typedef struct agx_ntt_group agx_ntt_group;

/* block of shard `index` when num_frames frames are dealt to num_shards shards: pure arithmetic, no device needed */
AGX_API int agx_ntt_shard_range(uint64_t num_frames, uint32_t num_shards, uint32_t index, uint64_t* first, uint64_t* count);

/* arguments after num_devices as agx_ntt_plan_create / agx_ntt_plan_create_auto */
AGX_API int agx_ntt_group_create(agx_ntt_group** group, const int* devices, uint32_t num_devices, uint32_t n, uint32_t num_primes,
                                 const uint64_t* moduli, const uint64_t* twiddles, const uint64_t* precons,
                                 const uint64_t* inv_twiddles, const uint64_t* inv_precons);
AGX_API int agx_ntt_group_create_auto(agx_ntt_group** group, const int* devices, uint32_t num_devices, uint32_t n, uint32_t num_primes,
                                      const uint64_t* moduli, const uint64_t* psi);
AGX_API int agx_ntt_group_destroy(agx_ntt_group* group);
AGX_API int agx_ntt_group_info(const agx_ntt_group* group, uint32_t* num_shards, uint32_t* n, uint32_t* num_primes);
/* shard `index`: its device, its plan (owned by the group) and its stream (a hipStream_t: record events on it to time a shard) */
AGX_API int agx_ntt_group_shard(const agx_ntt_group* group, uint32_t index, int* device, agx_ntt_plan** plan, void** stream);

/* host frames (one modulus): agx_ntt_forward_host_stream / agx_ntt_inverse_host_stream on every shard's block at once, each shard */
/* from its own host thread through its own three-slot pipeline; synchronous                                                       */
AGX_API int agx_ntt_group_forward_host(const agx_ntt_group* group, const uint64_t* in, const uint64_t* in2, uint64_t* out, uint64_t num_frames);
AGX_API int agx_ntt_group_inverse_host(const agx_ntt_group* group, const uint64_t* in, uint64_t* out, uint64_t num_frames);

/* device pointers: arrays with one entry per shard -- d_in[i] / d_out[i] live on shard i's device in the dense [prime][batch[i]][n] */
/* layout.  Every shard is launched from its own host thread on its own stream; the call returns when every launch has been queued  */
/* (asynchronous like agx_ntt_forward); agx_ntt_group_synchronize waits for all shards' streams.                                    */
AGX_API int agx_ntt_group_forward(const agx_ntt_group* group, const uint64_t* const* d_in, uint64_t* const* d_out, const uint64_t* batch);
AGX_API int agx_ntt_group_inverse(const agx_ntt_group* group, const uint64_t* const* d_in, uint64_t* const* d_out, const uint64_t* batch);
/* d_scratch may be NULL (or hold NULL entries) wherever agx_ntt_polymul accepts a NULL scratch */
AGX_API int agx_ntt_group_polymul(const agx_ntt_group* group, const uint64_t* const* d_a, const uint64_t* const* d_b, uint64_t* const* d_c,
                                  uint64_t* const* d_scratch, const uint64_t* batch);
AGX_API int agx_ntt_group_synchronize(const agx_ntt_group* group);

#ifdef __cplusplus
}
#endif
#endif /* AGX_NTT_H */
