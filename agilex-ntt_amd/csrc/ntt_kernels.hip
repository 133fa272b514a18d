// ntt_kernels.hip -- gfx950 (CDNA4, wave64) kernels for the batched negacyclic NTT.
//
// Two families:
//  * radix-2 / LDS-resident kernels: one workgroup per frame, one butterfly stage per barrier.
//    They perform exactly the reference's operation sequence (src/kernel/ntt.cpp:147-180,
//    298-300, 331-369, 377-394), so they are bit-identical to it even on out-of-contract
//    tables.  Any power-of-two n; used for small n, as the always-available fallback and
//    for the inverse transform.
//  * register-blocked kernels: every thread keeps 2^R coefficients in VGPRs and runs R
//    butterfly stages per pass with no memory traffic; passes exchange through one padded
//    LDS slab.  This is the throughput path (n >= 32; their registry is rb_registry.hpp).
//
// No MFMA: this is 64-bit integer modular arithmetic (v_mad_u64_u32 / v_mul_hi_u32), bounded
// by VALU integer multiply issue and HBM bandwidth.
#include "rb_registry.hpp"
#include "modarith.hpp"

#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include <utility>

namespace agx {

extern __shared__ __attribute__((aligned(16))) unsigned char agx_dyn_lds[];

static constexpr int kMaxLdsLog = 14;  // radix-2 kernels: 16384 coefficients = 128 KiB of the CU's 160 KiB LDS; n = 32768 runs as two blocks + one global stage

// ---------------------------------------------------------------------------------------
// radix-2 forward, LDS resident.  Block = one sub-transform of 2^nb_log coefficients:
// sub-block `blk` of the 2^split_log contiguous blocks a frame falls into after the first
// split_log stages (done by fwd_global_stage).  split_log = 0 -> the whole frame.
// ---------------------------------------------------------------------------------------
__global__ void fwd_radix2_lds(const uint64_t* __restrict__ in, uint64_t* __restrict__ out,
                               const prime_consts* __restrict__ consts, const twpair* __restrict__ tw,
                               uint32_t log_n, uint32_t nb_log, uint32_t split_log,
                               int64_t prime_stride, int64_t poly_stride) {
    uint64_t* x = reinterpret_cast<uint64_t*>(agx_dyn_lds);
    const uint32_t nb = 1u << nb_log;
    const uint32_t prime = blockIdx.y;
    const uint64_t poly = blockIdx.x >> split_log;
    const uint32_t blk = blockIdx.x & ((1u << split_log) - 1u);
    const uint64_t q = consts[prime].q, q2 = q << 1;
    const twpair* twp = tw + ((size_t)prime << log_n);
    const int64_t base = (int64_t)prime * prime_stride + (int64_t)poly * poly_stride + ((int64_t)blk << nb_log);

    for (uint32_t e = threadIdx.x; e < nb; e += blockDim.x) x[e] = in[base + e];

    uint32_t t_log = nb_log - 1;
    for (uint32_t m = 1u << split_log; m < (1u << log_n); m <<= 1, --t_log) {
        __syncthreads();
        const uint32_t t = 1u << t_log;
        const uint32_t m_local = m >> split_log;
        for (uint32_t bf = threadIdx.x; bf < (nb >> 1); bf += blockDim.x) {
            const uint32_t i = bf >> t_log, j = bf & (t - 1);
            const uint32_t pos = (i << (t_log + 1)) + j;
            const twpair w = twp[m + blk * m_local + i];   // ntt.cpp:298-300
            uint64_t a = x[pos], b = x[pos + t];
            ct_butterfly(a, b, w.x, w.y, q, q2);
            if (t == 1) {                                    // ntt.cpp:377-394
                a = reduce_4q(a, q, q2);
                b = reduce_4q(b, q, q2);
            }
            x[pos] = a;
            x[pos + t] = b;
        }
    }
    __syncthreads();
    for (uint32_t e = threadIdx.x; e < nb; e += blockDim.x) out[base + e] = x[e];
}

// one forward stage straight through global memory (only for frames larger than the LDS slab).
// stage s (0 = first): m = 2^s groups, gap t = n / 2^(s+1).
__global__ void fwd_global_stage(const uint64_t* __restrict__ src, uint64_t* __restrict__ dst,
                                 const prime_consts* __restrict__ consts, const twpair* __restrict__ tw,
                                 uint32_t log_n, uint32_t stage, uint64_t batch,
                                 int64_t prime_stride, int64_t poly_stride) {
    const uint32_t prime = blockIdx.y;
    const uint64_t q = consts[prime].q, q2 = q << 1;
    const twpair* twp = tw + ((size_t)prime << log_n);
    const uint32_t half_log = log_n - 1, t_log = log_n - 1 - stage, t = 1u << t_log, m = 1u << stage;
    const uint64_t total = batch << half_log;
    for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t poly = g >> half_log;
        const uint32_t bf = (uint32_t)(g & ((1u << half_log) - 1u));
        const uint32_t i = bf >> t_log, j = bf & (t - 1);
        const int64_t pos = (int64_t)prime * prime_stride + (int64_t)poly * poly_stride + ((int64_t)i << (t_log + 1)) + j;
        const twpair w = twp[m + i];
        uint64_t a = src[pos], b = src[pos + t];
        ct_butterfly(a, b, w.x, w.y, q, q2);
        if (t == 1) { a = reduce_4q(a, q, q2); b = reduce_4q(b, q, q2); }
        dst[pos] = a;
        dst[pos + t] = b;
    }
}

// ---------------------------------------------------------------------------------------
// radix-2 inverse (Gentleman-Sande), LDS resident: stages t = 1 .. nb/2 of each 2^nb_log
// block; the remaining split_log stages (gaps >= nb) run in inv_global_stage.  The stage with
// m = 1 also multiplies by n^-1 and fully reduces.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void gs_last(uint64_t& a, uint64_t& b, const prime_consts& k, uint64_t q2) {
    const uint64_t s = a + b;            // < 4q: mul_shoup_lazy takes any 64-bit value
    const uint64_t d = a + q2 - b;
    a = csub(mul_shoup_lazy(s, k.n_inv, k.n_inv_p, k.q), k.q);
    b = csub(mul_shoup_lazy(d, k.w1n, k.w1n_p, k.q), k.q);
}

__global__ void inv_radix2_lds(const uint64_t* __restrict__ in, uint64_t* __restrict__ out,
                               const prime_consts* __restrict__ consts, const twpair* __restrict__ itw,
                               uint32_t log_n, uint32_t nb_log, uint32_t split_log,
                               int64_t prime_stride, int64_t poly_stride) {
    uint64_t* x = reinterpret_cast<uint64_t*>(agx_dyn_lds);
    const uint32_t nb = 1u << nb_log;
    const uint32_t prime = blockIdx.y;
    const uint64_t poly = blockIdx.x >> split_log;
    const uint32_t blk = blockIdx.x & ((1u << split_log) - 1u);
    const prime_consts k = consts[prime];
    const uint64_t q = k.q, q2 = q << 1;
    const twpair* twp = itw + ((size_t)prime << log_n);
    const int64_t base = (int64_t)prime * prime_stride + (int64_t)poly * poly_stride + ((int64_t)blk << nb_log);

    for (uint32_t e = threadIdx.x; e < nb; e += blockDim.x) x[e] = csub(in[base + e], q2);  // [0,4q) -> [0,2q)

    uint32_t t_log = 0;
    for (uint32_t m = 1u << (log_n - 1); m >= (1u << split_log); m >>= 1, ++t_log) {
        __syncthreads();
        const uint32_t t = 1u << t_log;
        const uint32_t m_local = m >> split_log;
        for (uint32_t bf = threadIdx.x; bf < (nb >> 1); bf += blockDim.x) {
            const uint32_t i = bf >> t_log, j = bf & (t - 1);
            const uint32_t pos = (i << (t_log + 1)) + j;
            uint64_t a = x[pos], b = x[pos + t];
            if (m == 1) {
                gs_last(a, b, k, q2);
            } else {
                const twpair w = twp[m + blk * m_local + i];
                gs_butterfly(a, b, w.x, w.y, q, q2);
            }
            x[pos] = a;
            x[pos + t] = b;
        }
        if (m == 1) break;
    }
    __syncthreads();
    if (nb_log == 0) x[0] = csub(x[0], q);
    for (uint32_t e = threadIdx.x; e < nb; e += blockDim.x) out[base + e] = x[e];
}

__global__ void inv_global_stage(uint64_t* __restrict__ data, const prime_consts* __restrict__ consts,
                                 const twpair* __restrict__ itw, uint32_t log_n, uint32_t stage, uint64_t batch,
                                 int64_t prime_stride, int64_t poly_stride) {
    // `stage` counts like the forward pass: m = 2^stage groups, gap t = n / 2^(stage+1)
    const uint32_t prime = blockIdx.y;
    const prime_consts k = consts[prime];
    const uint64_t q = k.q, q2 = q << 1;
    const twpair* twp = itw + ((size_t)prime << log_n);
    const uint32_t half_log = log_n - 1, t_log = log_n - 1 - stage, t = 1u << t_log, m = 1u << stage;
    const uint64_t total = batch << half_log;
    for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t poly = g >> half_log;
        const uint32_t bf = (uint32_t)(g & ((1u << half_log) - 1u));
        const uint32_t i = bf >> t_log, j = bf & (t - 1);
        const int64_t pos = (int64_t)prime * prime_stride + (int64_t)poly * poly_stride + ((int64_t)i << (t_log + 1)) + j;
        uint64_t a = data[pos], b = data[pos + t];
        if (m == 1) {
            gs_last(a, b, k, q2);
        } else {
            const twpair w = twp[m + i];
            gs_butterfly(a, b, w.x, w.y, q, q2);
        }
        data[pos] = a;
        data[pos + t] = b;
    }
}

// ---------------------------------------------------------------------------------------
// pointwise product and synthetic fill
// ---------------------------------------------------------------------------------------
__global__ void pointwise_kernel(const uint64_t* __restrict__ a, const uint64_t* __restrict__ b, uint64_t* __restrict__ c,
                                 const prime_consts* __restrict__ consts, uint64_t per_prime) {
    const uint32_t prime = blockIdx.y;
    const prime_consts k = consts[prime];
    const barrett128 bk{k.q, k.mu_hi, k.mu_lo};
    const uint64_t q2 = k.q << 1;
    const uint64_t off = (uint64_t)prime * per_prime;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < per_prime; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t u = reduce_4q(a[off + i], k.q, q2), v = reduce_4q(b[off + i], k.q, q2);
        c[off + i] = mul_mod_barrett(u, v, bk);
    }
}

__device__ __forceinline__ uint64_t splitmix_mix(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// element (prime, poly, i) = mix(mix(mix(seed ^ prime) + poly) + i) mod q : counter based, so the
// same coefficients come out whatever the sharding over devices
__global__ void fill_kernel(uint64_t* __restrict__ out, const prime_consts* __restrict__ consts,
                            uint32_t log_n, uint64_t batch, uint64_t first_poly, uint64_t seed) {
    const uint32_t prime = blockIdx.y;
    const uint64_t q = consts[prime].q;
    const uint64_t per_prime = batch << log_n;
    const uint64_t kp = splitmix_mix(seed ^ (uint64_t)prime);
    for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < per_prime; g += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t poly = first_poly + (g >> log_n), i = g & ((1ull << log_n) - 1ull);
        out[(uint64_t)prime * per_prime + g] = splitmix_mix(splitmix_mix(kp + poly) + i) % q;
    }
}

// ---------------------------------------------------------------------------------------
// host side: registry look-up and launches
// ---------------------------------------------------------------------------------------
namespace {

// every group of the registry, one per translation unit
template <class F>
void for_each_entry(F&& f) {
    const rb_span groups[] = {rb_entries_n4096(), rb_entries_s1024(), rb_entries_s2048(), rb_entries_s4096(), rb_entries_s8192(), rb_entries_s16384(), rb_entries_s32768(),
                              rb_entries_q32a(), rb_entries_q32b(), rb_entries_wp(), rb_entries_wp32(),
#ifdef AGX_DIAG
                              rb_entries_diag(),
#endif
    };
    for (const rb_span& g : groups)
        for (size_t i = 0; i < g.count; ++i) f(g.first[i]);
}

const rb_entry* rb_lookup(int id) {
    const rb_entry* hit = nullptr;
    for_each_entry([&](const rb_entry& e) { if (e.id == id && !hit) hit = &e; });
    return hit;
}

unsigned grid_1d(uint64_t work_items, unsigned threads) {
    uint64_t blocks = (work_items + threads - 1) / threads;
    if (blocks < 1) blocks = 1;
    if (blocks > 2048 * 8) blocks = 2048 * 8;  // grid-stride the rest
    return (unsigned)blocks;
}

template <typename F>
hipError_t set_lds_attr(F* fn, size_t bytes) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

}  // namespace

regblock_layout regblock_choose(uint32_t n, int config_id, int arith_level, int narrow_level) {
    regblock_layout rb;
    int log_n = 0;
    while ((1u << log_n) < n) ++log_n;
    if (log_n < 1) return rb;
    // a 32-bit entry is legal when every modulus fits its tier and the tables honour the precon contract
    auto legal = [&](const rb_entry& c) { return c.arith <= arith_level && (c.narrow == 0 || (arith_level >= 1 && narrow_level >= (c.narrow == 2 ? 2 : 1))); };
    const rb_entry* e = nullptr;
    if (config_id >= 0) {
        e = rb_lookup(config_id);
        if (e && (e->log_n != log_n || !legal(*e))) e = nullptr;
    } else {
        // tuned defaults, best first; the lazier arithmetic forms only when every modulus allows them
        static const int kDefaults[] = {260, 261, 262, 263, 264, 265, 266, 267, 230, 231, 232, 233, 234, 240, 241, 242, 243, 244,      // n = 2 ... 512, narrow moduli: wave-packed 32-bit kernels (tier 2, then tier 1)
                                        250, 251, 252, 253, 254, 255, 256, 257,                          // n = 2 ... 16: one lane per frame (fast, exact per size)
                                        200, 201, 202, 203, 204, 205, 206, 207, 208, 209, 210, 211, 212, 213, 214,      // n = 32 ... 512: wave-packed kernels (16q-lazy, fast, exact per size)
                                        130, 131, 132, 133, 134, 135, 136, 137, 138, 139, 140, 141,      // narrow moduli: 32-bit arithmetic (tier 2, then tier 1)
                                        93, 92, 91,                                                      // n = 4096: R = 3, 8 waves/SIMD (16q-lazy, fast, exact)
                                        150, 151, 152, 153, 154, 155, 156, 157, 158,                     // n = 1024 / 2048 / 8192: streamed single-frame kernels
                                        119, 117, 121, 120, 123, 122};                                  // n = 32768 / 16384
        for (int id : kDefaults) {
            const rb_entry* c = rb_lookup(id);
            if (c && c->log_n == log_n && legal(*c)) { e = c; break; }
        }
    }
    if (!e) return rb;
    rb.config_id = e->id;
    rb.log_n = log_n;
    rb.r = e->r;
    rb.pairs_per_prime = e->table_pairs;
    return rb;
}

regblock_layout regblock_forward_companion(const regblock_layout& main, uint32_t n, int arith_level, int narrow_level) {
    const rb_entry* e = main.valid() ? rb_lookup(main.config_id) : nullptr;
    if (!e || e->fwd_companion <= 0) return regblock_layout{};
    regblock_layout rb = regblock_choose(n, e->fwd_companion, arith_level, narrow_level);
    rb.min_frames = e->fwd_companion_min_frames;
    return rb;
}

void regblock_build_table(const regblock_layout& rb, const uint64_t* tw, const uint64_t* pre, std::vector<ulonglong2>& out) {
    const rb_entry* e = rb_lookup(rb.config_id);
    if (e) e->build(rb, tw, pre, out);
}

hipError_t kernels_init() {
    hipError_t e;
    const size_t big = (size_t)8 << kMaxLdsLog;
    if ((e = set_lds_attr(fwd_radix2_lds, big)) != hipSuccess) return e;
    if ((e = set_lds_attr(inv_radix2_lds, big)) != hipSuccess) return e;
    for_each_entry([&](const rb_entry& c) { if (e == hipSuccess) e = c.init(); });
    return e;
}

static unsigned radix2_threads(uint32_t nb) {
    uint32_t t = nb / 2;
    if (t < 64) t = 64;
    if (t > 1024) t = 1024;
    return t;
}

hipError_t launch_forward_radix2(const plan_view& pv, const uint64_t* in, uint64_t* out, const frame_layout& fl, hipStream_t s) {
    const uint32_t split = pv.log_n > (uint32_t)kMaxLdsLog ? pv.log_n - kMaxLdsLog : 0;
    const uint32_t nb_log = pv.log_n - split;
    const uint64_t* src = in;
    for (uint32_t st = 0; st < split; ++st) {
        dim3 grid(grid_1d(fl.batch << (pv.log_n - 1), 256), pv.num_primes);
        hipLaunchKernelGGL(fwd_global_stage, grid, dim3(256), 0, s, src, out, pv.consts, pv.tw, pv.log_n, st, fl.batch,
                           fl.prime_stride, fl.poly_stride);
        src = out;
    }
    dim3 grid((unsigned)(fl.batch << split), pv.num_primes);
    hipLaunchKernelGGL(fwd_radix2_lds, grid, dim3(radix2_threads(1u << nb_log)), (size_t)8 << nb_log, s, src, out, pv.consts,
                       pv.tw, pv.log_n, nb_log, split, fl.prime_stride, fl.poly_stride);
    return hipGetLastError();
}

hipError_t launch_inverse_radix2(const plan_view& pv, const uint64_t* in, uint64_t* out, const frame_layout& fl, hipStream_t s) {
    const uint32_t split = pv.log_n > (uint32_t)kMaxLdsLog ? pv.log_n - kMaxLdsLog : 0;
    const uint32_t nb_log = pv.log_n - split;
    dim3 grid((unsigned)(fl.batch << split), pv.num_primes);
    hipLaunchKernelGGL(inv_radix2_lds, grid, dim3(radix2_threads(1u << nb_log)), (size_t)8 << nb_log, s, in, out, pv.consts,
                       pv.itw, pv.log_n, nb_log, split, fl.prime_stride, fl.poly_stride);
    for (int st = (int)split - 1; st >= 0; --st) {
        dim3 g2(grid_1d(fl.batch << (pv.log_n - 1), 256), pv.num_primes);
        hipLaunchKernelGGL(inv_global_stage, g2, dim3(256), 0, s, out, pv.consts, pv.itw, pv.log_n, (uint32_t)st, fl.batch,
                           fl.prime_stride, fl.poly_stride);
    }
    return hipGetLastError();
}

hipError_t launch_forward_regblock(const plan_view& pv, const uint64_t* in, uint64_t* out, const frame_layout& fl, hipStream_t s) {
    const rb_entry* e = pv.rb.valid() ? rb_lookup(pv.rb.config_id) : nullptr;
    if (!e) return hipErrorInvalidValue;
    return e->launch(pv, in, out, fl, s);
}

bool regblock_has_inverse(const regblock_layout& rb) {
    const rb_entry* e = rb_lookup(rb.config_id);
    return e && e->launch_inv;
}

bool regblock_has_polymul(const regblock_layout& rb) {
    const rb_entry* e = rb_lookup(rb.config_id);
    return e && e->launch_mul;
}

hipError_t launch_inverse_regblock(const plan_view& pv, const uint64_t* in, const uint64_t* in2, uint64_t* out, const frame_layout& fl, hipStream_t s) {
    const rb_entry* e = rb_lookup(pv.rb.config_id);
    if (!e || !e->launch_inv || !pv.itw_rb) return hipErrorInvalidValue;
    return e->launch_inv(pv, in, in2, out, fl, s);
}

hipError_t launch_polymul_regblock(const plan_view& pv, const uint64_t* a, const uint64_t* b, uint64_t* c, const frame_layout& fl, hipStream_t s) {
    const rb_entry* e = rb_lookup(pv.rb.config_id);
    if (!e || !e->launch_mul || !pv.itw_rb) return hipErrorInvalidValue;
    return e->launch_mul(pv, a, b, c, fl, s);
}

hipError_t launch_pointwise(const plan_view& pv, const uint64_t* a, const uint64_t* b, uint64_t* c, uint64_t batch, hipStream_t s) {
    const uint64_t per_prime = batch << pv.log_n;
    dim3 grid(grid_1d(per_prime, 256), pv.num_primes);
    hipLaunchKernelGGL(pointwise_kernel, grid, dim3(256), 0, s, a, b, c, pv.consts, per_prime);
    return hipGetLastError();
}

hipError_t launch_fill(const plan_view& pv, uint64_t* out, uint64_t batch, uint64_t first_poly, uint64_t seed, hipStream_t s) {
    dim3 grid(grid_1d(batch << pv.log_n, 256), pv.num_primes);
    hipLaunchKernelGGL(fill_kernel, grid, dim3(256), 0, s, out, pv.consts, pv.log_n, batch, first_poly, seed);
    return hipGetLastError();
}

}  // namespace agx
