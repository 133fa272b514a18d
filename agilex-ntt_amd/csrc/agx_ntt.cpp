// agx_ntt.cpp -- the extern "C" boundary declared in include/agx_ntt.h.
// Owns argument validation, plan objects (device tables) and the mapping of HIP errors to
// status codes.  Kernels live in ntt_kernels.hip; number theory in host_math.cpp.
#include "../../include/agx_ntt.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include "host_math.hpp"
#include "ntt_kernels.hpp"
#include "plan_internal.hpp"
#ifdef AGX_DIAG
#include "../../tools/agx_ntt_diag.h"
namespace agx { hipError_t regblock_set_trace(uint64_t* buf, uint64_t waves); }
#endif

using namespace agx;

thread_local int g_last_hip_error = 0;

namespace agx {
int& last_hip_error_slot() { return g_last_hip_error; }
int hip_fail(hipError_t e) {
    g_last_hip_error = (int)e;
    return e == hipErrorOutOfMemory ? AGX_ERR_ALLOC : AGX_ERR_HIP;
}
}  // namespace agx

namespace {

// function attributes (large dynamic LDS) are per device and never change: set them once per device,
// not on every plan creation
hipError_t kernels_init_once(int device) {
    constexpr int kMaxDevices = 64;
    static std::once_flag once[kMaxDevices];
    static hipError_t result[kMaxDevices];
    if (device < 0 || device >= kMaxDevices) return kernels_init();
    std::call_once(once[device], [device] { result[device] = kernels_init(); });
    return result[device];
}

int check_size(uint32_t n) {
    return (n >= AGX_NTT_MIN_N && n <= AGX_NTT_MAX_N && is_pow2(n)) ? AGX_OK : AGX_ERR_BAD_SIZE;
}

// what the butterfly arithmetic needs (src/kernel/ntt.cpp:302-369): 4q < 2^64 and 2n | q-1
int check_modulus(uint64_t q, uint32_t n) {
    if (q < 3 || (q & 1) == 0 || q >= (1ull << 62)) return AGX_ERR_BAD_MODULUS;
    if ((q - 1) % (2ull * n)) return AGX_ERR_BAD_MODULUS;
    return AGX_OK;
}

uint32_t* plan_ticket_for(void* ctx, hipStream_t s) {
    const agx_ntt_plan* p = static_cast<const agx_ntt_plan*>(ctx);
    if (!p->d_ticket) return nullptr;
    // hipStreamPerThread is ONE handle value that names a different stream in every host thread: two threads launching on "it" run
    // concurrently and must not share a pair.  No pair is provably free for it: stateless fixed-stride kernels.
    if (s == hipStreamPerThread) return nullptr;
    std::lock_guard<std::mutex> lock(p->ticket_mu);
    for (size_t i = 0; i < p->ticket_streams.size(); ++i)
        if (p->ticket_streams[i] == s) {
            p->ticket_pending[i] = 1;
            return p->d_ticket + 2 * i;
        }
    size_t slot = p->ticket_streams.size();
    if (slot >= kTicketSlots) {
        // every slot is taken: one whose stream's latest ticket launch has completed is idle -- its pair is zero again -- and can change hands
        slot = kTicketSlots;
        for (size_t i = 0; i < kTicketSlots && slot == kTicketSlots; ++i)
            if (!p->ticket_pending[i] && p->ticket_events[i] && hipEventQuery(p->ticket_events[i]) == hipSuccess) slot = i;
        if (slot == kTicketSlots) return nullptr;      // all busy: the stateless kernels
        p->ticket_streams[slot] = s;
    } else {
        p->ticket_streams.push_back(s);      // capacity reserved at plan creation: no allocation here
    }
    if (!p->ticket_events[slot] && hipEventCreateWithFlags(&p->ticket_events[slot], hipEventDisableTiming) != hipSuccess) {
        p->ticket_events[slot] = nullptr;      // without an event the slot can never be proven idle: it simply stays with this stream
    }
    p->ticket_pending[slot] = 1;
    return p->d_ticket + 2 * slot;
}

void plan_ticket_launched(void* ctx, hipStream_t s, uint32_t* pair) {
    const agx_ntt_plan* p = static_cast<const agx_ntt_plan*>(ctx);
    if (!pair || !p->d_ticket) return;
    const size_t slot = (size_t)(pair - p->d_ticket) / 2;
    std::lock_guard<std::mutex> lock(p->ticket_mu);
    if (slot >= p->ticket_events.size()) return;
    if (p->ticket_events[slot] && hipEventRecord(p->ticket_events[slot], s) != hipSuccess) {
        (void)hipEventDestroy(p->ticket_events[slot]);
        p->ticket_events[slot] = nullptr;
    }
    p->ticket_pending[slot] = 0;
}

plan_view view_of(const agx_ntt_plan* p) {
    plan_view v;
    v.n = p->n;
    v.log_n = p->log_n;
    v.num_primes = p->num_primes;
    v.consts = p->d_consts;
    v.tw = p->d_tw;
    v.itw = p->d_itw;
    v.rb = p->rb;
    v.tw_rb = p->d_tw_rb;
    v.itw_rb = p->d_itw_rb;
    v.ticket_for = &plan_ticket_for;
    v.ticket_launched = &plan_ticket_launched;
    v.ticket_ctx = const_cast<agx_ntt_plan*>(p);
    return v;
}

template <typename T>
int upload(T** dst, const std::vector<T>& src) {
    AGX_HIP(hipMalloc(reinterpret_cast<void**>(dst), src.size() * sizeof(T)));
    AGX_HIP(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return AGX_OK;
}

// n^-1 mod q for odd q and n a power of two: ((q+1)/2)^log2(n)
uint64_t inv_pow2_mod(uint32_t log_n, uint64_t q) {
    const uint64_t half = (q + 1) >> 1;
    uint64_t r = 1 % q;
    for (uint32_t i = 0; i < log_n; ++i) r = mul_mod(r, half, q);
    return r;
}

}  // namespace

namespace agx {

void free_plan(agx_ntt_plan* p) {
    if (!p) return;
    if (p->d_consts) (void)hipFree(p->d_consts);
    if (p->d_ticket) (void)hipFree(p->d_ticket);
    for (hipEvent_t ev : p->ticket_events)
        if (ev) (void)hipEventDestroy(ev);
    if (p->d_tw) (void)hipFree(p->d_tw);
    if (p->d_itw) (void)hipFree(p->d_itw);
    if (p->d_tw_rb) (void)hipFree(p->d_tw_rb);
    if (p->d_itw_rb) (void)hipFree(p->d_itw_rb);
    if (p->d_tw_rb_fwd) (void)hipFree(p->d_tw_rb_fwd);
    delete p;
}

void prepare_plan_image(plan_image& img, uint32_t n, uint32_t num_primes, const uint64_t* moduli, const uint64_t* psi,
                        const uint64_t* tw, const uint64_t* pre, const uint64_t* itw, const uint64_t* ipre) {
    img.n = n;
    img.log_n = (uint32_t)log2u(n);
    img.num_primes = num_primes;
    img.has_inverse = itw != nullptr;
    img.moduli.assign(moduli, moduli + num_primes);
    img.psi.assign(num_primes, 0);
    if (psi) img.psi.assign(psi, psi + num_primes);
    img.consts.assign(num_primes, prime_consts{});
    img.tw_pairs.resize((size_t)num_primes * n);
    if (itw) img.itw_pairs.resize((size_t)num_primes * n);
    // The fast butterfly is only value-equivalent to the reference's when every modulus is <= 2^61
    // and the tables honour the contract precon[j] = floor(twiddle[j] * 2^64 / q), twiddle[j] < q.
    // Tables that do not (e.g. the placeholders of the reference's main.cpp:49-55) get the exact
    // kernels, which repeat the reference's operations mod 2^64 whatever they are fed.
    img.arith_level = 2;
    for (uint32_t k = 0; k < num_primes && img.arith_level > 0; ++k) {
        const uint64_t q = moduli[k];
        if (q > (1ull << 60)) img.arith_level = std::min(img.arith_level, 1);
        if (q > (1ull << 61)) img.arith_level = 0;
        for (uint32_t j = 1; j < n && img.arith_level > 0; ++j) {
            const uint64_t w = tw[(size_t)k * n + j];
            if (w >= q || pre[(size_t)k * n + j] != shoup_quotient(w, q)) img.arith_level = 0;
            // the inverse kernels of a plan use the same arithmetic form: its tables must honour the contract too
            if (itw && img.arith_level > 0) {
                const uint64_t iw = itw[(size_t)k * n + j];
                if (iw >= q || ipre[(size_t)k * n + j] != shoup_quotient(iw, q)) img.arith_level = 0;
            }
        }
    }
    img.narrow_level = 2;
    for (uint32_t k = 0; k < num_primes; ++k) {
        if (moduli[k] >= (1ull << 30)) img.narrow_level = std::min(img.narrow_level, 1);
        if (moduli[k] >= (1ull << 31)) img.narrow_level = 0;
    }
    img.rb = regblock_choose(n, -1, img.arith_level, img.narrow_level);
    img.rb_fwd = regblock_forward_companion(img.rb, n, img.arith_level, img.narrow_level);
    for (uint32_t k = 0; k < num_primes; ++k) {
        const uint64_t q = moduli[k];
        prime_consts& c = img.consts[k];
        std::memset(&c, 0, sizeof(c));
        c.q = q;
        const unsigned __int128 mu = ~(unsigned __int128)0 / q;  // = floor(2^128 / q) for odd q > 1
        c.mu_hi = (uint64_t)(mu >> 64);
        c.mu_lo = (uint64_t)mu;
        if (q >= (1ull << 58) && q <= (1ull << 60)) {
            // reduce_final_est (modarith.hpp) needs v < 16q <= 2^64: only the 16q-lazy kernels (arith level 2, q <= 2^60) call it, and
            // the estimate is only ever set for moduli they accept, so a wider modulus can never reach that path with est != 0
            // float slightly below 2^32 / q (reduce_final_est): scaled down by 2^-20, then rounded toward zero
            float f = (float)(4294967296.0 / (double)q * (1.0 - 1.0 / 1048576.0));
            if ((double)f > 4294967296.0 / (double)q * (1.0 - 1.0 / 1048576.0)) f = std::nextafterf(f, 0.0f);
            uint32_t bits;
            std::memcpy(&bits, &f, sizeof(bits));
            c.est = bits;
        }
        c.n_inv = inv_pow2_mod(img.log_n, q);
        c.n_inv_p = shoup_quotient(c.n_inv, q);
        const uint64_t* twk = tw + (size_t)k * n;
        const uint64_t* prek = pre + (size_t)k * n;
        for (uint32_t j = 0; j < n; ++j) img.tw_pairs[(size_t)k * n + j] = make_ulonglong2(twk[j], prek[j]);
        if (itw) {
            const uint64_t* itwk = itw + (size_t)k * n;
            const uint64_t* iprek = ipre + (size_t)k * n;
            for (uint32_t j = 0; j < n; ++j) img.itw_pairs[(size_t)k * n + j] = make_ulonglong2(itwk[j], iprek[j]);
            c.w1n = mul_mod(itwk[1] % q, c.n_inv, q);
            c.w1n_p = shoup_quotient(c.w1n, q);
            if (img.rb.valid()) regblock_build_table(img.rb, itwk, iprek, img.irb_pairs);
        }
        if (img.rb.valid()) regblock_build_table(img.rb, twk, prek, img.rb_pairs);
        if (img.rb_fwd.valid()) regblock_build_table(img.rb_fwd, twk, prek, img.fwd_pairs);
    }
}

int instantiate_plan(agx_ntt_plan** out, const plan_image& img) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return AGX_ERR_NO_DEVICE;
    agx_ntt_plan* p = new (std::nothrow) agx_ntt_plan;
    if (!p) return AGX_ERR_ALLOC;
    p->n = img.n;
    p->log_n = img.log_n;
    p->num_primes = img.num_primes;
    p->has_inverse = img.has_inverse;
    p->arith_level = img.arith_level;
    p->narrow_level = img.narrow_level;
    p->rb = img.rb;
    p->rb_fwd = img.rb_fwd;
    p->ticket_streams.reserve(kTicketSlots);      // plan_ticket_for() runs inside unguarded launch calls: it must never allocate
    p->ticket_events.assign(kTicketSlots, nullptr);
    p->ticket_pending.assign(kTicketSlots, 0);
    p->moduli = img.moduli;
    p->psi = img.psi;
    int rc = AGX_OK;
    hipError_t he = hipGetDevice(&p->device);
    if (he == hipSuccess) he = kernels_init_once(p->device);
    if (he != hipSuccess) { free_plan(p); return hip_fail(he); }
    if ((rc = upload(&p->d_consts, img.consts)) != AGX_OK) { free_plan(p); return rc; }
    {
        hipError_t te = hipMalloc(reinterpret_cast<void**>(&p->d_ticket), 2 * kTicketSlots * sizeof(uint32_t));
        if (te == hipSuccess) te = hipMemset(p->d_ticket, 0, 2 * kTicketSlots * sizeof(uint32_t));
        if (te != hipSuccess) { free_plan(p); return hip_fail(te); }
    }
    if ((rc = upload(&p->d_tw, img.tw_pairs)) != AGX_OK) { free_plan(p); return rc; }
    if (img.has_inverse && (rc = upload(&p->d_itw, img.itw_pairs)) != AGX_OK) { free_plan(p); return rc; }
    if (p->rb.valid() && (rc = upload(&p->d_tw_rb, img.rb_pairs)) != AGX_OK) { free_plan(p); return rc; }
    if (p->rb.valid() && img.has_inverse && (rc = upload(&p->d_itw_rb, img.irb_pairs)) != AGX_OK) { free_plan(p); return rc; }
    if (p->rb_fwd.valid() && (rc = upload(&p->d_tw_rb_fwd, img.fwd_pairs)) != AGX_OK) { free_plan(p); return rc; }
    *out = p;
    return AGX_OK;
}

}  // namespace agx

namespace {

int build_plan(agx_ntt_plan** out, uint32_t n, uint32_t num_primes, const uint64_t* moduli, const uint64_t* psi,
               const uint64_t* tw, const uint64_t* pre, const uint64_t* itw, const uint64_t* ipre) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return AGX_ERR_NO_DEVICE;
    plan_image img;
    prepare_plan_image(img, n, num_primes, moduli, psi, tw, pre, itw, ipre);
    return instantiate_plan(out, img);
}

// Do two frame sets of the same shape -- frame (p, b) at base + p prime_stride + b poly_stride, n elements each -- overlap without being
// the same set?  Identical bases are in place (legal: a workgroup reads its frame before it writes it); otherwise NO frame of one may
// touch any frame of the other, because workgroups run in any order (include/agx_ntt.h: AGX_ERR_BAD_ARGUMENT "overlapping in/out").
// Interleaved layouts whose frames do not touch (out = in + n with poly_stride = 2n) are legal and pass.
bool partial_overlap(const void* a, const void* b, uint32_t n, uint32_t num_primes, uint64_t batch, int64_t prime_stride, int64_t poly_stride) {
    if (a == b || batch == 0) return false;
    const int64_t delta = (int64_t)((reinterpret_cast<intptr_t>(b) - reinterpret_cast<intptr_t>(a)) / (intptr_t)sizeof(uint64_t));     // elements (both 8-byte aligned)
    const int64_t extent = (int64_t)(num_primes - 1) * prime_stride + (int64_t)(batch - 1) * poly_stride + (int64_t)n;
    if (delta >= extent || -delta >= extent) return false;      // disjoint ranges
    // frames i of A and j of B touch iff |delta + dp prime_stride + db poly_stride| < n for their index differences (dp, db)
    const int64_t P = (int64_t)num_primes, B = (int64_t)batch;
    for (int64_t dp = -(P - 1); dp <= P - 1; ++dp) {
        const int64_t base = delta + dp * prime_stride;
        if (poly_stride == 0 || B == 1) {
            if (base > -(int64_t)n && base < (int64_t)n) return true;
            continue;
        }
        // db closest to -base / poly_stride, within [-(B-1), B-1]: try the two neighbours of the quotient
        int64_t d0 = -base / poly_stride;
        for (int64_t db = d0 - 1; db <= d0 + 1; ++db) {
            const int64_t dbc = std::max<int64_t>(-(B - 1), std::min<int64_t>(B - 1, db));
            const int64_t v = base + dbc * poly_stride;
            if (v > -(int64_t)n && v < (int64_t)n) return true;
        }
    }
    return false;
}

int check_call(const agx_ntt_plan* plan, const void* a, const void* b, uint64_t batch, int64_t prime_stride, int64_t poly_stride) {
    if (!plan || !a || !b) return AGX_ERR_NULL_POINTER;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev != plan->device) return AGX_ERR_BAD_ARGUMENT;   // the plan's tables live on plan->device
    if (prime_stride < 0 || poly_stride < 0) return AGX_ERR_BAD_ARGUMENT;
    if (batch > 1 && poly_stride < (int64_t)plan->n) return AGX_ERR_BAD_ARGUMENT;   // frames would overlap
    if ((batch << (plan->log_n > 14 ? plan->log_n - 14 : 0)) > 0x7fffffffull) return AGX_ERR_BAD_ARGUMENT;  // grid.x limit (the radix-2 kernels split n = 32768 in two blocks)
    if (((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 7u) != 0) return AGX_ERR_BAD_ARGUMENT;      // uint64_t data
    if (partial_overlap(a, b, plan->n, plan->num_primes, batch, prime_stride, poly_stride)) return AGX_ERR_BAD_ARGUMENT;
    return AGX_OK;
}

bool use_regblock(const agx_ntt_plan* plan) {
    if (plan->variant == AGX_VARIANT_LDS_RADIX2) return false;
    return plan->rb.valid();
}

// ---- staging resources of the host-pointer pipeline (agx_ntt_forward_host_stream) ------------------------------------
}  // namespace

namespace agx {
hipError_t staging_set::ensure(size_t want) {
    if (bytes >= want) return hipSuccess;
    destroy();
    hipError_t e = hipSuccess;
    for (int k = 0; k < kSlots && e == hipSuccess; ++k) {
        e = hipHostMalloc(reinterpret_cast<void**>(&pin_in[k]), want, hipHostMallocDefault);
        if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&pin_out[k]), want, hipHostMallocDefault);
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&dev[k]), want);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&st[k], hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&done[k], hipEventDisableTiming);
    }
    if (e == hipSuccess) bytes = want;
    else destroy();
    return e;
}
void staging_set::destroy() {
    for (int k = 0; k < kSlots; ++k) {
        if (st[k]) { (void)hipStreamSynchronize(st[k]); (void)hipStreamDestroy(st[k]); }
        if (done[k]) (void)hipEventDestroy(done[k]);
        if (dev[k]) (void)hipFree(dev[k]);
        if (pin_in[k]) (void)hipHostFree(pin_in[k]);
        if (pin_out[k]) (void)hipHostFree(pin_out[k]);
        st[k] = nullptr; done[k] = nullptr; dev[k] = nullptr; pin_in[k] = nullptr; pin_out[k] = nullptr;
    }
    bytes = 0;
}
}  // namespace agx

namespace {

// one cached set per device, handed to one call at a time; a second concurrent call gets nullptr and builds its own
constexpr int kPoolDevices = 64;
std::mutex g_stage_mu[kPoolDevices];
staging_set g_stage_pool[kPoolDevices];

staging_set* acquire_staging(int device) {
    if (device < 0 || device >= kPoolDevices) return nullptr;
    return g_stage_mu[device].try_lock() ? &g_stage_pool[device] : nullptr;
}
void release_staging(int device) { g_stage_mu[device].unlock(); }

unsigned stage_workers() {
    const unsigned hc = std::thread::hardware_concurrency();
    unsigned cap = 4;
    if (const char* v = std::getenv("AGX_STAGE_WORKERS")) cap = (unsigned)std::max(1, std::atoi(v));   // tuning knob (tools/host_stream_bench.py)
    return std::max(1u, std::min(cap, hc / 2));
}

template <class F>
void parallel_for(uint64_t items, unsigned workers, F&& fn) {
    if (workers <= 1 || items < 2) { fn(0, items); return; }
    const unsigned t = (unsigned)std::min<uint64_t>(workers, items);
    std::vector<std::thread> pool;
    pool.reserve(t - 1);
    for (unsigned w = 1; w < t; ++w) pool.emplace_back([&, w] { fn(items * w / t, items * (w + 1) / t); });
    fn(0, items / t);
    for (std::thread& th : pool) th.join();
}

void parallel_memcpy(void* dst, const void* src, size_t bytes, unsigned workers) {
    const uint64_t blocks = (bytes + 4095) / 4096;      // split on page multiples
    parallel_for(blocks, workers, [&](uint64_t lo, uint64_t hi) {
        const size_t b0 = (size_t)lo * 4096, b1 = std::min<size_t>(bytes, (size_t)hi * 4096);
        if (b1 > b0) std::memcpy(static_cast<char*>(dst) + b0, static_cast<const char*>(src) + b0, b1 - b0);
    });
}

}  // namespace

extern "C" {

const char* agx_ntt_strerror(int status) {
    switch (status) {
        case AGX_OK: return "success";
        case AGX_ERR_NULL_POINTER: return "required pointer argument is NULL";
        case AGX_ERR_BAD_SIZE: return "n must be a power of two in [2, 32768]";
        case AGX_ERR_BAD_MODULUS: return "modulus must be an odd prime < 2^62 with q = 1 (mod 2n)";
        case AGX_ERR_BAD_ROOT: return "psi is not a primitive 2n-th root of unity mod q";
        case AGX_ERR_BAD_ARGUMENT: return "invalid argument";
        case AGX_ERR_NO_DEVICE: return "no HIP device available";
        case AGX_ERR_HIP: return "HIP runtime error (see agx_ntt_last_hip_error)";
        case AGX_ERR_ALLOC: return "allocation failed";
        case AGX_ERR_NO_INVERSE: return "plan has no inverse tables";
    }
    return "unknown status";
}

int agx_ntt_last_hip_error(void) { return g_last_hip_error; }

int agx_ntt_device_count(int* count) {
    if (!count) return AGX_ERR_NULL_POINTER;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
    *count = n;
    return AGX_OK;
}

#ifdef AGX_DIAG
// lib/libagxntt_diag.so only (tools/agx_ntt_diag.h): where the registry's trace kernel writes its phase stamps
int agx_ntt_debug_set_trace_buffer(void* d_buf, uint64_t bytes) {
    if (!d_buf && bytes) return AGX_ERR_NULL_POINTER;
    AGX_HIP(agx::regblock_set_trace(static_cast<uint64_t*>(d_buf), bytes / 128));
    return AGX_OK;
}
#endif

int agx_ntt_plan_create(agx_ntt_plan** plan, uint32_t n, uint32_t num_primes, const uint64_t* moduli,
                        const uint64_t* twiddles, const uint64_t* precons,
                        const uint64_t* inv_twiddles, const uint64_t* inv_precons) {
    if (!plan || !moduli || !twiddles || !precons) return AGX_ERR_NULL_POINTER;
    *plan = nullptr;
    if ((inv_twiddles == nullptr) != (inv_precons == nullptr)) return AGX_ERR_NULL_POINTER;
    int rc = check_size(n);
    if (rc) return rc;
    if (num_primes == 0 || num_primes > 65535) return AGX_ERR_BAD_ARGUMENT;
    for (uint32_t k = 0; k < num_primes; ++k)
        if ((rc = check_modulus(moduli[k], n))) return rc;
    return guarded([&] { return build_plan(plan, n, num_primes, moduli, nullptr, twiddles, precons, inv_twiddles, inv_precons); });
}

int agx_ntt_plan_create_auto(agx_ntt_plan** plan, uint32_t n, uint32_t num_primes, const uint64_t* moduli, const uint64_t* psi) {
    if (!plan || !moduli) return AGX_ERR_NULL_POINTER;
    *plan = nullptr;
    int rc = check_size(n);
    if (rc) return rc;
    if (num_primes == 0 || num_primes > 65535) return AGX_ERR_BAD_ARGUMENT;
    return guarded([&]() -> int {
        std::vector<uint64_t> roots(num_primes), tw((size_t)num_primes * n), pre(tw.size()), itw(tw.size()), ipre(tw.size());
        for (uint32_t k = 0; k < num_primes; ++k) {
            const uint64_t q = moduli[k];
            if (int mrc = check_modulus(q, n)) return mrc;
            if (!is_prime_u64(q)) return AGX_ERR_BAD_MODULUS;
            roots[k] = psi ? psi[k] : min_primitive_root_2n(q, n);
            if (!is_primitive_root_2n(roots[k], q, n)) return AGX_ERR_BAD_ROOT;
            power_tables_bitrev(q, roots[k], n, &tw[(size_t)k * n], &pre[(size_t)k * n]);
            power_tables_bitrev(q, inv_mod(roots[k], q), n, &itw[(size_t)k * n], &ipre[(size_t)k * n]);
        }
        return build_plan(plan, n, num_primes, moduli, roots.data(), tw.data(), pre.data(), itw.data(), ipre.data());
    });
}

int agx_ntt_plan_destroy(agx_ntt_plan* plan) {
    free_plan(plan);
    return AGX_OK;
}

static int plan_set_variant_impl(agx_ntt_plan* plan, int variant);
int agx_ntt_plan_set_variant(agx_ntt_plan* plan, int variant) {
    return guarded([&] { return plan_set_variant_impl(plan, variant); });
}
static int plan_set_variant_impl(agx_ntt_plan* plan, int variant) {
    if (!plan) return AGX_ERR_NULL_POINTER;
    int config_id = -1;
    if (variant >= AGX_VARIANT_REGBLOCK_BASE) {
        config_id = variant - AGX_VARIANT_REGBLOCK_BASE;
        variant = AGX_VARIANT_REGBLOCK;
    }
    if (variant != AGX_VARIANT_AUTO && variant != AGX_VARIANT_LDS_RADIX2 && variant != AGX_VARIANT_REGBLOCK) return AGX_ERR_BAD_ARGUMENT;
    if (variant == AGX_VARIANT_REGBLOCK || variant == AGX_VARIANT_AUTO) {
        // (re)build the pass tables for the requested kernel configuration
        const regblock_layout rb = regblock_choose(plan->n, config_id, plan->arith_level, plan->narrow_level);
        if (!rb.valid()) {
            if (variant == AGX_VARIANT_REGBLOCK) return AGX_ERR_BAD_SIZE;
        } else if (rb.config_id != plan->rb.config_id) {
            std::vector<uint64_t> w(plan->n), wp(plan->n);
            ulonglong2* d_new[2] = {nullptr, nullptr};
            const ulonglong2* d_src[2] = {plan->d_tw, plan->d_itw};
            for (int which = 0; which < 2; ++which) {
                if (!d_src[which]) continue;
                std::vector<ulonglong2> tw((size_t)plan->num_primes * plan->n), rb_pairs;
                AGX_HIP(hipMemcpy(tw.data(), d_src[which], tw.size() * sizeof(ulonglong2), hipMemcpyDeviceToHost));
                for (uint32_t k = 0; k < plan->num_primes; ++k) {
                    for (uint32_t j = 0; j < plan->n; ++j) { w[j] = tw[(size_t)k * plan->n + j].x; wp[j] = tw[(size_t)k * plan->n + j].y; }
                    regblock_build_table(rb, w.data(), wp.data(), rb_pairs);
                }
                int rc = upload(&d_new[which], rb_pairs);
                if (rc != AGX_OK) {
                    for (ulonglong2* d : d_new)
                        if (d) (void)hipFree(d);
                    return rc;
                }
            }
            AGX_HIP(hipDeviceSynchronize());
            if (plan->d_tw_rb) (void)hipFree(plan->d_tw_rb);
            if (plan->d_itw_rb) (void)hipFree(plan->d_itw_rb);
            plan->d_tw_rb = d_new[0];
            plan->d_itw_rb = d_new[1];
            plan->rb = rb;
        }
    }
    if (config_id >= 0 || variant == AGX_VARIANT_LDS_RADIX2) plan->rb_fwd = regblock_layout{};      // an explicit kernel choice applies to every call
    else if (plan->d_tw_rb_fwd) plan->rb_fwd = regblock_forward_companion(plan->rb, plan->n, plan->arith_level, plan->narrow_level);
    plan->variant = variant;
    return AGX_OK;
}

int agx_ntt_plan_info(const agx_ntt_plan* plan, uint32_t* n, uint32_t* num_primes, int* device, int* has_inverse) {
    if (!plan) return AGX_ERR_NULL_POINTER;
    if (n) *n = plan->n;
    if (num_primes) *num_primes = plan->num_primes;
    if (device) *device = plan->device;
    if (has_inverse) *has_inverse = plan->has_inverse ? 1 : 0;
    return AGX_OK;
}

int agx_ntt_plan_get_modulus(const agx_ntt_plan* plan, uint32_t prime_index, uint64_t* q, uint64_t* psi) {
    if (!plan) return AGX_ERR_NULL_POINTER;
    if (prime_index >= plan->num_primes) return AGX_ERR_BAD_ARGUMENT;
    if (q) *q = plan->moduli[prime_index];
    if (psi) *psi = plan->psi[prime_index];
    return AGX_OK;
}

static int forward_common(const agx_ntt_plan* plan, const uint64_t* d_in, uint64_t* d_out, uint64_t batch,
                          int64_t prime_stride, int64_t poly_stride, bool lazy_out, void* stream) {
    int rc = check_call(plan, d_in, d_out, batch, prime_stride, poly_stride);
    if (rc) return rc;
    if (batch == 0) return AGX_OK;
    frame_layout fl{batch, prime_stride, poly_stride};
    fl.lazy_out = lazy_out;
    plan_view pv = view_of(plan);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (use_regblock(plan) && plan->rb_fwd.valid() && batch * plan->num_primes >= plan->rb_fwd.min_frames) {      // the forward-only companion of the tuned default
        pv.rb = plan->rb_fwd;
        pv.tw_rb = plan->d_tw_rb_fwd;
    }
    AGX_HIP(use_regblock(plan) ? launch_forward_regblock(pv, d_in, d_out, fl, s) : launch_forward_radix2(pv, d_in, d_out, fl, s));
    return AGX_OK;
}

int agx_ntt_forward_strided(const agx_ntt_plan* plan, const uint64_t* d_in, uint64_t* d_out, uint64_t batch,
                            int64_t prime_stride, int64_t poly_stride, void* stream) {
    return forward_common(plan, d_in, d_out, batch, prime_stride, poly_stride, false, stream);
}

int agx_ntt_forward_lazy(const agx_ntt_plan* plan, const uint64_t* d_in, uint64_t* d_out, uint64_t batch, void* stream) {
    if (!plan) return AGX_ERR_NULL_POINTER;
    return forward_common(plan, d_in, d_out, batch, (int64_t)(batch * plan->n), (int64_t)plan->n, true, stream);
}

int agx_ntt_inverse_strided(const agx_ntt_plan* plan, const uint64_t* d_in, uint64_t* d_out, uint64_t batch,
                            int64_t prime_stride, int64_t poly_stride, void* stream) {
    int rc = check_call(plan, d_in, d_out, batch, prime_stride, poly_stride);
    if (rc) return rc;
    if (!plan->has_inverse) return AGX_ERR_NO_INVERSE;
    if (batch == 0) return AGX_OK;
    const frame_layout fl{batch, prime_stride, poly_stride};
    const plan_view pv = view_of(plan);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool fast_path = use_regblock(plan) && regblock_has_inverse(plan->rb) && plan->d_itw_rb;
    AGX_HIP(fast_path ? launch_inverse_regblock(pv, d_in, nullptr, d_out, fl, s) : launch_inverse_radix2(pv, d_in, d_out, fl, s));
    return AGX_OK;
}

int agx_ntt_forward(const agx_ntt_plan* plan, const uint64_t* d_in, uint64_t* d_out, uint64_t batch, void* stream) {
    if (!plan) return AGX_ERR_NULL_POINTER;
    return agx_ntt_forward_strided(plan, d_in, d_out, batch, (int64_t)(batch * plan->n), (int64_t)plan->n, stream);
}

int agx_ntt_inverse(const agx_ntt_plan* plan, const uint64_t* d_in, uint64_t* d_out, uint64_t batch, void* stream) {
    if (!plan) return AGX_ERR_NULL_POINTER;
    return agx_ntt_inverse_strided(plan, d_in, d_out, batch, (int64_t)(batch * plan->n), (int64_t)plan->n, stream);
}

int agx_ntt_pointwise(const agx_ntt_plan* plan, const uint64_t* d_a, const uint64_t* d_b, uint64_t* d_c, uint64_t batch, void* stream) {
    if (!plan || !d_a || !d_b || !d_c) return AGX_ERR_NULL_POINTER;
    int rc = check_call(plan, d_a, d_c, batch, (int64_t)(batch * plan->n), (int64_t)plan->n);
    if (rc) return rc;
    if ((rc = check_call(plan, d_b, d_c, batch, (int64_t)(batch * plan->n), (int64_t)plan->n))) return rc;
    if (batch == 0) return AGX_OK;
    AGX_HIP(launch_pointwise(view_of(plan), d_a, d_b, d_c, batch, static_cast<hipStream_t>(stream)));
    return AGX_OK;
}

int agx_ntt_polymul(const agx_ntt_plan* plan, const uint64_t* d_a, const uint64_t* d_b, uint64_t* d_c,
                    uint64_t* d_scratch, uint64_t batch, void* stream) {
    if (!plan || !d_a || !d_b || !d_c) return AGX_ERR_NULL_POINTER;
    int rc = check_call(plan, d_a, d_c, batch, (int64_t)(batch * plan->n), (int64_t)plan->n);
    if (rc) return rc;
    if ((rc = check_call(plan, d_b, d_c, batch, (int64_t)(batch * plan->n), (int64_t)plan->n))) return rc;      // c may BE a or b, never straddle them
    if (!plan->has_inverse) return AGX_ERR_NO_INVERSE;
    if (batch == 0) return AGX_OK;
    if (use_regblock(plan) && regblock_has_polymul(plan->rb) && plan->d_itw_rb) {
        // one kernel: both forward transforms, the product and the inverse stay on chip (24n bytes of HBM traffic);
        // every workgroup reads its a and b frames completely before it writes c, so c may alias either
        const frame_layout fl{batch, (int64_t)(batch * plan->n), (int64_t)plan->n};
        AGX_HIP(launch_polymul_regblock(view_of(plan), d_a, d_b, d_c, fl, static_cast<hipStream_t>(stream)));
        return AGX_OK;
    }
    if (!d_scratch) return AGX_ERR_NULL_POINTER;
    for (const uint64_t* other : {d_a, d_b, (const uint64_t*)d_c})      // the scratch frames must not touch a, b or c anywhere
        if (d_scratch == other || partial_overlap(other, d_scratch, plan->n, plan->num_primes, batch, (int64_t)(batch * plan->n), (int64_t)plan->n))
            return AGX_ERR_BAD_ARGUMENT;
    // scratch <- NTT(a); c <- NTT(b) (a is dead by now, so c may alias it); c <- INTT(c o scratch).
    // With the register-blocked inverse the product is taken while it loads (no pointwise pass) and
    // the forward results may stay lazily reduced.
    const bool fused_tail = use_regblock(plan) && regblock_has_inverse(plan->rb) && plan->d_itw_rb;
    if (fused_tail) {
        if ((rc = agx_ntt_forward_lazy(plan, d_a, d_scratch, batch, stream))) return rc;
        if ((rc = agx_ntt_forward_lazy(plan, d_b, d_c, batch, stream))) return rc;
        const frame_layout fl{batch, (int64_t)(batch * plan->n), (int64_t)plan->n};
        AGX_HIP(launch_inverse_regblock(view_of(plan), d_c, d_scratch, d_c, fl, static_cast<hipStream_t>(stream)));
        return AGX_OK;
    }
    if ((rc = agx_ntt_forward(plan, d_a, d_scratch, batch, stream))) return rc;
    if ((rc = agx_ntt_forward(plan, d_b, d_c, batch, stream))) return rc;
    if ((rc = agx_ntt_pointwise(plan, d_c, d_scratch, d_c, batch, stream))) return rc;
    return agx_ntt_inverse(plan, d_c, d_c, batch, stream);
}

int agx_ntt_fill_synthetic(const agx_ntt_plan* plan, uint64_t* d_out, uint64_t batch, uint64_t first_poly, uint64_t seed, void* stream) {
    if (!plan || !d_out) return AGX_ERR_NULL_POINTER;
    int rc = check_call(plan, d_out, d_out, batch, (int64_t)(batch * plan->n), (int64_t)plan->n);
    if (rc) return rc;
    if (batch == 0) return AGX_OK;
    AGX_HIP(launch_fill(view_of(plan), d_out, batch, first_poly, seed, static_cast<hipStream_t>(stream)));
    return AGX_OK;
}

// Host-resident frames through a device plan with transfers and compute overlapped: the GPU
// analogue of the reference's streaming ntt_input_kernel / ntt_output_kernel pair
// (src/kernel/ntt.cpp:508-640).  Three slots of pinned staging + device memory rotate over three
// streams: while slot k computes, slot k+1 uploads and the host thread assembles slot k+2
// (lower half of each frame from `in`, upper half from `in2`, src/kernel/ntt.cpp:584-590).
int agx_ntt_forward_host_stream(const agx_ntt_plan* plan, const uint64_t* in, const uint64_t* in2, uint64_t* out,
                                uint64_t num_frames) {
    return guarded([&] { return host_stream_pipeline(plan, in, in2, out, num_frames, false, nullptr); });      // std::thread / std::vector inside may throw
}

// the inverse transform through the same pipeline (bit-reversed order in, natural order out; no operand pairing: the reference has no
// inverse path)
int agx_ntt_inverse_host_stream(const agx_ntt_plan* plan, const uint64_t* in, uint64_t* out, uint64_t num_frames) {
    return guarded([&] { return host_stream_pipeline(plan, in, in, out, num_frames, true, nullptr); });
}

}  // extern "C"

namespace agx {

int host_stream_pipeline(const agx_ntt_plan* plan, const uint64_t* in, const uint64_t* in2, uint64_t* out, uint64_t num_frames,
                         bool inverse, staging_set* own) {
    if (!plan || !in || !in2 || !out) return AGX_ERR_NULL_POINTER;
    if (plan->num_primes != 1) return AGX_ERR_BAD_ARGUMENT;   // one modulus per stream, as the reference (ntt.cpp:143-144)
    if (inverse && !plan->has_inverse) return AGX_ERR_NO_INVERSE;
    {
        int dev = -1;
        if (hipGetDevice(&dev) != hipSuccess || dev != plan->device) return AGX_ERR_BAD_ARGUMENT;   // staging memory is allocated on the current device
    }
    if (num_frames == 0) return AGX_OK;
    auto transform = [&](uint64_t* d, uint64_t frames, hipStream_t s) {
        return inverse ? agx_ntt_inverse(plan, d, d, frames, s) : agx_ntt_forward(plan, d, d, frames, s);
    };
    const size_t n = plan->n, row = n * sizeof(uint64_t), half = row / 2;
    if (num_frames * row <= ((size_t)4 << 20)) {
        // small inputs: staging buffers and streams would cost more than they hide
        uint64_t* d = nullptr;
        hipError_t se = hipMalloc(reinterpret_cast<void**>(&d), row * num_frames);
        int src = AGX_OK;
        if (se == hipSuccess) se = hipMemcpy2D(d, row, in, row, half, num_frames, hipMemcpyHostToDevice);
        if (se == hipSuccess) se = hipMemcpy2D(reinterpret_cast<char*>(d) + half, row, reinterpret_cast<const char*>(in2) + half, row, half, num_frames, hipMemcpyHostToDevice);
        if (se == hipSuccess) {
            src = transform(d, num_frames, nullptr);
            if (src == AGX_OK) se = hipMemcpy(out, d, row * num_frames, hipMemcpyDeviceToHost);
        }
        if (d) (void)hipFree(d);
        if (src != AGX_OK) return src;
        return se == hipSuccess ? AGX_OK : hip_fail(se);
    }
    // Pipeline: 3 slots of {pinned in, pinned out, device} on 3 streams.  The ABI takes pageable host pointers, so every
    // byte is also copied once into and once out of pinned memory by the CPU; at ~12 GiB/s per core that staging, not
    // PCIe, is the bottleneck of a single-threaded pipeline (tools/host_stream_bench.py).  Hence: the staging copies of a
    // chunk are split over a few worker threads, the drain of chunk c-3 runs beside the staging of chunk c, and the
    // pinned / device slots and streams are kept between calls (allocating 192 MiB of pinned memory costs more than moving
    // 1 GiB through it): in the caller's own set (a group's shard), else in a per-device pool.
    // Whatever leaves this function -- a status, an exception from std::thread -- `lease` first waits for the slot streams and
    // then gives the pooled set back / frees the temporary one (a throw used to leave the pool's lock held for good).
    struct lease_t {
        staging_set* set = nullptr;
        staging_set local;
        int pooled_device = -1;
        ~lease_t() {
            if (set)
                for (int k = 0; k < staging_set::kSlots; ++k)
                    if (set->st[k]) (void)hipStreamSynchronize(set->st[k]);
            if (pooled_device >= 0) release_staging(pooled_device);
            local.destroy();
        }
    } lease;
    if (own) {
        lease.set = own;
    } else if (staging_set* pooled = acquire_staging(plan->device)) {
        lease.set = pooled;
        lease.pooled_device = plan->device;
    } else {
        lease.set = &lease.local;
    }
    staging_set* set = lease.set;
    hipError_t e = set->ensure(kStageChunkBytes);
    int rc = AGX_OK;
    const uint64_t chunk = std::max<uint64_t>(1, std::min<uint64_t>(num_frames, kStageChunkBytes / row));
    const uint64_t nchunks = (num_frames + chunk - 1) / chunk;
    const int slots = (int)std::min<uint64_t>(staging_set::kSlots, nchunks);
    const unsigned workers = stage_workers();
    auto frames_of = [&](uint64_t c) { return std::min<uint64_t>(chunk, num_frames - c * chunk); };
    auto copy_out = [&](uint64_t c) {   // chunk c's results: pinned -> caller's buffer (ntt.cpp:628-633)
        parallel_memcpy(out + c * chunk * n, set->pin_out[c % slots], frames_of(c) * row, workers);
    };
    auto drain = [&](uint64_t c) {
        hipError_t de = hipEventSynchronize(set->done[c % slots]);
        if (de == hipSuccess) copy_out(c);
        return de;
    };
    for (uint64_t c = 0; c < nchunks && e == hipSuccess && rc == AGX_OK; ++c) {
        const int k = (int)(c % slots);
        joining_thread drainer;      // joined on every path out of this iteration
        if (c >= (uint64_t)slots) {
            // slot k still belongs to chunk c-slots: done[k] (recorded behind its download) says its upload source
            // pin_in[k] is free again and its results sit in pin_out[k]
            e = hipEventSynchronize(set->done[k]);
            if (e != hipSuccess) break;
            drainer = joining_thread([&, c] { copy_out(c - slots); });     // beside the staging of chunk c
        }
        const uint64_t f = frames_of(c);
        const uint64_t* a = in + c * chunk * n;
        const uint64_t* b = in2 + c * chunk * n;
        if (a == b) {
            parallel_memcpy(set->pin_in[k], a, f * row, workers);
        } else {
            // lower half of each frame from `in`, upper half from `in2` (src/kernel/ntt.cpp:584-590)
            parallel_for(f, workers, [&](uint64_t lo, uint64_t hi) {
                for (uint64_t i = lo; i < hi; ++i) {
                    std::memcpy(reinterpret_cast<char*>(set->pin_in[k]) + i * row, reinterpret_cast<const char*>(a) + i * row, half);
                    std::memcpy(reinterpret_cast<char*>(set->pin_in[k]) + i * row + half, reinterpret_cast<const char*>(b) + i * row + half, half);
                }
            });
        }
        e = hipMemcpyAsync(set->dev[k], set->pin_in[k], f * row, hipMemcpyHostToDevice, set->st[k]);
        if (e == hipSuccess) rc = transform(set->dev[k], f, set->st[k]);
        drainer.join();      // pin_out[k] must be empty before this chunk's download may land in it
        if (e == hipSuccess && rc == AGX_OK) e = hipMemcpyAsync(set->pin_out[k], set->dev[k], f * row, hipMemcpyDeviceToHost, set->st[k]);
        if (e == hipSuccess && rc == AGX_OK) e = hipEventRecord(set->done[k], set->st[k]);
    }
    for (uint64_t c = nchunks > (uint64_t)slots ? nchunks - slots : 0; c < nchunks && e == hipSuccess && rc == AGX_OK; ++c) e = drain(c);
    if (rc != AGX_OK) return rc;
    return e == hipSuccess ? AGX_OK : hip_fail(e);
}

}  // namespace agx

extern "C" {

// One-shot calls usually repeat with the same (n, modulus, tables): building a plan verifies every table
// entry (a 128-bit divide each) and uploads four tables, which costs tens of milliseconds, so the last
// plan of each device is kept and reused while the caller's tables still compare equal word for word.
// One slot and one lock PER DEVICE: calls on different GPUs neither evict each other's plan nor serialise.
namespace {
struct oneshot_cache {
    std::mutex mu;
    agx_ntt_plan* plan = nullptr;
    uint64_t q = 0;
    std::vector<uint64_t> tw, pre;     // host copies of the tables the cached plan was built from
};
oneshot_cache g_oneshot[kPoolDevices];

// AGX_NTT_DEVICES=0,1,2,3 in the environment: the one-shot call deals its frames to those devices (a group, include/agx_ntt.h section 5)
// instead of running on the current one -- the reference's NUM_NTT_COMPUTE_UNITS replication (src/kernel/ntt.cpp:8-12, 526-536) for a
// caller that cannot change its code.  One cached group, rebuilt when (n, modulus, tables, device list) change.
struct oneshot_group_cache {
    std::mutex mu;
    agx_ntt_group* group = nullptr;
    uint32_t n = 0;
    uint64_t q = 0;
    std::vector<uint64_t> tw, pre;
    std::vector<int> devices;
};
oneshot_group_cache g_oneshot_group;

void env_device_list(std::vector<int>& out) {
    out.clear();
    const char* v = std::getenv("AGX_NTT_DEVICES");
    if (!v || !*v) return;
    for (const char* p = v; *p;) {
        char* end = nullptr;
        const long d = std::strtol(p, &end, 10);
        if (end == p) {      // not a number: a device id the group will refuse
            out.assign(1, -1);
            return;
        }
        out.push_back((int)d);
        p = end;
        while (*p == ',' || *p == ' ') ++p;
    }
}
}  // namespace

int agx_ntt_forward_host(const uint64_t* in, const uint64_t* in2, const uint64_t* modulus,
                         const uint64_t* twiddles, const uint64_t* precons, uint64_t* out,
                         uint32_t n, uint32_t num_frames) {
    if (!in || !in2 || !modulus || !twiddles || !precons || !out) return AGX_ERR_NULL_POINTER;
    int rc = check_size(n);
    if (rc) return rc;
    if ((rc = check_modulus(modulus[0], n))) return rc;
    if (num_frames == 0) return AGX_OK;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return AGX_ERR_NO_DEVICE;
    try {
        std::vector<int> devices;
        env_device_list(devices);
        if (!devices.empty()) {
            oneshot_group_cache& c = g_oneshot_group;
            std::lock_guard<std::mutex> lock(c.mu);
            const bool hit = c.group && c.n == n && c.q == modulus[0] && c.devices == devices &&
                             std::memcmp(c.tw.data(), twiddles, (size_t)n * 8) == 0 && std::memcmp(c.pre.data(), precons, (size_t)n * 8) == 0;
            if (!hit) {
                agx_ntt_group* g = nullptr;
                if ((rc = agx_ntt_group_create(&g, devices.data(), (uint32_t)devices.size(), n, 1, modulus, twiddles, precons, nullptr, nullptr))) return rc;
                agx_ntt_group_destroy(c.group);
                c.group = g;
                c.n = n;
                c.q = modulus[0];
                c.tw.assign(twiddles, twiddles + n);
                c.pre.assign(precons, precons + n);
                c.devices = devices;
            }
            return agx_ntt_group_forward_host(c.group, in, in2, out, num_frames);
        }
        if (dev < 0 || dev >= kPoolDevices) {      // no cache slot: build, use, destroy
            agx_ntt_plan* plan = nullptr;
            if ((rc = build_plan(&plan, n, 1, modulus, nullptr, twiddles, precons, nullptr, nullptr))) return rc;
            rc = agx_ntt_forward_host_stream(plan, in, in2, out, num_frames);
            free_plan(plan);
            return rc;
        }
        oneshot_cache& c = g_oneshot[dev];
        std::lock_guard<std::mutex> lock(c.mu);    // one-shot calls are synchronous; those of one device also serialise
        const bool hit = c.plan && c.plan->n == n && c.q == modulus[0] &&
                         std::memcmp(c.tw.data(), twiddles, (size_t)n * 8) == 0 && std::memcmp(c.pre.data(), precons, (size_t)n * 8) == 0;
        if (!hit) {
            agx_ntt_plan* plan = nullptr;
            if ((rc = build_plan(&plan, n, 1, modulus, nullptr, twiddles, precons, nullptr, nullptr))) return rc;
            free_plan(c.plan);
            c.plan = plan;
            c.q = modulus[0];
            c.tw.assign(twiddles, twiddles + n);
            c.pre.assign(precons, precons + n);
        }
        return agx_ntt_forward_host_stream(c.plan, in, in2, out, num_frames);
    } catch (const std::bad_alloc&) {
        return AGX_ERR_ALLOC;
    } catch (...) {
        return AGX_ERR_BAD_ARGUMENT;
    }
}

// Frees what the library keeps between calls: the one-shot plans and the pinned / device staging buffers of every device
// (192 MiB pinned + 96 MiB device memory per device that has streamed host frames).  Call it before hipDeviceReset or at
// shutdown; calls running at the same time keep what they hold (a busy staging set is skipped).  Later calls rebuild on demand.
int agx_ntt_release_caches(void) {
    {
        std::lock_guard<std::mutex> lock(g_oneshot_group.mu);
        agx_ntt_group_destroy(g_oneshot_group.group);      // joins its workers; each frees its shard on its own device
        g_oneshot_group.group = nullptr;
        g_oneshot_group.tw.clear();
        g_oneshot_group.pre.clear();
    }
    for (int d = 0; d < kPoolDevices; ++d) {
        {
            std::lock_guard<std::mutex> lock(g_oneshot[d].mu);
            if (g_oneshot[d].plan) {
                free_plan(g_oneshot[d].plan);
                g_oneshot[d].plan = nullptr;
                g_oneshot[d].tw.clear();
                g_oneshot[d].pre.clear();
            }
        }
        if (g_stage_mu[d].try_lock()) {
            if (g_stage_pool[d].bytes) {
                int cur = -1;
                if (hipGetDevice(&cur) == hipSuccess && hipSetDevice(d) == hipSuccess) {
                    g_stage_pool[d].destroy();
                    (void)hipSetDevice(cur);
                }
            }
            g_stage_mu[d].unlock();
        }
    }
    return AGX_OK;
}

int agx_ntt_find_primes(uint32_t bits, uint32_t n, uint32_t count, uint64_t* primes_out) {
    if (!primes_out) return AGX_ERR_NULL_POINTER;
    int rc = check_size(n);
    if (rc) return rc;
    if (bits < 2 || bits > 62) return AGX_ERR_BAD_ARGUMENT;
    return guarded([&]() -> int {
        std::vector<uint64_t> v = find_ntt_primes(bits, n, count);
        if (v.size() < count) return AGX_ERR_BAD_ARGUMENT;
        std::memcpy(primes_out, v.data(), sizeof(uint64_t) * count);
        return AGX_OK;
    });
}

int agx_ntt_min_root(uint64_t q, uint32_t n, uint64_t* psi_out) {
    if (!psi_out) return AGX_ERR_NULL_POINTER;
    int rc = check_size(n);
    if (rc) return rc;
    if ((rc = check_modulus(q, n))) return rc;
    if (!is_prime_u64(q)) return AGX_ERR_BAD_MODULUS;
    const uint64_t r = min_primitive_root_2n(q, n);
    if (!r) return AGX_ERR_BAD_ROOT;
    *psi_out = r;
    return AGX_OK;
}

static int make_tables_common(uint64_t q, uint64_t psi, uint32_t n, uint64_t* tw, uint64_t* pre, bool inverse) {
    if (!tw || !pre) return AGX_ERR_NULL_POINTER;
    int rc = check_size(n);
    if (rc) return rc;
    if ((rc = check_modulus(q, n))) return rc;
    if (!is_prime_u64(q)) return AGX_ERR_BAD_MODULUS;
    if (!is_primitive_root_2n(psi, q, n)) return AGX_ERR_BAD_ROOT;
    power_tables_bitrev(q, inverse ? inv_mod(psi, q) : psi, n, tw, pre);
    return AGX_OK;
}

int agx_ntt_make_tables(uint64_t q, uint64_t psi, uint32_t n, uint64_t* twiddles, uint64_t* precons) {
    return make_tables_common(q, psi, n, twiddles, precons, false);
}

int agx_ntt_make_inverse_tables(uint64_t q, uint64_t psi, uint32_t n, uint64_t* inv_twiddles, uint64_t* inv_precons) {
    return make_tables_common(q, psi, n, inv_twiddles, inv_precons, true);
}

}  // extern "C"
