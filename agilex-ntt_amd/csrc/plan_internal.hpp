// plan_internal.hpp -- what agx_ntt.cpp (plans, single-device calls) and agx_group.cpp (multi-device groups) share.
// Not installed; the public surface is include/agx_ntt.h.
#pragma once
#include "../../include/agx_ntt.h"

#include <hip/hip_runtime.h>

#include <mutex>
#include <new>
#include <system_error>
#include <thread>
#include <vector>

#include "ntt_kernels.hpp"

constexpr uint32_t kTicketSlots = 64;    // distinct streams of one plan that may run ticket-drawing kernels; further streams get the stateless kernels

struct agx_ntt_plan {
    uint32_t n = 0, log_n = 0, num_primes = 0;
    int device = -1;
    int variant = AGX_VARIANT_AUTO;
    bool has_inverse = false;
    int arith_level = 0;     // 0 exact only; 1: every modulus <= 2^61 (8q-lazy legal); 2: <= 2^60 (16q-lazy legal)
    int narrow_level = 0;    // 1: every modulus < 2^31, 2: < 2^30 -- the 32-bit kernels are legal (with arith_level >= 1: tables honour the contract)
    std::vector<uint64_t> moduli, psi;  // psi = 0 when the tables came from the caller
    agx::prime_consts* d_consts = nullptr;
    ulonglong2* d_tw = nullptr;
    ulonglong2* d_itw = nullptr;
    ulonglong2* d_tw_rb = nullptr;
    ulonglong2* d_itw_rb = nullptr;
    agx::regblock_layout rb;
    agx::regblock_layout rb_fwd;            // forward-only layout (another kernel shape that is faster for the forward transform), or invalid
    ulonglong2* d_tw_rb_fwd = nullptr;
    // {next frame, retired workgroups} pairs for the kernels that hand out frames dynamically (the inverse loop kernels of n >= 16384),
    // one pair per STREAM: launches on one stream run one after the other, and the last workgroup of a launch zeroes the
    // pair, so a stream's launches can share one; launches on different streams get different pairs.  Only a launcher that needs a
    // pair asks for one (plan_view::ticket).  When all kTicketSlots are taken, a new stream gets the slot of a stream whose latest ticket
    // launch has provably completed (its event has fired: the pair is idle and zero); if there is none, no pair: the stateless fixed-stride
    // kernels (slots used to be claimed for good, so a long-lived plan used with short-lived streams ended up on the slower form).
    uint32_t* d_ticket = nullptr;
    mutable std::mutex ticket_mu;
    mutable std::vector<hipStream_t> ticket_streams;
    // slot i: an event recorded behind the stream's latest ticket launch (once it has completed the pair reads {0, 0} again: the last
    // workgroup out zeroes it), and whether a launch that took the slot has not recorded its event yet
    mutable std::vector<hipEvent_t> ticket_events;
    mutable std::vector<char> ticket_pending;
};


namespace agx {

int hip_fail(hipError_t e);      // records the HIP error for agx_ntt_last_hip_error() and maps it to a status
int& last_hip_error_slot();      // this thread's agx_ntt_last_hip_error() value (a group copies its failing shard's into the caller's)

#define AGX_HIP(expr)                                    \
    do {                                                 \
        hipError_t e_ = (expr);                          \
        if (e_ != hipSuccess) return agx::hip_fail(e_);  \
    } while (0)

// nothing may propagate across the C boundary: std::vector / std::thread can throw inside the entry points
template <class F>
int guarded(F&& f) {
    try {
        return f();
    } catch (const std::bad_alloc&) {
        return AGX_ERR_ALLOC;
    } catch (const std::system_error&) {      // std::thread could not be created
        return AGX_ERR_ALLOC;
    } catch (...) {
        return AGX_ERR_BAD_ARGUMENT;
    }
}

// a std::thread that is always joined before it is destroyed (unwinding past a joinable std::thread calls std::terminate)
struct joining_thread {
    std::thread t;
    joining_thread() = default;
    template <class F>
    explicit joining_thread(F&& f) : t(std::forward<F>(f)) {}
    joining_thread(joining_thread&&) = default;
    joining_thread& operator=(joining_thread&& o) {
        join();
        t = std::move(o.t);
        return *this;
    }
    void join() {
        if (t.joinable()) t.join();
    }
    ~joining_thread() { join(); }
};

// Everything of a plan that does not depend on the device: validated levels, per-prime constants and the tables in the layouts the
// kernels read.  Built once on the host (a 128-bit divide per table entry), uploaded to as many devices as the caller has.
struct plan_image {
    uint32_t n = 0, log_n = 0, num_primes = 0;
    bool has_inverse = false;
    int arith_level = 0, narrow_level = 0;
    std::vector<uint64_t> moduli, psi;
    regblock_layout rb, rb_fwd;
    std::vector<prime_consts> consts;
    std::vector<ulonglong2> tw_pairs, itw_pairs, rb_pairs, irb_pairs, fwd_pairs;
};
// arguments as validated by agx_ntt_plan_create*: moduli legal for n, tables [num_primes][n]; psi / itw / ipre may be null
void prepare_plan_image(plan_image& img, uint32_t n, uint32_t num_primes, const uint64_t* moduli, const uint64_t* psi,
                        const uint64_t* tw, const uint64_t* pre, const uint64_t* itw, const uint64_t* ipre);
// a plan on the CURRENT device from a prepared image
int instantiate_plan(agx_ntt_plan** out, const plan_image& img);
void free_plan(agx_ntt_plan* p);

// staging resources of the host-pointer pipeline: three slots of {pinned in, pinned out, device} on three streams
constexpr size_t kStageChunkBytes = (size_t)32 << 20;
struct staging_set {
    static constexpr int kSlots = 3;
    uint64_t *pin_in[kSlots] = {}, *pin_out[kSlots] = {}, *dev[kSlots] = {};
    hipStream_t st[kSlots] = {};
    hipEvent_t done[kSlots] = {};
    size_t bytes = 0;
    hipError_t ensure(size_t want);
    void destroy();
};
// host frames through `plan` (one modulus) with upload, transform and download overlapped; `own` = staging resources the caller
// keeps for this purpose (a group's shard), or null: the per-device pool, or a temporary set when the pool is busy
int host_stream_pipeline(const agx_ntt_plan* plan, const uint64_t* in, const uint64_t* in2, uint64_t* out, uint64_t num_frames,
                         bool inverse, staging_set* own);

}  // namespace agx
