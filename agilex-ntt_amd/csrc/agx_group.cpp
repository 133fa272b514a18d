// agx_group.cpp -- the multi-GPU driver behind the C ABI (include/agx_ntt.h section 5).
//
// The reference deals the frames of one call to its replicated compute units inside the call: ntt_input_kernel computes the minibatch
// of compute unit i as floor(F / C) + [i < F mod C] (src/kernel/ntt.cpp:526-536), hands frame b to unit b % C (:579-582) and
// ntt_output_kernel collects them in the same order (:622-625); the units never talk to each other.  A group is that scheme over whole
// GPUs: one SHARD per listed device (a device may be listed more than once), each with its own plan (tables prepared once on the host,
// uploaded to every device), its own stream, its own staging buffers and its own HOST THREAD, which sets its device once and then
// serves the group's calls.  Frames are dealt in contiguous blocks of the reference's minibatch sizes (a block is one memcpy and one
// launch; round-robin dealing would scatter both).  Nothing is exchanged between shards: no collective, no RCCL, no peer access.
#include <condition_variable>
#include <functional>
#include <memory>

#include "plan_internal.hpp"

using namespace agx;

namespace {

struct shard {
    int device = -1;
    agx_ntt_plan* plan = nullptr;
    hipStream_t stream = nullptr;
    staging_set stage;      // host-pointer calls: allocated on first use (192 MiB pinned + 96 MiB device), kept for the group's life
    // one worker thread per shard: current device set once, then jobs one at a time
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::function<int()> job;
    bool has_job = false, done = false, quit = false;
    int result = AGX_OK, hip_error = 0;      // hip_error: this worker's agx_ntt_last_hip_error() after the job (thread-local: the caller cannot see it otherwise)

    void loop() {
        const bool dev_ok = hipSetDevice(device) == hipSuccess;
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv.wait(lk, [&] { return has_job || quit; });
            if (quit) break;
            std::function<int()> j = std::move(job);
            has_job = false;
            lk.unlock();
            int rc = dev_ok ? guarded(j) : AGX_ERR_HIP;
            lk.lock();
            result = rc;
            hip_error = last_hip_error_slot();
            done = true;
            cv.notify_all();
        }
        lk.unlock();
        // the shard's device resources go on its own thread, with its device current
        if (dev_ok) {
            stage.destroy();
            if (stream) (void)hipStreamDestroy(stream);
            free_plan(plan);
        }
        plan = nullptr;
        stream = nullptr;
    }
    void post(std::function<int()> j) {
        std::lock_guard<std::mutex> lk(mu);
        job = std::move(j);
        has_job = true;
        done = false;
        cv.notify_all();
    }
    int wait() {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return done; });
        return result;
    }
};

}  // namespace

struct agx_ntt_group {
    uint32_t n = 0, num_primes = 0;
    bool has_inverse = false;
    std::vector<std::unique_ptr<shard>> shards;
    std::mutex call_mu;      // one call at a time per group (each worker serves one job at a time)
    ~agx_ntt_group() {
        for (auto& s : shards) {
            if (!s->th.joinable()) continue;
            {
                std::lock_guard<std::mutex> lk(s->mu);
                s->quit = true;
                s->cv.notify_all();
            }
            s->th.join();
        }
    }
};

namespace {

// fn(shard index) on every shard's own thread, all at once; the status of the lowest-numbered failing shard
template <class F>
int on_every_shard(const agx_ntt_group* cg, F&& fn) {
    agx_ntt_group* g = const_cast<agx_ntt_group*>(cg);
    std::lock_guard<std::mutex> call(g->call_mu);
    for (size_t i = 0; i < g->shards.size(); ++i) g->shards[i]->post([&fn, i] { return fn((uint32_t)i); });
    int rc = AGX_OK;
    for (size_t i = 0; i < g->shards.size(); ++i) {
        const int r = g->shards[i]->wait();
        if (rc == AGX_OK && r != AGX_OK) {
            rc = r;
            last_hip_error_slot() = g->shards[i]->hip_error;      // agx_ntt_last_hip_error() on the caller's thread names the failing shard's HIP error
        }
    }
    return rc;
}

void shard_range(uint64_t num_frames, uint32_t num_shards, uint32_t index, uint64_t* first, uint64_t* count) {
    // the reference's minibatch sizes (src/kernel/ntt.cpp:526-536): floor(F / C) frames each, one more for the first F mod C units
    const uint64_t base = num_frames / num_shards, extra = num_frames % num_shards;
    *count = base + (index < extra ? 1 : 0);
    *first = (uint64_t)index * base + (index < extra ? index : extra);
}

int check_devices(const int* devices, uint32_t num_devices) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return AGX_ERR_NO_DEVICE;
    for (uint32_t i = 0; i < num_devices; ++i)
        if (devices[i] < 0 || devices[i] >= ndev) return AGX_ERR_BAD_ARGUMENT;
    return AGX_OK;
}

int create_group(agx_ntt_group** out, const int* devices, uint32_t num_devices, const plan_image& img) {
    std::unique_ptr<agx_ntt_group> g(new agx_ntt_group);
    g->n = img.n;
    g->num_primes = img.num_primes;
    g->has_inverse = img.has_inverse;
    for (uint32_t i = 0; i < num_devices; ++i) {
        g->shards.emplace_back(new shard);
        shard* s = g->shards.back().get();
        s->device = devices[i];
        s->th = std::thread([s] { s->loop(); });      // a throw here unwinds through ~agx_ntt_group, which joins the threads started so far
    }
    const int rc = on_every_shard(g.get(), [&](uint32_t i) -> int {
        shard* s = g->shards[i].get();
        int prc = instantiate_plan(&s->plan, img);      // on this thread's device
        if (prc != AGX_OK) return prc;
        AGX_HIP(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
        return AGX_OK;
    });
    if (rc != AGX_OK) return rc;      // ~agx_ntt_group releases whatever the shards got
    *out = g.release();
    return AGX_OK;
}

int check_group_size(uint32_t n, uint32_t num_primes, uint32_t num_devices) {
    if (n < AGX_NTT_MIN_N || n > AGX_NTT_MAX_N || (n & (n - 1))) return AGX_ERR_BAD_SIZE;
    if (num_primes == 0 || num_primes > 65535 || num_devices == 0 || num_devices > 1024) return AGX_ERR_BAD_ARGUMENT;
    return AGX_OK;
}

}  // namespace

extern "C" {

int agx_ntt_shard_range(uint64_t num_frames, uint32_t num_shards, uint32_t index, uint64_t* first, uint64_t* count) {
    if (!first || !count) return AGX_ERR_NULL_POINTER;
    if (num_shards == 0 || index >= num_shards) return AGX_ERR_BAD_ARGUMENT;
    shard_range(num_frames, num_shards, index, first, count);
    return AGX_OK;
}

int agx_ntt_group_create(agx_ntt_group** group, const int* devices, uint32_t num_devices, uint32_t n, uint32_t num_primes,
                         const uint64_t* moduli, const uint64_t* twiddles, const uint64_t* precons,
                         const uint64_t* inv_twiddles, const uint64_t* inv_precons) {
    if (!group || !devices || !moduli || !twiddles || !precons) return AGX_ERR_NULL_POINTER;
    *group = nullptr;
    if ((inv_twiddles == nullptr) != (inv_precons == nullptr)) return AGX_ERR_NULL_POINTER;
    int rc = check_group_size(n, num_primes, num_devices);
    if (rc) return rc;
    if ((rc = check_devices(devices, num_devices))) return rc;
    // the modulus rules of agx_ntt_plan_create (src/kernel/ntt.cpp:302-369: 4q < 2^64, 2n | q - 1)
    for (uint32_t k = 0; k < num_primes; ++k) {
        const uint64_t q = moduli[k];
        if (q < 3 || (q & 1) == 0 || q >= (1ull << 62) || (q - 1) % (2ull * n)) return AGX_ERR_BAD_MODULUS;
    }
    return guarded([&]() -> int {
        plan_image img;
        prepare_plan_image(img, n, num_primes, moduli, nullptr, twiddles, precons, inv_twiddles, inv_precons);      // once, on the host
        return create_group(group, devices, num_devices, img);
    });
}

int agx_ntt_group_create_auto(agx_ntt_group** group, const int* devices, uint32_t num_devices, uint32_t n, uint32_t num_primes,
                              const uint64_t* moduli, const uint64_t* psi) {
    if (!group || !devices || !moduli) return AGX_ERR_NULL_POINTER;
    *group = nullptr;
    int rc = check_group_size(n, num_primes, num_devices);
    if (rc) return rc;
    if ((rc = check_devices(devices, num_devices))) return rc;
    return guarded([&]() -> int {
        // tables generated once: through the public host-math entry points, which validate (q, psi) exactly as agx_ntt_plan_create_auto does
        std::vector<uint64_t> roots(num_primes), tw((size_t)num_primes * n), pre(tw.size()), itw(tw.size()), ipre(tw.size());
        for (uint32_t k = 0; k < num_primes; ++k) {
            roots[k] = psi ? psi[k] : 0;
            int mrc = AGX_OK;
            if (!psi && (mrc = agx_ntt_min_root(moduli[k], n, &roots[k]))) return mrc;
            if ((mrc = agx_ntt_make_tables(moduli[k], roots[k], n, &tw[(size_t)k * n], &pre[(size_t)k * n]))) return mrc;
            if ((mrc = agx_ntt_make_inverse_tables(moduli[k], roots[k], n, &itw[(size_t)k * n], &ipre[(size_t)k * n]))) return mrc;
        }
        plan_image img;
        prepare_plan_image(img, n, num_primes, moduli, roots.data(), tw.data(), pre.data(), itw.data(), ipre.data());
        return create_group(group, devices, num_devices, img);
    });
}

int agx_ntt_group_destroy(agx_ntt_group* group) {
    delete group;      // joins the workers; each frees its own shard on its own device
    return AGX_OK;
}

int agx_ntt_group_info(const agx_ntt_group* group, uint32_t* num_shards, uint32_t* n, uint32_t* num_primes) {
    if (!group) return AGX_ERR_NULL_POINTER;
    if (num_shards) *num_shards = (uint32_t)group->shards.size();
    if (n) *n = group->n;
    if (num_primes) *num_primes = group->num_primes;
    return AGX_OK;
}

int agx_ntt_group_shard(const agx_ntt_group* group, uint32_t index, int* device, agx_ntt_plan** plan, void** stream) {
    if (!group) return AGX_ERR_NULL_POINTER;
    if (index >= group->shards.size()) return AGX_ERR_BAD_ARGUMENT;
    const shard* s = group->shards[index].get();
    if (device) *device = s->device;
    if (plan) *plan = s->plan;
    if (stream) *stream = s->stream;
    return AGX_OK;
}

// ---- host frames: contiguous blocks, one host thread + one streaming pipeline per shard --------------------------------------
static int group_host(const agx_ntt_group* group, const uint64_t* in, const uint64_t* in2, uint64_t* out, uint64_t num_frames, bool inverse) {
    if (!group || !in || !in2 || !out) return AGX_ERR_NULL_POINTER;
    if (group->num_primes != 1) return AGX_ERR_BAD_ARGUMENT;      // one modulus per call, as the reference (src/kernel/ntt.cpp:143-144)
    if (inverse && !group->has_inverse) return AGX_ERR_NO_INVERSE;
    if (num_frames == 0) return AGX_OK;
    const uint32_t shards = (uint32_t)group->shards.size();
    const size_t n = group->n;
    return guarded([&] {
        return on_every_shard(group, [&](uint32_t i) -> int {
            uint64_t first = 0, count = 0;
            shard_range(num_frames, shards, i, &first, &count);
            if (count == 0) return AGX_OK;      // more shards than frames
            shard* s = group->shards[i].get();
            return host_stream_pipeline(s->plan, in + first * n, in2 + first * n, out + first * n, count, inverse, &s->stage);
        });
    });
}

int agx_ntt_group_forward_host(const agx_ntt_group* group, const uint64_t* in, const uint64_t* in2, uint64_t* out, uint64_t num_frames) {
    return group_host(group, in, in2, out, num_frames, false);
}

int agx_ntt_group_inverse_host(const agx_ntt_group* group, const uint64_t* in, uint64_t* out, uint64_t num_frames) {
    return group_host(group, in, in, out, num_frames, true);
}

// ---- device pointers: one pointer and one batch per shard, every shard launched from its own thread on its own stream ---------
enum group_op { OP_FORWARD, OP_INVERSE, OP_POLYMUL };

static int group_device(const agx_ntt_group* group, group_op op, const uint64_t* const* a, const uint64_t* const* b, uint64_t* const* c,
                        uint64_t* const* scratch, const uint64_t* batch) {
    if (!group || !a || !c || !batch || (op == OP_POLYMUL && !b)) return AGX_ERR_NULL_POINTER;
    return guarded([&] {
        return on_every_shard(group, [&](uint32_t i) -> int {
            shard* s = group->shards[i].get();
            if (batch[i] == 0) return AGX_OK;
            switch (op) {
                case OP_FORWARD: return agx_ntt_forward(s->plan, a[i], c[i], batch[i], s->stream);
                case OP_INVERSE: return agx_ntt_inverse(s->plan, a[i], c[i], batch[i], s->stream);
                default: return agx_ntt_polymul(s->plan, a[i], b[i], c[i], scratch ? scratch[i] : nullptr, batch[i], s->stream);
            }
        });
    });
}

int agx_ntt_group_forward(const agx_ntt_group* group, const uint64_t* const* d_in, uint64_t* const* d_out, const uint64_t* batch) {
    return group_device(group, OP_FORWARD, d_in, nullptr, d_out, nullptr, batch);
}

int agx_ntt_group_inverse(const agx_ntt_group* group, const uint64_t* const* d_in, uint64_t* const* d_out, const uint64_t* batch) {
    return group_device(group, OP_INVERSE, d_in, nullptr, d_out, nullptr, batch);
}

int agx_ntt_group_polymul(const agx_ntt_group* group, const uint64_t* const* d_a, const uint64_t* const* d_b, uint64_t* const* d_c,
                          uint64_t* const* d_scratch, const uint64_t* batch) {
    return group_device(group, OP_POLYMUL, d_a, d_b, d_c, d_scratch, batch);
}

int agx_ntt_group_synchronize(const agx_ntt_group* group) {
    if (!group) return AGX_ERR_NULL_POINTER;
    return guarded([&] {
        return on_every_shard(group, [&](uint32_t i) -> int {
            AGX_HIP(hipStreamSynchronize(group->shards[i]->stream));
            return AGX_OK;
        });
    });
}

}  // extern "C"
