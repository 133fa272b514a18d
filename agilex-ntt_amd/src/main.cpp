// src/main.cpp -- harness for the GPU forward-NTT path, the counterpart of the reference's
// smoke driver (src/main.cpp:14-89 there) with real tables and known answers instead of
// placeholders.  `make run` builds and runs it.
//
// Checks, for n = 32 (30- and 60-bit q), n = 512 (60-bit), n = 1024 (30-bit q), n = 4096 / 16384 / 32768 (60-bit q) and n = 16384 under the reference's own 17-bit modulus 65537:
//   NTT(delta_0) = (1,...,1);  NTT(X)[bitrev(k)] = psi^(2k+1);  INTT(NTT(x)) = x on random x;
// the reference's operand pairing with inData2 != inData (src/kernel/ntt.cpp:584-590); and the convolution theorem against the
// schoolbook product mod X^n + 1 (n = 64 and 1024).
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "kernel/ntt.h"

using agx::buffer;

static uint64_t mulmod(uint64_t a, uint64_t b, uint64_t q) { return (uint64_t)((unsigned __int128)a * b % q); }
static uint32_t bitrev(uint32_t x, int bits) {
    uint32_t r = 0;
    for (int i = 0; i < bits; ++i) { r = (r << 1) | (x & 1); x >>= 1; }
    return r;
}

static int run_case(uint32_t n, uint32_t bits) {
    int lg = 0;
    while ((1u << lg) < n) ++lg;
    uint64_t q = 0, psi = 0;
    if (agx_ntt_find_primes(bits, n, 1, &q) || agx_ntt_min_root(q, n, &psi)) { std::printf("n=%u: prime/root search failed\n", n); return 1; }
    const unsigned numFrames = 3;
    buffer<uint64_t> inData(numFrames * n, 0), modulus(1, q), tw(n), pre(n), outData(numFrames * n, 0);
    if (agx_ntt_make_tables(q, psi, n, tw.data(), pre.data())) return 1;
    inData[0] = 1;          // frame 0: delta_0
    inData[n + 1] = 1;      // frame 1: X
    uint64_t s = 0x1234567u + n;
    for (uint32_t i = 0; i < n; ++i) { s = s * 6364136223846793005ull + 1442695040888963407ull; inData[2 * n + i] = (s >> 3) % q; }

    agx::queue qu;
    auto t0 = std::chrono::steady_clock::now();
    agx::ntt_input_kernel(inData, inData, modulus, tw, pre, numFrames, qu);
    agx::fwd_ntt_kernel<0>(qu);
    agx::ntt_output_kernel(outData, numFrames, qu);
    int rc = qu.wait();
    double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (rc) { std::printf("n=%u: forward failed: %s\n", n, agx_ntt_strerror(rc)); return 1; }
    // the same three calls again (timing the third round): the one-shot path keeps its plan while (n, modulus, tables) repeat
    double ms2 = 0;
    for (int rep = 0; rep < 2 && !rc; ++rep) {
        t0 = std::chrono::steady_clock::now();
        agx::ntt_input_kernel(inData, inData, modulus, tw, pre, numFrames, qu);
        agx::fwd_ntt_kernel<0>(qu);
        agx::ntt_output_kernel(outData, numFrames, qu);
        rc = qu.wait();
        ms2 = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    if (rc) { std::printf("n=%u: repeated forward failed: %s\n", n, agx_ntt_strerror(rc)); return 1; }

    int bad = 0;
    for (uint32_t i = 0; i < n; ++i) bad += outData[i] != 1;
    uint64_t p = psi, psi2 = mulmod(psi, psi, q);
    for (uint32_t k = 0; k < n; ++k) { bad += outData[n + bitrev(k, lg)] != p; p = mulmod(p, psi2, q); }
    std::vector<uint64_t> rt(outData.begin() + 2 * n, outData.end());
    rc = agx::intt(rt.data(), n, q, 1, psi);
    if (rc) { std::printf("n=%u: inverse failed: %s\n", n, agx_ntt_strerror(rc)); return 1; }
    for (uint32_t i = 0; i < n; ++i) bad += rt[i] != inData[2 * n + i];
    std::printf("n=%5u q=%llu psi=%llu frames=%u  one-shot %.2f ms, repeated %.3f ms  mismatches=%d  %s\n", n, (unsigned long long)q,
                (unsigned long long)psi, numFrames, ms, ms2, bad, bad ? "FAIL" : "PASS");
    return bad != 0;
}

// the reference's operand pairing (src/kernel/ntt.cpp:584-590): the lower half of every frame comes from inData, the upper half
// from inData2.  Two DIFFERENT buffers, each holding garbage in the half that must not be read: the result must equal the
// transform of the stitched frame (computed through agx::ntt on a host copy).
static int run_pairing_case(uint32_t n, uint32_t bits) {
    uint64_t q = 0, psi = 0;
    if (agx_ntt_find_primes(bits, n, 1, &q) || agx_ntt_min_root(q, n, &psi)) return 1;
    const unsigned numFrames = 2;
    buffer<uint64_t> a(numFrames * n), b(numFrames * n), modulus(1, q), tw(n), pre(n), outData(numFrames * n, 0), stitched(numFrames * n);
    if (agx_ntt_make_tables(q, psi, n, tw.data(), pre.data())) return 1;
    uint64_t s = 0xabcdefu + n;
    auto next = [&] { s = s * 6364136223846793005ull + 1442695040888963407ull; return (s >> 3) % q; };
    for (unsigned f = 0; f < numFrames; ++f)
        for (uint32_t i = 0; i < n; ++i) {
            const bool lower = i < n / 2;
            const uint64_t v = next();
            stitched[f * n + i] = v;
            a[f * n + i] = lower ? v : ~0ull;      // inData's upper half and inData2's lower half are never read
            b[f * n + i] = lower ? ~0ull : v;
        }
    agx::queue qu;
    agx::ntt_input_kernel(a, b, modulus, tw, pre, numFrames, qu);
    agx::fwd_ntt_kernel<0>(qu);
    agx::ntt_output_kernel(outData, numFrames, qu);
    int rc = qu.wait();
    if (!rc) rc = agx::ntt(stitched.data(), n, q, numFrames, psi);
    if (rc) { std::printf("n=%u pairing: failed: %s\n", n, agx_ntt_strerror(rc)); return 1; }
    int bad = 0;
    for (size_t i = 0; i < stitched.size(); ++i) bad += outData[i] != stitched[i];
    std::printf("n=%5u q=%llu in2 != in (lower half from inData, upper from inData2)  mismatches=%d  %s\n", n, (unsigned long long)q, bad, bad ? "FAIL" : "PASS");
    return bad != 0;
}

// the convolution theorem against the definition: c = INTT(NTT(a) o NTT(b)) must be the schoolbook product of a and b in
// Z_q[X]/(X^n + 1), computed here coefficient by coefficient (n = 64: 4,096 multiply-adds per polynomial)
static int run_schoolbook_case(uint32_t n, uint32_t bits) {
    uint64_t q = 0, psi = 0;
    if (agx_ntt_find_primes(bits, n, 1, &q) || agx_ntt_min_root(q, n, &psi)) return 1;
    std::vector<uint64_t> a(n), b(n), want(n, 0);
    uint64_t s = 0x5eed5eedu + n + bits;
    auto next = [&] { s = s * 6364136223846793005ull + 1442695040888963407ull; return (s >> 3) % q; };
    for (uint32_t i = 0; i < n; ++i) { a[i] = next(); b[i] = next(); }
    for (uint32_t i = 0; i < n; ++i)
        for (uint32_t j = 0; j < n; ++j) {
            const uint64_t p = mulmod(a[i], b[j], q);
            const uint32_t k = (i + j) % n;
            want[k] = (i + j < n) ? (want[k] + p) % q : (want[k] + q - p) % q;      // X^n = -1
        }
    std::vector<uint64_t> fa(a), fb(b);
    int rc = agx::ntt(fa.data(), n, q, 1, psi);
    if (!rc) rc = agx::ntt(fb.data(), n, q, 1, psi);
    for (uint32_t i = 0; i < n; ++i) fa[i] = mulmod(fa[i], fb[i], q);
    if (!rc) rc = agx::intt(fa.data(), n, q, 1, psi);
    if (rc) { std::printf("n=%u schoolbook: failed: %s\n", n, agx_ntt_strerror(rc)); return 1; }
    int bad = 0;
    for (uint32_t i = 0; i < n; ++i) bad += fa[i] != want[i];
    std::printf("n=%5u q=%llu INTT(NTT(a) o NTT(b)) vs schoolbook product mod X^n+1  mismatches=%d  %s\n", n, (unsigned long long)q, bad, bad ? "FAIL" : "PASS");
    return bad != 0;
}

// ---- scaling mode (--gpus N / --devices a,b,...): the library-level multi-GPU driver (agx_ntt_group_*, include/agx_ntt.h section 5) ----
// Host frames dealt to the listed devices (agx::ntt with a device list) against the single-device result, then the device-pointer
// form timed: every shard transforms its own resident batch from its own host thread on its own stream, HIP events per shard.
// Prints per-device and aggregate NTT/s and the fraction of the per-device HBM roofline (16n bytes per NTT at 8 TB/s).
static int group_parity(const std::vector<int>& devices, uint32_t n, uint32_t bits, uint32_t frames) {
    uint64_t q = 0, psi = 0;
    if (agx_ntt_find_primes(bits, n, 1, &q) || agx_ntt_min_root(q, n, &psi)) return 1;
    std::vector<uint64_t> x((size_t)frames * n), one, many;
    uint64_t s = 0x9e3779b9u + n + frames;
    for (auto& v : x) { s = s * 6364136223846793005ull + 1442695040888963407ull; v = (s >> 3) % q; }
    one = x;
    many = x;
    int rc = agx::ntt(one.data(), n, q, frames, psi);
    if (!rc) rc = agx::ntt(many.data(), n, q, frames, psi, devices);
    if (rc) { std::printf("n=%u group parity: failed: %s\n", n, agx_ntt_strerror(rc)); return 1; }
    size_t bad = 0;
    for (size_t i = 0; i < x.size(); ++i) bad += one[i] != many[i];
    rc = agx::intt(many.data(), n, q, frames, psi, devices);
    if (rc) { std::printf("n=%u group inverse: failed: %s\n", n, agx_ntt_strerror(rc)); return 1; }
    for (size_t i = 0; i < x.size(); ++i) bad += many[i] != x[i];
    std::printf("n=%5u q=%llu %u frames over %zu shards (contiguous blocks) == one device, inverse round trip  mismatches=%zu  %s\n", n,
                (unsigned long long)q, frames, devices.size(), bad, bad ? "FAIL" : "PASS");
    return bad != 0;
}

static int group_bench(const std::vector<int>& devices, uint32_t n, uint32_t primes, uint64_t batch, int slabs, int steps, int warmup, const char* label) {
    const uint32_t shards = (uint32_t)devices.size();
    std::vector<uint64_t> qs(primes);
    if (agx_ntt_find_primes(60, n, primes, qs.data())) return 1;
    agx_ntt_group* g = nullptr;
    int rc = agx_ntt_group_create_auto(&g, devices.data(), shards, n, primes, qs.data(), nullptr);
    if (rc) { std::printf("%s: group create failed: %s\n", label, agx_ntt_strerror(rc)); return 1; }
    const size_t per = (size_t)primes * batch * n;
    std::vector<std::vector<uint64_t*>> buf(shards, std::vector<uint64_t*>(slabs, nullptr));
    std::vector<hipEvent_t> e0(shards), e1(shards);
    std::vector<void*> streams(shards);
    hipError_t he = hipSuccess;
    for (uint32_t i = 0; i < shards && he == hipSuccess && !rc; ++i) {
        agx_ntt_plan* plan = nullptr;
        agx_ntt_group_shard(g, i, nullptr, &plan, &streams[i]);
        he = hipSetDevice(devices[i]);
        for (int k = 0; k < slabs && he == hipSuccess && !rc; ++k) {
            he = hipMalloc(reinterpret_cast<void**>(&buf[i][k]), per * sizeof(uint64_t));
            // frame (p, b) of shard i, slab k is polynomial (k shards + i) batch + b of the global batch: the same data whatever the sharding
            if (he == hipSuccess) rc = agx_ntt_fill_synthetic(plan, buf[i][k], batch, ((uint64_t)k * shards + i) * batch, 42, streams[i]);
        }
        if (he == hipSuccess) he = hipEventCreate(&e0[i]);
        if (he == hipSuccess) he = hipEventCreate(&e1[i]);
    }
    std::vector<const uint64_t*> in(shards);
    std::vector<uint64_t*> out(shards);
    std::vector<uint64_t> batches(shards, batch);
    auto step = [&](int it) {
        for (uint32_t i = 0; i < shards; ++i) in[i] = out[i] = buf[i][it % slabs];
        return agx_ntt_group_forward(g, in.data(), out.data(), batches.data());      // in place, every shard on its own thread and stream
    };
    for (int it = 0; it < warmup && !rc && he == hipSuccess; ++it) rc = step(it);
    if (!rc && he == hipSuccess) rc = agx_ntt_group_synchronize(g);
    for (uint32_t i = 0; i < shards && he == hipSuccess; ++i) { he = hipSetDevice(devices[i]); if (he == hipSuccess) he = hipEventRecord(e0[i], static_cast<hipStream_t>(streams[i])); }
    const auto t0 = std::chrono::steady_clock::now();
    for (int it = 0; it < steps && !rc && he == hipSuccess; ++it) rc = step(warmup + it);
    for (uint32_t i = 0; i < shards && he == hipSuccess; ++i) { he = hipSetDevice(devices[i]); if (he == hipSuccess) he = hipEventRecord(e1[i], static_cast<hipStream_t>(streams[i])); }
    if (!rc && he == hipSuccess) rc = agx_ntt_group_synchronize(g);
    const double wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (rc || he != hipSuccess) {
        std::printf("%s: failed: %s / %s\n", label, agx_ntt_strerror(rc), hipGetErrorString(he));
    } else {
        const double ntts = (double)primes * (double)batch, bytes = 16.0 * n;
        std::printf("%s: n=%u, %u primes, batch %llu per shard, %u shard(s), %d timed steps\n", label, n, primes, (unsigned long long)batch, shards, steps);
        for (uint32_t i = 0; i < shards; ++i) {
            float ms = 0;
            (void)hipEventElapsedTime(&ms, e0[i], e1[i]);
            const double rate = ntts * steps / (ms * 1e-3);
            std::printf("  shard %u (device %d): %.4f ms per step (HIP events), %.2f M NTT/s, %.1f %% of 8 TB/s\n", i, devices[i], ms / steps, rate / 1e6, rate * bytes / 8e12 * 100);
        }
        const double agg = ntts * shards * steps / (wall_ms * 1e-3);
        std::printf("  aggregate: %.2f M NTT/s over %.3f ms wall (%.2f M per shard)\n", agg / 1e6, wall_ms, agg / shards / 1e6);
    }
    for (uint32_t i = 0; i < shards; ++i) {
        (void)hipSetDevice(devices[i]);
        for (uint64_t* p : buf[i]) if (p) (void)hipFree(p);
    }
    agx_ntt_group_destroy(g);
    return rc || he != hipSuccess;
}

static int scaling_mode(const std::vector<int>& devices, bool small, int steps) {
    int fail = group_parity(devices, 4096, 60, 1101) | group_parity(devices, 16384, 60, 301) | group_parity(devices, 1024, 30, 3);
    // BASELINE configs[2] (n=4096, 4 primes, batch 4096 per GPU) and configs[3]'s per-GPU slice (n=16384, 8 primes, batch 8192: 8 GiB per shard);
    // --small divides the batches by 8 so that several shards fit one GPU
    const uint64_t div = small ? 8 : 1;
    fail |= group_bench(devices, 4096, 4, 4096 / div, 4, steps, 10, "configs[2]");
    fail |= group_bench(devices, 16384, 8, 8192 / div, 1, steps > 20 ? 20 : steps, 2, "configs[3] slice");
    std::printf(fail ? "SCALING FAILED\n" : "SCALING PASSED\n");
    return fail;
}

int main(int argc, char** argv) {
    int ndev = 0;
    agx_ntt_device_count(&ndev);
    if (ndev == 0) { std::printf("no HIP device: the forward path needs an MI355X\n"); return 2; }
    std::vector<int> devices;
    bool small = false;
    int steps = 100;
    for (int i = 1; i < argc; ++i) {
        if (!std::strcmp(argv[i], "--gpus") && i + 1 < argc) {
            const int g = std::atoi(argv[++i]);
            for (int d = 0; d < g; ++d) devices.push_back(d);
        } else if (!std::strcmp(argv[i], "--devices") && i + 1 < argc) {
            for (const char* p = argv[++i]; *p;) { devices.push_back(std::atoi(p)); while (*p && *p != ',') ++p; if (*p == ',') ++p; }
        } else if (!std::strcmp(argv[i], "--small")) {
            small = true;
        } else if (!std::strcmp(argv[i], "--steps") && i + 1 < argc) {
            steps = std::atoi(argv[++i]);
        } else {
            std::printf("usage: ntt_harness [--gpus N | --devices a,b,...] [--small] [--steps K]\n");
            return 2;
        }
    }
    if (!devices.empty()) return scaling_mode(devices, small, steps);
    // 17 bits: the modulus class of the reference's own smoke driver (65537, src/main.cpp:55) -> the 32-bit arithmetic kernels
    // n = 32 is the smallest size of the reference's table (include/kernel/ntt.h:11-12): the wave-packed kernels (several frames per wave)
    int fail = run_case(32, 30) | run_case(32, 60) | run_case(512, 60) | run_case(1024, 30) | run_case(4096, 60) | run_case(16384, 60) | run_case(32768, 60) | run_case(16384, 17);
    fail |= run_pairing_case(1024, 30) | run_pairing_case(16384, 60);
    fail |= run_schoolbook_case(64, 30) | run_schoolbook_case(64, 60) | run_schoolbook_case(1024, 30);
    std::printf(fail ? "HARNESS FAILED\n" : "HARNESS PASSED\n");
    return fail;
}
