// src/kernel/ntt.cpp -- implementation of the C++ call surface (see ntt.h).
#include "kernel/ntt.h"

#include <hip/hip_runtime.h>

namespace agx {

namespace {

int transform(uint64_t* coeffs, uint32_t n, uint64_t q, uint32_t frames, uint64_t psi, bool inverse) {
    if (!coeffs) return AGX_ERR_NULL_POINTER;
    if (frames == 0) return AGX_OK;
    agx_ntt_plan* plan = nullptr;
    int rc = agx_ntt_plan_create_auto(&plan, n, 1, &q, psi ? &psi : nullptr);
    if (rc != AGX_OK) return rc;
    const size_t bytes = (size_t)n * frames * sizeof(uint64_t);
    uint64_t* d = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&d), bytes);
    if (e == hipSuccess) e = hipMemcpy(d, coeffs, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        rc = inverse ? agx_ntt_inverse(plan, d, d, frames, nullptr) : agx_ntt_forward(plan, d, d, frames, nullptr);
        if (rc == AGX_OK) e = hipMemcpy(coeffs, d, bytes, hipMemcpyDeviceToHost);
    }
    if (d) (void)hipFree(d);
    agx_ntt_plan_destroy(plan);
    if (rc != AGX_OK) return rc;
    return e == hipSuccess ? AGX_OK : (e == hipErrorOutOfMemory ? AGX_ERR_ALLOC : AGX_ERR_HIP);
}

// the same over several devices: a group for the duration of the call, the frames dealt in contiguous blocks (in place: a shard's
// pipeline has staged a chunk before it writes that chunk's results back)
int transform_on(uint64_t* coeffs, uint32_t n, uint64_t q, uint32_t frames, uint64_t psi, bool inverse, const std::vector<int>& devices) {
    if (!coeffs) return AGX_ERR_NULL_POINTER;
    if (frames == 0) return AGX_OK;
    agx_ntt_group* group = nullptr;
    int rc = agx_ntt_group_create_auto(&group, devices.data(), (uint32_t)devices.size(), n, 1, &q, psi ? &psi : nullptr);
    if (rc != AGX_OK) return rc;
    rc = inverse ? agx_ntt_group_inverse_host(group, coeffs, coeffs, frames) : agx_ntt_group_forward_host(group, coeffs, coeffs, coeffs, frames);
    agx_ntt_group_destroy(group);
    return rc;
}

}  // namespace

int ntt(uint64_t* coeffs, uint32_t n, uint64_t q, uint32_t frames, uint64_t psi, const std::vector<int>& devices) {
    return devices.empty() ? transform(coeffs, n, q, frames, psi, false) : transform_on(coeffs, n, q, frames, psi, false, devices);
}
int intt(uint64_t* coeffs, uint32_t n, uint64_t q, uint32_t frames, uint64_t psi, const std::vector<int>& devices) {
    return devices.empty() ? transform(coeffs, n, q, frames, psi, true) : transform_on(coeffs, n, q, frames, psi, true, devices);
}

void ntt_input_kernel(const buffer<uint64_t>& inData, const buffer<uint64_t>& inData2, const buffer<uint64_t>& modulus,
                      const buffer<uint64_t>& twiddleFactors, const buffer<uint64_t>& barrettTwiddleFactors,
                      unsigned int numFrames, queue& q) {
    q.in_ = &inData;
    q.in2_ = &inData2;
    q.mod_ = &modulus;
    q.tw_ = &twiddleFactors;
    q.pre_ = &barrettTwiddleFactors;
    q.frames_in_ = numFrames;
}

void ntt_output_kernel(buffer<uint64_t>& outData, int numFrames, queue& q) {
    q.out_ = &outData;
    q.frames_out_ = numFrames;
}

void fwd_ntt(queue& q) { fwd_ntt_kernel<0>(q); }

int queue::wait() {
    if (!in_ && !out_ && !compute_) return status_ = AGX_OK;
    // the reference's pipes deadlock unless all three kernels were enqueued; here that is an error
    if (!in_ || !out_ || !compute_ || frames_out_ < 0 || (unsigned)frames_out_ != frames_in_) return status_ = AGX_ERR_BAD_ARGUMENT;
    const unsigned frames = frames_in_;
    status_ = AGX_OK;
    if (frames > 0) {
        const size_t n = tw_->size();
        if (n == 0 || mod_->empty() || pre_->size() != n || in_->size() < n * frames || in2_->size() < n * frames ||
            out_->size() < n * frames)
            status_ = AGX_ERR_BAD_ARGUMENT;
        else
            status_ = agx_ntt_forward_host(in_->data(), in2_->data(), mod_->data(), tw_->data(), pre_->data(), out_->data(),
                                           (uint32_t)n, frames);
    }
    in_ = in2_ = mod_ = tw_ = pre_ = nullptr;
    out_ = nullptr;
    compute_ = false;
    return status_;
}

}  // namespace agx
