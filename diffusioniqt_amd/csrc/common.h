// Shared helpers for the libdiqt_hip.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdarg.h>
#include <stdint.h>
#include "../../include/diqt.h"

namespace diqt {

void set_error(const char* fmt, ...);
// launch census (lib.cpp): every launch passes its tag here -- remembered per thread (diqt_get_last_launch) and, while
// diqt_census_enable(1), counted per tag (diqt_census_count): the parity suite asserts which kernels a network really dispatched
void census_note(const char* what);

inline int check_launch(const char* what) {
    census_note(what);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return DIQT_E_LAUNCH;
    }
    return DIQT_OK;
}

#define DIQT_REQUIRE(cond, code, ...)            \
    do {                                         \
        if (!(cond)) {                           \
            diqt::set_error(__VA_ARGS__);        \
            return (code);                       \
        }                                        \
    } while (0)

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

constexpr int kNumXcd = 8;   // MI355X: 8 XCDs, blocks are dealt round-robin over them

// Bijective remap of a linear block id so that each XCD receives a contiguous run of logical tiles
// (neighbouring tiles share halo voxels / weight panels -> same L2).  Speed only, never correctness.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nwg) {
    const unsigned q = nwg / kNumXcd, r = nwg % kNumXcd;
    const unsigned xcd = bid % kNumXcd, idx = bid / kNumXcd;
    const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

__device__ __forceinline__ float act_fwd(float x, int act) {
    switch (act) {
        case DIQT_ACT_MISH: {
            // x * tanh(softplus(x)); with n = e^x: tanh(log(1+n)) = (n^2+2n)/(n^2+2n+2)
            // (hardware reciprocal, 1 ulp, as conv_fwd9_kernel's fused GroupNorm-apply has always computed it: an IEEE division is ~12
            // instructions, and the GroupNorm-apply passes writing 16-bit tensors are bound by this arithmetic, not by their bytes)
            if (x > 20.f) return x;
            const float n = __expf(x);
            const float m = n * (n + 2.f);
            return x * (m * __builtin_amdgcn_rcpf(m + 2.f));
        }
        case DIQT_ACT_SILU: return x * __builtin_amdgcn_rcpf(1.f + __expf(-x));
        case DIQT_ACT_GELU: return 0.5f * x * (1.f + erff(x * 0.70710678118654752f));
        case DIQT_ACT_RELU: return x > 0.f ? x : 0.f;
        case DIQT_ACT_SIGMOID: return 1.f / (1.f + __expf(-x));
        default: return x;
    }
}

// d act(x) / dx
__device__ __forceinline__ float act_grad(float x, int act) {
    switch (act) {
        case DIQT_ACT_MISH: {
            if (x > 20.f) return 1.f;
            const float n = __expf(x);
            const float m = n * (n + 2.f);
            const float t = m / (m + 2.f);                 // tanh(softplus(x))
            const float sg = n / (1.f + n);                // sigmoid(x)
            return t + x * sg * (1.f - t * t);
        }
        case DIQT_ACT_SILU: {
            const float s = 1.f / (1.f + __expf(-x));
            return s * (1.f + x * (1.f - s));
        }
        case DIQT_ACT_GELU: {
            const float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752f));
            const float pdf = 0.39894228040143267f * __expf(-0.5f * x * x);
            return cdf + x * pdf;
        }
        case DIQT_ACT_RELU: return x > 0.f ? 1.f : 0.f;
        case DIQT_ACT_SIGMOID: {
            const float s = 1.f / (1.f + __expf(-x));
            return s * (1.f - s);
        }
        default: return 1.f;
    }
}

// the same with hardware reciprocals (1 ulp) instead of IEEE divisions (~12 instructions each): the GroupNorm backward of a low-precision
// training step, where x and / or dy hold 16-bit values anyway, is bound by these instructions (both of its passes evaluate it per element)
__device__ __forceinline__ float act_grad_fast(float x, int act) {
    switch (act) {
        case DIQT_ACT_MISH: {
            if (x > 20.f) return 1.f;
            const float n = __expf(x);                     // <= 4.9e8
            const float m = n * (n + 2.f);                 // <= 2.4e17
            const float r = __builtin_amdgcn_rcpf((m + 2.f) * (1.f + n));      // one reciprocal for both quotients (<= 1.2e26: in range)
            const float t = m * (1.f + n) * r;             // tanh(softplus(x)) = m / (m + 2)
            const float sg = n * (m + 2.f) * r;            // sigmoid(x) = n / (1 + n)
            return t + x * sg * (1.f - t * t);
        }
        case DIQT_ACT_SILU: {
            const float s = __builtin_amdgcn_rcpf(1.f + __expf(-x));
            return s * (1.f + x * (1.f - s));
        }
        default: return act_grad(x, act);
    }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// block-wide sum for blockDim.x == 256 (4 waves); result valid in every thread
__device__ __forceinline__ float block_sum256(float v, float* sh /* >= 4 floats */) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

inline unsigned grid_for(size_t n, unsigned block, unsigned cap = 2048u * 4u) {
    size_t g = (n + block - 1) / block;
    if (g > cap) g = cap;
    if (g == 0) g = 1;
    return (unsigned)g;
}

}  // namespace diqt
