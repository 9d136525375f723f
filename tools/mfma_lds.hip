// Inner-loop study: MFMA f32 32x32x2 fed from LDS.  Variants:
//  0: pure MFMA (registers)           1: per k-step {1+NT ds_read_b32 -> NT MFMA}, compiler-scheduled
//  2: same, reads of step s+1 issued before the MFMAs of step s (explicit register double buffer + sched_barrier)
//  3: fwd-conv style: ds_read_b128 x3 per 8 MFMA (2 acc), prefetched one group ahead
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int NT = 7;
__device__ int g_random;
template <int VAR>
__global__ __launch_bounds__(256, 2) void k(float* out, int iters) {
    extern __shared__ float lds[];
    // DATA=1: full-range pseudo-random operands (hash) -> realistic toggle rate; DATA=0: smooth small values
    for (int i = threadIdx.x; i < 16384; i += 256) {
        unsigned u = (unsigned)i * 2654435761u + blockIdx.x * 40503u; u ^= u >> 15; u *= 2246822519u; u ^= u >> 13;
        lds[i] = g_random ? ((int)(u & 0xFFFFFF) - 0x800000) * (1.0f / 0x800000) : 1e-3f * (i & 255);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, l31 = lane & 31, h = lane >> 5;
    f32x16 acc[NT];
    for (int t = 0; t < NT; ++t) for (int j = 0; j < 16; ++j) acc[t][j] = 0.f;
    const float* ap0 = lds + l31;
    const float* bp0 = lds + 4096 + l31;
    int off[NT];
    for (int t = 0; t < NT; ++t) off[t] = (t * 37 + (blockIdx.x & 3)) * 32;
    if (VAR == 0) {
        float a = lds[lane], b = lds[lane + 64];
        for (int it = 0; it < iters * 64; ++it)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
    } else if (VAR == 1) {
        for (int it = 0; it < iters; ++it)
#pragma unroll 2
            for (int s = 0; s < 64; ++s) {
                const int v = 2 * s + h;
                const float a = ap0[v * 32];
                const float* bp = bp0 + (v & 63) * 32;
                float bv[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) bv[t] = bp[off[t]];
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv[t], acc[t], 0, 0, 0);
            }
    } else if (VAR == 2) {
        for (int it = 0; it < iters; ++it) {
            float a = ap0[h * 32], bv[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) bv[t] = (bp0 + h * 32)[off[t]];
#pragma unroll 2
            for (int s = 0; s < 64; ++s) {
                const int vn = 2 * ((s + 1) & 63) + h;
                const float an = ap0[vn * 32];
                const float* bpn = bp0 + (vn & 63) * 32;
                float bn[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) bn[t] = bpn[off[t]];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv[t], acc[t], 0, 0, 0);
                a = an;
#pragma unroll
                for (int t = 0; t < NT; ++t) bv[t] = bn[t];
            }
        }
    } else {
        const float* ap = lds + l31 * 36 + 4 * h;
        const float* bp = lds + 8192 + l31 * 36 + 4 * h;
        for (int it = 0; it < iters * 14; ++it) {     // 14 "taps" x 32 MFMA ~ same MFMA count as 64 x 7
            float4 a = *(const float4*)ap, b0 = *(const float4*)bp, b1 = *(const float4*)(bp + 32 * 36);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4 an, b0n, b1n;
                if (q < 3) { an = *(const float4*)(ap + 8 * (q + 1)); b0n = *(const float4*)(bp + 8 * (q + 1)); b1n = *(const float4*)(bp + 32 * 36 + 8 * (q + 1)); }
                __builtin_amdgcn_sched_barrier(0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b0.x, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b1.x, acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b0.y, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b1.y, acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b0.z, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b1.z, acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b0.w, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b1.w, acc[1], 0, 0, 0);
                if (q < 3) { a = an; b0 = b0n; b1 = b1n; }
            }
        }
    }
    float s = 0.f;
    for (int t = 0; t < NT; ++t) for (int j = 0; j < 16; ++j) s += acc[t][j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int VAR> void run(const char* name, float* out, int wgs) {
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    const int iters = 600;
    for (int rep = 0; rep < 6; ++rep) {
        hipEventRecord(s);
        hipLaunchKernelGGL(k<VAR>, dim3(wgs), dim3(256), 65536, 0, out, iters);
        hipEventRecord(e); hipEventSynchronize(e);
        float ms; hipEventElapsedTime(&ms, s, e);
        const double nm = VAR == 3 ? (double)iters * 14 * 32 : (double)iters * 64 * NT;
        if (rep == 5) printf("%-44s %d WG/CU: %.3f ms  %.1f TFLOP/s\n", name, wgs / 256, ms, wgs * 4 * nm * 4096.0 / ms / 1e9);
    }
}
int main(int argc, char** argv) {
    float* out; hipMalloc(&out, 4096 * 256 * 4);
    int rnd = argc > 1 ? atoi(argv[1]) : 0;
    hipMemcpyToSymbol(HIP_SYMBOL(g_random), &rnd, sizeof(int));
    printf("operands: %s\n", rnd ? "random full-range" : "smooth small");
    for (int w = 256; w <= 512; w += 256) {
        run<0>("pure MFMA, 7 acc", out, w);
        run<1>("LDS b32 x8 -> 7 MFMA (compiler order)", out, w);
        run<2>("LDS b32 x8 prefetched one step ahead", out, w);
        run<3>("LDS b128 x3 / 8 MFMA prefetched (fwd style)", out, w);
    }
    return 0;
}
