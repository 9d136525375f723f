// Practical ceiling of v_mfma_f32_32x32x2_f32 on this chip: pure register loop, W waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    float a = a0 + threadIdx.x * 1e-3f, b = b0 + threadIdx.x * 2e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float* out; hipMalloc(&out, 4096 * 256 * 4);
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    for (int wg_per_cu = 1; wg_per_cu <= 2; ++wg_per_cu) {
        const int blocks = 256 * wg_per_cu, iters = 4000;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(s);
            hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f);
            hipEventRecord(e); hipEventSynchronize(e);
            float ms; hipEventElapsedTime(&ms, s, e);
            double flops = (double)blocks * 4 * iters * 8 * 2 * 4096.0;
            printf("pure MFMA f32 32x32x2, %d WG/CU (4 waves each), 2 acc: %.3f ms  %.1f TFLOP/s\n", wg_per_cu, ms, flops / ms / 1e9);
        }
    }
    return 0;
}
