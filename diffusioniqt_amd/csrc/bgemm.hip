// Batched fp32 GEMM on v_mfma_f32_32x32x2_f32 for the attention contractions (QK^T, PV, k^T v, q ctx):
// C[g] = alpha * op(A[g]) * op(B[g]) + beta * C[g].  64x64 tile per 256-thread workgroup (4 waves of
// 32x32), K staged 16 at a time through LDS in k-major order so both MFMA operands are conflict-free
// ds_read_b32.  General strides / transposes; edges are zero-filled.
#include "common.h"

namespace diqt {
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int GT = 64, GK = 16;

__global__ __launch_bounds__(256) void bgemm_kernel(const float* __restrict__ A, const float* __restrict__ Bm,
                                                    float* __restrict__ C, int M, int N, int K, int transA, int transB,
                                                    long long sA, long long sB, long long sC, int lda, int ldb, int ldc,
                                                    float alpha, float beta) {
    __shared__ float As[GK][GT + 1];
    __shared__ float Bs[GK][GT + 1];
    const int g = blockIdx.z;
    const float* Ag = A + (size_t)g * sA;
    const float* Bg = Bm + (size_t)g * sB;
    float* Cg = C + (size_t)g * sC;
    const int m0 = blockIdx.y * GT, n0 = blockIdx.x * GT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    for (int k0 = 0; k0 < K; k0 += GK) {
        __syncthreads();
        // A tile: element (m, k) ; coalesce along the contiguous axis of the source
        for (int e = tid; e < GT * GK; e += 256) {
            int m, k;
            if (transA) { m = e % GT; k = e / GT; } else { k = e % GK; m = e / GK; }
            const int gm = m0 + m, gk = k0 + k;
            float v = 0.f;
            if (gm < M && gk < K) v = transA ? Ag[(size_t)gk * lda + gm] : Ag[(size_t)gm * lda + gk];
            As[k][m] = v;
        }
        for (int e = tid; e < GT * GK; e += 256) {
            int n, k;
            if (transB) { k = e % GK; n = e / GK; } else { n = e % GT; k = e / GT; }
            const int gn = n0 + n, gk = k0 + k;
            float v = 0.f;
            if (gn < N && gk < K) v = transB ? Bg[(size_t)gn * ldb + gk] : Bg[(size_t)gk * ldb + gn];
            Bs[k][n] = v;
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < GK / 2; ++s) {
            const float a = As[2 * s + h][wm + l31];
            const float b = Bs[2 * s + h][wn + l31];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
    }
    const int col = n0 + wn + l31;
    if (col < N) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + wm + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (row < M) {
                float v = alpha * acc[r];
                float* dst = Cg + (size_t)row * ldc + col;
                if (beta != 0.f) v += beta * (*dst);
                *dst = v;
            }
        }
    }
}
}  // namespace diqt

using namespace diqt;
extern "C" int diqt_bgemm(const float* A, const float* Bm, float* C, int batch, int M, int N, int K, int transA,
                          int transB, long long strideA, long long strideB, long long strideC, int lda, int ldb,
                          int ldc, float alpha, float beta, void* stream) {
    DIQT_REQUIRE(A && Bm && C, DIQT_E_ALIGN, "bgemm: null pointer");
    DIQT_REQUIRE(batch > 0 && M > 0 && N > 0 && K > 0 && lda > 0 && ldb > 0 && ldc > 0, DIQT_E_SHAPE, "bgemm: bad shape");
    DIQT_REQUIRE(batch <= 65535, DIQT_E_SHAPE, "bgemm: batch > 65535");
    const dim3 grid((N + GT - 1) / GT, (M + GT - 1) / GT, batch);
    hipLaunchKernelGGL(bgemm_kernel, grid, dim3(256), 0, (hipStream_t)stream, A, Bm, C, M, N, K, transA, transB, strideA,
                       strideB, strideC, lda, ldb, ldc, alpha, beta);
    return check_launch("bgemm");
}
