// Batched fp32 GEMM on v_mfma_f32_32x32x2_f32 for the attention contractions (QK^T, PV, k^T v, q ctx):
// C[g] = alpha * op(A[g]) * op(B[g]) + beta * C[g].  64x64 tile per 256-thread workgroup (4 waves of
// 32x32), K staged 16 at a time through LDS in k-major order so both MFMA operands are conflict-free
// ds_read_b32.  General strides / transposes; edges are zero-filled.
#include "common.h"
#include <stdlib.h>

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
// ---------------------------------------------------------------------------------------------
// version 2 (M > 64): 128 x 64 tile per 256-thread workgroup, each wave 32 rows x 64 columns (2 accumulators), K staged 32
// at a time in the conv kernel's LDS image (36-float padded rows, k-contiguous) so both operands are ds_read_b128
// fragments; the next K-chunk is fetched global -> registers while the current one's 32 MFMAs per wave run.
// ---------------------------------------------------------------------------------------------
constexpr int G2N = 64, G2K = 32, G2ROW = 36;

// MT = 128: 4 waves stacked along M, each 32 rows x 64 columns (2 accumulators); MT = 64 (M <= 64): 2 x 2 waves of 32 x 32.
// Split-K: blockIdx.z = g * ks + slice; a slice covers K range [slice*kslice, ...) and writes its own output slab
// (C then points at the slabs, sC is the slab stride and beta is 0); bgemm_slab_reduce_kernel sums the slices in order.
template <bool TA, bool TB, int MT>
__global__ __launch_bounds__(256, 2) void bgemm2_kernel(const float* __restrict__ A, const float* __restrict__ Bm,
                                                        float* __restrict__ C, int M, int N, int Kfull, long long sA, long long sB,
                                                        long long sC, int lda, int ldb, int ldc, float alpha, float beta, int ks,
                                                        int kslice) {
    constexpr int G2M = MT;
    constexpr int WPM = MT / 32, NACC = 2 / (4 / WPM);          // waves along M; accumulators (32-column blocks) per wave
    // operand images: k-contiguous rows [row][36] for an operand whose K axis is contiguous in memory (b128 fragments), k-major
    // [k][rows + 4] for a transposed one (its rows are contiguous in memory: 16-byte LDS writes, conflict-free b32 fragments)
    constexpr int AT = G2M + 4, BT = G2N + 4;
    __shared__ __attribute__((aligned(16))) float As[TA ? G2K * AT : G2M * G2ROW];
    __shared__ __attribute__((aligned(16))) float Bs[TB ? G2N * G2ROW : G2K * BT];
    const int g = blockIdx.z / ks, slice = blockIdx.z % ks;
    const long long kbeg = (long long)slice * kslice;
    const int K = (int)((Kfull - kbeg) < kslice ? (Kfull - kbeg) : kslice);
    const float* Ag = A + (size_t)g * sA + (TA ? kbeg * lda : kbeg);
    const float* Bg = Bm + (size_t)g * sB + (TB ? kbeg : kbeg * ldb);
    float* Cg = C + (size_t)blockIdx.z * sC;
    const int m0 = blockIdx.y * G2M, n0 = blockIdx.x * G2N;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wave % WPM, wn = wave / WPM;
    const bool vecA = (lda % 4 == 0) && ((reinterpret_cast<uintptr_t>(Ag) & 15u) == 0);
    const bool vecB = (ldb % 4 == 0) && ((reinterpret_cast<uintptr_t>(Bg) & 15u) == 0);

    // staging pieces: A = 128 x 32 floats = 1024 float4 (4 per thread), B = 64 x 32 = 512 float4 (2 per thread).
    // non-transposed operand (k contiguous): piece e -> row e/8, k4 = (e%8)*4.  transposed (row index contiguous in memory):
    // piece e -> k = e / (rows/4), r4 = (e % (rows/4)) * 4, scattered into 4 LDS rows.
    float4 ra[MT / 32], rb[2];
    auto load_chunk = [&](int k0) {
#pragma unroll
        for (int u = 0; u < MT / 32; ++u) {
            const int e = tid + 256 * u;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!TA) {
                const int m = m0 + (e >> 3), k = k0 + (e & 7) * 4;
                if (m < M) {
                    const float* p = Ag + (size_t)m * lda + k;
                    if (vecA && k + 3 < K) v = *reinterpret_cast<const float4*>(p);
                    else { if (k < K) v.x = p[0]; if (k + 1 < K) v.y = p[1]; if (k + 2 < K) v.z = p[2]; if (k + 3 < K) v.w = p[3]; }
                }
            } else {
                const int k = k0 + e / (MT / 4), m = m0 + (e % (MT / 4)) * 4;
                if (k < K) {
                    const float* p = Ag + (size_t)k * lda + m;
                    if (vecA && m + 3 < M) v = *reinterpret_cast<const float4*>(p);
                    else { if (m < M) v.x = p[0]; if (m + 1 < M) v.y = p[1]; if (m + 2 < M) v.z = p[2]; if (m + 3 < M) v.w = p[3]; }
                }
            }
            ra[u] = v;
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int e = tid + 256 * u;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (TB) {
                const int n = n0 + (e >> 3), k = k0 + (e & 7) * 4;
                if (n < N) {
                    const float* p = Bg + (size_t)n * ldb + k;
                    if (vecB && k + 3 < K) v = *reinterpret_cast<const float4*>(p);
                    else { if (k < K) v.x = p[0]; if (k + 1 < K) v.y = p[1]; if (k + 2 < K) v.z = p[2]; if (k + 3 < K) v.w = p[3]; }
                }
            } else {
                const int k = k0 + (e >> 4), n = n0 + (e & 15) * 4;
                if (k < K) {
                    const float* p = Bg + (size_t)k * ldb + n;
                    if (vecB && n + 3 < N) v = *reinterpret_cast<const float4*>(p);
                    else { if (n < N) v.x = p[0]; if (n + 1 < N) v.y = p[1]; if (n + 2 < N) v.z = p[2]; if (n + 3 < N) v.w = p[3]; }
                }
            }
            rb[u] = v;
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int u = 0; u < MT / 32; ++u) {
            const int e = tid + 256 * u;
            if (!TA) *reinterpret_cast<float4*>(As + (e >> 3) * G2ROW + (e & 7) * 4) = ra[u];
            else *reinterpret_cast<float4*>(As + (e / (MT / 4)) * AT + (e % (MT / 4)) * 4) = ra[u];
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int e = tid + 256 * u;
            if (TB) *reinterpret_cast<float4*>(Bs + (e >> 3) * G2ROW + (e & 7) * 4) = rb[u];
            else *reinterpret_cast<float4*>(Bs + (e >> 4) * BT + (e & 15) * 4) = rb[u];
        }
    };

    f32x16 acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
    load_chunk(0);
    for (int k0 = 0; k0 < K; k0 += G2K) {
        __syncthreads();                 // the previous chunk's fragment reads are done
        store_chunk();
        __syncthreads();
        if (k0 + G2K < K) load_chunk(k0 + G2K);
        // fragment of k-group q: lane half h supplies k = 8q + 4h + j to the j-th MFMA pair
        auto frag_a = [&](int q) -> float4 {
            if (!TA) return *reinterpret_cast<const float4*>(As + (wm * 32 + l31) * G2ROW + 4 * h + 8 * q);
            const float* p = As + (8 * q + 4 * h) * AT + wm * 32 + l31;
            return make_float4(p[0], p[AT], p[2 * AT], p[3 * AT]);
        };
        auto frag_b = [&](int q, int half) -> float4 {
            if (TB) return *reinterpret_cast<const float4*>(Bs + ((half + wn) * 32 + l31) * G2ROW + 4 * h + 8 * q);
            const float* p = Bs + (8 * q + 4 * h) * BT + (half + wn) * 32 + l31;
            return make_float4(p[0], p[BT], p[2 * BT], p[3 * BT]);
        };
        float4 a = frag_a(0), b0 = frag_b(0, 0), b1 = NACC == 2 ? frag_b(0, 1) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 an, b0n, b1n;
            if (q < 3) { an = frag_a(q + 1); b0n = frag_b(q + 1, 0); if (NACC == 2) b1n = frag_b(q + 1, 1); }
            __builtin_amdgcn_sched_barrier(0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b0.x, acc0, 0, 0, 0);
            if (NACC == 2) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b1.x, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b0.y, acc0, 0, 0, 0);
            if (NACC == 2) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b1.y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b0.z, acc0, 0, 0, 0);
            if (NACC == 2) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b1.z, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b0.w, acc0, 0, 0, 0);
            if (NACC == 2) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b1.w, acc1, 0, 0, 0);
            if (q < 3) { a = an; b0 = b0n; if (NACC == 2) b1 = b1n; }
        }
    }
    // D[row][col]: col = lane & 31 (+32 for acc1), row = (r&3) + 8*(r>>2) + 4*h
    const int c0 = n0 + wn * 32 + l31, c1 = n0 + 32 + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row >= M) continue;
        float* dst = Cg + (size_t)row * ldc;
        if (c0 < N) { float v = alpha * acc0[r]; if (beta != 0.f) v += beta * dst[c0]; dst[c0] = v; }
        if (NACC == 2 && c1 < N) { float v = alpha * acc1[r]; if (beta != 0.f) v += beta * dst[c1]; dst[c1] = v; }
    }
}

// C[g] = alpha * sum_s slab[g*ks + s] + beta * C[g]   (slabs are dense [M][N])
__global__ __launch_bounds__(256) void bgemm_slab_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ C, int M, int N,
                                                                int ks, long long sC, int ldc, float alpha, float beta) {
    const int g = blockIdx.y;
    const int MN = M * N;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < MN; i += gridDim.x * 256) {
        float s = 0.f;
        for (int k = 0; k < ks; ++k) s += slabs[((size_t)g * ks + k) * MN + i];
        float* dst = C + (size_t)g * sC + (size_t)(i / N) * ldc + (i % N);
        float v = alpha * s;
        if (beta != 0.f) v += beta * (*dst);
        *dst = v;
    }
}
}  // namespace diqt

using namespace diqt;
// split-K plan: few output tiles and a long K (dK/dV of the attentions, 1-row / 5-row GEMMs) -> slices over blockIdx.z
static int bgemm_ksplit(int batch, int M, int N, int K) {
    const int MT = M > 64 ? 128 : 64;
    const long long wgs = (long long)batch * ((M + MT - 1) / MT) * ((N + G2N - 1) / G2N);
    if (wgs >= 256 || K < 1024) return 1;
    long long ks = 512 / wgs;
    if (ks > K / 256) ks = K / 256;
    if (ks * batch > 65535) ks = 65535 / batch;
    return ks < 2 ? 1 : (int)ks;
}

extern "C" size_t diqt_bgemm_workspace_bytes(int batch, int M, int N, int K) {
    const int ks = bgemm_ksplit(batch, M, N, K);
    return ks > 1 ? (size_t)batch * ks * M * N * sizeof(float) : 0;
}

static int bgemm_impl(const float* A, const float* Bm, float* C, void* workspace, size_t workspace_bytes, int batch, int M, int N,
                      int K, int transA, int transB, long long strideA, long long strideB, long long strideC, int lda, int ldb,
                      int ldc, float alpha, float beta, void* stream) {
    DIQT_REQUIRE(A && Bm && C, DIQT_E_ALIGN, "bgemm: null pointer");
    DIQT_REQUIRE(batch > 0 && M > 0 && N > 0 && K > 0 && lda > 0 && ldb > 0 && ldc > 0, DIQT_E_SHAPE, "bgemm: bad shape");
    DIQT_REQUIRE(batch <= 65535, DIQT_E_SHAPE, "bgemm: batch > 65535");
    static const bool v1 = [] { const char* e = getenv("DIQT_BGEMM_V1"); return e && e[0] == '1'; }();
    if (v1) {
        const dim3 grid((N + GT - 1) / GT, (M + GT - 1) / GT, batch);
        hipLaunchKernelGGL(bgemm_kernel, grid, dim3(256), 0, (hipStream_t)stream, A, Bm, C, M, N, K, transA, transB, strideA,
                           strideB, strideC, lda, ldb, ldc, alpha, beta);
        return check_launch("bgemm(v1)");
    }
    typedef void (*kfn)(const float*, const float*, float*, int, int, int, long long, long long, long long, int, int, int, float, float,
                        int, int);
    const bool big = M > 64;
    const kfn k2 = big ? (transA ? (transB ? (kfn)bgemm2_kernel<true, true, 128> : (kfn)bgemm2_kernel<true, false, 128>)
                                 : (transB ? (kfn)bgemm2_kernel<false, true, 128> : (kfn)bgemm2_kernel<false, false, 128>))
                       : (transA ? (transB ? (kfn)bgemm2_kernel<true, true, 64> : (kfn)bgemm2_kernel<true, false, 64>)
                                 : (transB ? (kfn)bgemm2_kernel<false, true, 64> : (kfn)bgemm2_kernel<false, false, 64>));
    const int MT = big ? 128 : 64;
    int ks = workspace ? bgemm_ksplit(batch, M, N, K) : 1;
    if (ks > 1 && workspace_bytes < (size_t)batch * ks * M * N * sizeof(float)) ks = 1;
    const dim3 grid2((N + G2N - 1) / G2N, (M + MT - 1) / MT, batch * ks);
    DIQT_REQUIRE(grid2.y <= 65535, DIQT_E_SHAPE, "bgemm: M too large");
    if (ks == 1) {
        hipLaunchKernelGGL(k2, grid2, dim3(256), 0, (hipStream_t)stream, A, Bm, C, M, N, K, strideA, strideB, strideC, lda, ldb, ldc,
                           alpha, beta, 1, K);
        return check_launch("bgemm(v2)");
    }
    float* slabs = static_cast<float*>(workspace);
    const int kslice = ((K + ks - 1) / ks + 31) / 32 * 32;            // whole 32-wide chunks per slice
    hipLaunchKernelGGL(k2, grid2, dim3(256), 0, (hipStream_t)stream, A, Bm, slabs, M, N, K, strideA, strideB, (long long)M * N, lda,
                       ldb, N, 1.f, 0.f, ks, kslice);
    int rc = check_launch("bgemm(v2 split-K)");
    if (rc) return rc;
    hipLaunchKernelGGL(bgemm_slab_reduce_kernel, dim3(grid_for((size_t)M * N, 256, 256), batch), dim3(256), 0, (hipStream_t)stream,
                       slabs, C, M, N, ks, strideC, ldc, alpha, beta);
    return check_launch("bgemm(split-K reduce)");
}

extern "C" int diqt_bgemm(const float* A, const float* Bm, float* C, int batch, int M, int N, int K, int transA,
                          int transB, long long strideA, long long strideB, long long strideC, int lda, int ldb,
                          int ldc, float alpha, float beta, void* stream) {
    return bgemm_impl(A, Bm, C, nullptr, 0, batch, M, N, K, transA, transB, strideA, strideB, strideC, lda, ldb, ldc, alpha, beta,
                      stream);
}

extern "C" int diqt_bgemm_ws(const float* A, const float* Bm, float* C, void* workspace, size_t workspace_bytes, int batch, int M,
                             int N, int K, int transA, int transB, long long strideA, long long strideB, long long strideC, int lda,
                             int ldb, int ldc, float alpha, float beta, void* stream) {
    return bgemm_impl(A, Bm, C, workspace, workspace_bytes, batch, M, N, K, transA, transB, strideA, strideB, strideC, lda, ldb, ldc,
                      alpha, beta, stream);
}
