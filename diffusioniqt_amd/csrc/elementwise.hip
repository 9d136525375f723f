// HBM-bound kernels of the DiffusionIQT hot path: GroupNorm statistics + fused normalise/scale-shift/
// activation (fwd/bwd), channel LayerNorm, squeeze-excite pooling/gating, pixel (un)shuffle, channel
// concat/split, sub-volume gather/scatter, diffusion step math, loss, Adam/EMA, softmax.
// All activations are fp32 channels-last rows x[row][C]; every streaming access is 16 B per lane when
// C % 4 == 0 and the pointers are 16-byte aligned (the scalar variants cover the rest).
#include "common.h"
#include <stdlib.h>

namespace diqt {

constexpr int RED_NBLK = 256;  // row slices per batch element in the column reductions (8 x 256 workgroups: 64 slices left the loads of a 32^3 x 64 pass at 4 TB/s)

// four consecutive channels of a tensor that holds fp32 (ty 0), fp16 (1) or bf16 (2) values; `i` = element index (a multiple of 4)
__device__ __forceinline__ float4 ld4_any(const void* p, size_t i, int ty) {
    if (ty == 0) return *reinterpret_cast<const float4*>(static_cast<const float*>(p) + i);
    const uint2 u = *reinterpret_cast<const uint2*>(static_cast<const unsigned short*>(p) + i);
    if (ty == 2) return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
    typedef _Float16 h2v __attribute__((ext_vector_type(2)));
    const unsigned ux = u.x, uy = u.y;       // (by value: __builtin_bit_cast on a vector component reads element 0 on this hipcc, see conv_half.hip)
    const h2v a = __builtin_bit_cast(h2v, ux), b = __builtin_bit_cast(h2v, uy);
    return make_float4((float)a[0], (float)a[1], (float)b[0], (float)b[1]);
}
// the type as a template argument (TY >= 0): no branch in front of the load, so the loads of an unrolled loop are issued together --
// with the run-time switch the GroupNorm-backward reduction over a 16-bit x AND a 16-bit dy ran at 1.9 TB/s (34.7 us vs 27.4 us fp32 dy)
template <int TY>
__device__ __forceinline__ float4 ld4_t(const void* p, size_t i, int ty) { return ld4_any(p, i, TY >= 0 ? TY : ty); }
__device__ __forceinline__ void st4_any(void* p, size_t i, float4 v, int ty) {
    if (ty == 0) { *reinterpret_cast<float4*>(static_cast<float*>(p) + i) = v; return; }
    uint2 u;
    if (ty == 2) {
        typedef __bf16 b2v __attribute__((ext_vector_type(2)));
        const b2v a = {(__bf16)v.x, (__bf16)v.y}, b = {(__bf16)v.z, (__bf16)v.w};
        u.x = __builtin_bit_cast(unsigned, a); u.y = __builtin_bit_cast(unsigned, b);
    } else {
        typedef _Float16 h2v __attribute__((ext_vector_type(2)));
        const h2v a = {(_Float16)v.x, (_Float16)v.y}, b = {(_Float16)v.z, (_Float16)v.w};
        u.x = __builtin_bit_cast(unsigned, a); u.y = __builtin_bit_cast(unsigned, b);
    }
    *reinterpret_cast<uint2*>(static_cast<unsigned short*>(p) + i) = u;
}

template <int TY>
__device__ __forceinline__ void st4_t(void* p, size_t i, float4 v, int ty) { st4_any(p, i, v, TY >= 0 ? TY : ty); }

// ---------------------------------------------------------------------------------------------
// generic per-(b,c) column reduction: partial[b][blk][NV][C] = sum over the block's rows of f(...)
// ---------------------------------------------------------------------------------------------
template <int NV, class F>
__global__ __launch_bounds__(256) void colreduce_kernel(F f, float* __restrict__ partial, int rows, int C) {
    __shared__ float4 sh[NV][256];
    const int b = blockIdx.y, blk = blockIdx.x, nblk = gridDim.x;
    const int rowsPer = (rows + nblk - 1) / nblk;
    const int r0 = blk * rowsPer;
    int r1 = r0 + rowsPer;
    if (r1 > rows) r1 = rows;
    float* dst = partial + ((size_t)b * nblk + blk) * NV * C;
    if ((C & 3) == 0 && C <= 1024) {
        const int tpr = C >> 2, rpar = 256 / tpr;
        const int rr = threadIdx.x / tpr, cq = threadIdx.x % tpr;
        float acc[NV][4];
#pragma unroll
        for (int v = 0; v < NV; ++v)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[v][j] = 0.f;
        F fl = f;
        fl.prep(b, cq * 4);          // per-thread loop invariants (a thread keeps its channel quad)
        if (rr < rpar) {
            int r = r0 + rr;
            // 4 independent 16-byte loads in flight per lane (the reduction is latency-bound otherwise)
            for (; r + 3 * rpar < r1; r += 4 * rpar) {
                float o[4][NV][4];
#pragma unroll
                for (int u = 0; u < 4; ++u) fl.vec4(((size_t)b * rows + r + u * rpar) * C + cq * 4, b, cq * 4, o[u]);
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int v = 0; v < NV; ++v)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[v][j] += o[u][v][j];
            }
            for (; r < r1; r += rpar) {
                float o[NV][4];
                fl.vec4(((size_t)b * rows + r) * C + cq * 4, b, cq * 4, o);
#pragma unroll
                for (int v = 0; v < NV; ++v)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[v][j] += o[v][j];
            }
        }
#pragma unroll
        for (int v = 0; v < NV; ++v) sh[v][threadIdx.x] = make_float4(acc[v][0], acc[v][1], acc[v][2], acc[v][3]);
        __syncthreads();
        if (threadIdx.x < tpr) {
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                float4 s = sh[v][threadIdx.x];
                for (int k = 1; k < rpar; ++k) {
                    const float4 t = sh[v][k * tpr + threadIdx.x];
                    s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
                }
                *reinterpret_cast<float4*>(dst + v * C + threadIdx.x * 4) = s;
            }
        }
    } else {
        for (int c = threadIdx.x; c < C; c += 256) {
            float acc[NV];
#pragma unroll
            for (int v = 0; v < NV; ++v) acc[v] = 0.f;
            for (int r = r0; r < r1; ++r) {
                float o[NV];
                f.scalar(((size_t)b * rows + r) * C + c, b, c, o);
#pragma unroll
                for (int v = 0; v < NV; ++v) acc[v] += o[v];
            }
#pragma unroll
            for (int v = 0; v < NV; ++v) dst[v * C + c] = acc[v];
        }
    }
}

static inline int red_nblk(int rows) {
    static const int cap = [] { const char* e = getenv("DIQT_RED_NBLK"); const int v = e ? atoi(e) : RED_NBLK; return v < 1 || v > RED_NBLK ? RED_NBLK : v; }();
    int n = rows / 64;
    if (n > cap) n = cap;
    if (n < 1) n = 1;
    return n;
}

// ---------------------------------------------------------------------------------------------
// GlobalContext pooling in ONE pass over x (imagen_video.py:957-982, sampling path):
//   pooled[b][c] = sum_n softmax_n(x[b][n][:] . w)[n] * x[b][n][c]
// The three-kernel form (to_k conv, soft-max over all positions, weighted column sum) reads x twice and round-trips the logits; here
// every row is read once: its logit is a dot product reduced over the C/4 lanes that hold the row, and the weighted sum is kept
// online (running maximum m, normaliser l, sums acc: rescaled by exp(m - m') when the maximum moves), per row group, per workgroup
// (LDS), then over the workgroups of a batch entry by gc_pool_final_kernel -- all in a fixed order.  The conv's bias shifts every logit
// alike and drops out of the soft-max.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gc_pool_kernel(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ partial,
                                                      int rows, int C) {
    __shared__ float4 shA[256];
    __shared__ float shM[64], shL[64];
    const int b = blockIdx.y, blk = blockIdx.x, nblk = gridDim.x;
    const int rowsPer = (rows + nblk - 1) / nblk;
    const int r0 = blk * rowsPer;
    const int r1 = min(r0 + rowsPer, rows);
    const int tpr = C >> 2, rpar = 256 / tpr;                  // lanes per row (16, 32 or 64: inside one wave), rows in flight
    const int rr = threadIdx.x / tpr, cq = threadIdx.x % tpr;
    const float4 wv = *reinterpret_cast<const float4*>(w + cq * 4);
    const float* xb = x + (size_t)b * rows * C + cq * 4;
    float m = -INFINITY, l = 0.f;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    auto row_dot = [&](const float4& v) {
        float d = v.x * wv.x + v.y * wv.y + v.z * wv.z + v.w * wv.w;
        for (int o = 1; o < tpr; o <<= 1) d += __shfl_xor(d, o, 64);
        return d;
    };
    auto take = [&](const float4& v, float sc) {
        const float mn = fmaxf(m, sc);
        const float f = __expf(m - mn), e = __expf(sc - mn);       // m = -inf on the first row: f = 0
        acc.x = acc.x * f + e * v.x; acc.y = acc.y * f + e * v.y; acc.z = acc.z * f + e * v.z; acc.w = acc.w * f + e * v.w;
        l = l * f + e;
        m = mn;
    };
    int r = r0 + rr;
    for (; r + 3 * rpar < r1; r += 4 * rpar) {                 // 4 independent 16-byte loads in flight per lane
        float4 v[4];
        float sc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(xb + (size_t)(r + u * rpar) * C);
#pragma unroll
        for (int u = 0; u < 4; ++u) sc[u] = row_dot(v[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u) take(v[u], sc[u]);
    }
    for (; r < r1; r += rpar) {                                // (a row group's lanes leave the loop together: r depends on rr only)
        const float4 v = *reinterpret_cast<const float4*>(xb + (size_t)r * C);
        take(v, row_dot(v));
    }
    shA[threadIdx.x] = acc;
    if (cq == 0) { shM[rr] = m; shL[rr] = l; }
    __syncthreads();
    if (threadIdx.x < tpr) {
        float M = -INFINITY;
        for (int k = 0; k < rpar; ++k) M = fmaxf(M, shM[k]);
        float L = 0.f;
        float4 A = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k = 0; k < rpar; ++k) {
            const float f = shM[k] == -INFINITY ? 0.f : __expf(shM[k] - M);
            const float4 t = shA[k * tpr + threadIdx.x];
            L += shL[k] * f;
            A.x += t.x * f; A.y += t.y * f; A.z += t.z * f; A.w += t.w * f;
        }
        float* dst = partial + ((size_t)b * nblk + blk) * (C + 4);
        if (threadIdx.x == 0) { dst[0] = M; dst[1] = L; }
        *reinterpret_cast<float4*>(dst + 4 + threadIdx.x * 4) = A;
    }
}
__global__ __launch_bounds__(256) void gc_pool_final_kernel(const float* __restrict__ partial, float* __restrict__ pooled, int nblk, int C) {
    // nblk <= 256 partials (m, l, acc[C]) of one batch entry: thread k scales partial k, then thread c sums channel c over k in order
    __shared__ float f[256], red[256];
    const int b = blockIdx.x, t = threadIdx.x;
    const float* p = partial + (size_t)b * nblk * (C + 4);
    const float mk = t < nblk ? p[(size_t)t * (C + 4)] : -INFINITY;
    red[t] = mk;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (t < o) red[t] = fmaxf(red[t], red[t + o]);
        __syncthreads();
    }
    const float M = red[0];
    __syncthreads();
    const float fk = mk == -INFINITY ? 0.f : __expf(mk - M);
    f[t] = fk;
    red[t] = t < nblk ? p[(size_t)t * (C + 4) + 1] * fk : 0.f;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {                  // fixed tree: deterministic
        if (t < o) red[t] += red[t + o];
        __syncthreads();
    }
    const float L = red[0];
    for (int c = t; c < C; c += 256) {
        float a = 0.f;
        for (int k = 0; k < nblk; ++k) a += p[(size_t)k * (C + 4) + 4 + c] * f[k];
        pooled[(size_t)b * C + c] = a / L;
    }
}

// ---------------------------------------------------------------------------------------------
// GroupNorm statistics
// ---------------------------------------------------------------------------------------------
struct MomentsF {
    const float* x;
    __device__ void prep(int, int) {}
    __device__ void vec4(size_t i, int, int, float (&o)[2][4]) const {
        const float4 v = *reinterpret_cast<const float4*>(x + i);
        o[0][0] = v.x; o[0][1] = v.y; o[0][2] = v.z; o[0][3] = v.w;
        o[1][0] = v.x * v.x; o[1][1] = v.y * v.y; o[1][2] = v.z * v.z; o[1][3] = v.w * v.w;
    }
    __device__ void scalar(size_t i, int, int, float (&o)[2]) const {
        const float v = x[i];
        o[0] = v; o[1] = v * v;
    }
};

struct GnCoefOut {          // optional second result of the statistics finalisation: y = act(A x + Bc) coefficients per (b, c)
    float* coef;            // [2][B][C]: A, then Bc (nullptr: none)
    const float *gamma, *beta, *scale, *shift;
    int cs;
};
__global__ __launch_bounds__(512) void gn_stats_final_kernel(const float* __restrict__ partial, float* __restrict__ mean,
                                                             float* __restrict__ rstd, int B, int C, int G, int nblk,
                                                             double count, float eps, GnCoefOut co) {
    // one 512-thread workgroup per (b,g): threads stride over the nblk x Cg partial sums (a 64^3 volume has 2048 tiles: one wave
    // per (b,g) took 58 us per launch on C4), fp64 combine in a fixed order: lanes by butterfly, then the 8 waves through LDS
    __shared__ double sh[16];
    const int i = blockIdx.x;
    const int b = i / G, g = i % G, Cg = C / G;
    double s = 0.0, ss = 0.0;
    for (int e = threadIdx.x; e < nblk * Cg; e += 512) {
        const int k = e / Cg, c = g * Cg + e % Cg;
        const float* p = partial + ((size_t)b * nblk + k) * 2 * C;
        s += p[c]; ss += p[C + c];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); ss += __shfl_xor(ss, o, 64); }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sh[wave] = s; sh[8 + wave] = ss; }
    __syncthreads();
    if (threadIdx.x == 0) {
        s = 0.0; ss = 0.0;
        for (int w = 0; w < 8; ++w) { s += sh[w]; ss += sh[8 + w]; }
        const double m = s / count;
        double var = ss / count - m * m;
        if (var < 0) var = 0;
        mean[i] = (float)m;
        rstd[i] = (float)(1.0 / sqrt(var + (double)eps));
        if (co.coef) { sh[0] = (double)mean[i]; sh[1] = (double)rstd[i]; }
    }
    if (co.coef) {          // kernel-uniform: the group's channels get their coefficients here (same expressions as GnCoef::get)
        __syncthreads();
        const float m = (float)sh[0], r = (float)sh[1];
        for (int e = threadIdx.x; e < Cg; e += 512) {
            const int c = g * Cg + e;
            const float ga = co.gamma ? co.gamma[c] : 1.f, be = co.beta ? co.beta[c] : 0.f;
            const float sc = co.scale ? co.scale[b * co.cs + c] + 1.f : 1.f, sf = co.shift ? co.shift[b * co.cs + c] : 0.f;
            co.coef[(size_t)b * C + c] = r * ga * sc;
            co.coef[(size_t)(B + b) * C + c] = (be - m * r * ga) * sc + sf;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// fused GN-apply + (scale+1)x+shift + activation
// ---------------------------------------------------------------------------------------------
struct GnCoef {   // y = act(A*x + Bc) per (b,c)
    const float *mean, *rstd, *gamma, *beta, *scale, *shift;
    int C, G, cs;   // cs: floats between consecutive batch rows of scale/shift
    __device__ __forceinline__ void get(int b, int c, float& A, float& Bc) const {
        const int g = c / (C / G);
        const float m = mean[b * G + g], r = rstd[b * G + g];
        const float ga = gamma ? gamma[c] : 1.f, be = beta ? beta[c] : 0.f;
        const float sc = scale ? scale[b * cs + c] + 1.f : 1.f, sf = shift ? shift[b * cs + c] : 0.f;
        A = r * ga * sc;
        Bc = (be - m * r * ga) * sc + sf;
    }
};

__global__ __launch_bounds__(256) void gn_coef_kernel(GnCoef k, float* __restrict__ coef, int B) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * k.C) return;
    float A, Bc;
    k.get(i / k.C, i % k.C, A, Bc);
    coef[i] = A;
    coef[(size_t)B * k.C + i] = Bc;
}

// OUTH: 0: fp32 output; 1 / 2: the output in fp16 / bf16 (VEC only) -- under autocast the consumer is a 16-bit-operand conv that would
// round these values to that type while staging them: the same numbers at half the bytes written here and read there
template <int OUTH>
__device__ __forceinline__ void gn_store4(float* yb, size_t i, float4 v) {
    if constexpr (OUTH == 0) {
        *reinterpret_cast<float4*>(yb + i * 4) = v;
    } else {
        uint2 p;
        if constexpr (OUTH == 1) {
            typedef _Float16 h2 __attribute__((ext_vector_type(2)));
            const h2 a = {(_Float16)v.x, (_Float16)v.y}, c = {(_Float16)v.z, (_Float16)v.w};
            p.x = __builtin_bit_cast(unsigned, a); p.y = __builtin_bit_cast(unsigned, c);
        } else {
            typedef __bf16 b2 __attribute__((ext_vector_type(2)));
            const b2 a = {(__bf16)v.x, (__bf16)v.y}, c = {(__bf16)v.z, (__bf16)v.w};
            p.x = __builtin_bit_cast(unsigned, a); p.y = __builtin_bit_cast(unsigned, c);
        }
        *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(yb) + i * 4) = p;
    }
}

// INH: x holds 16-bit values of type OUTH (the 16-bit output of the conv in front, diqt_conv3d_fwd_h_io y_half)
template <int INH>
__device__ __forceinline__ float4 gn_load4(const float* xb, size_t i) {
    if constexpr (INH == 0) {
        return *reinterpret_cast<const float4*>(xb + i * 4);
    } else {
        const uint2 p = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(xb) + i * 4);
        if constexpr (INH == 1) {
            typedef _Float16 h2 __attribute__((ext_vector_type(2)));
            const h2 a = __builtin_bit_cast(h2, p.x), c = __builtin_bit_cast(h2, p.y);
            return make_float4((float)a.x, (float)a.y, (float)c.x, (float)c.y);
        } else {
            return make_float4(__builtin_bit_cast(float, p.x << 16), __builtin_bit_cast(float, p.x & 0xffff0000u),
                               __builtin_bit_cast(float, p.y << 16), __builtin_bit_cast(float, p.y & 0xffff0000u));
        }
    }
}

template <bool VEC, int OUTH = 0, int INH = 0>
__global__ __launch_bounds__(256) void gn_act_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                         GnCoef k, int rows, int act) {
    const int b = blockIdx.y, C = k.C;
    const size_t per = (size_t)rows * C;
    const float* xb = INH ? reinterpret_cast<const float*>(reinterpret_cast<const unsigned short*>(x) + (size_t)b * per) : x + (size_t)b * per;
    float* yb = OUTH ? reinterpret_cast<float*>(reinterpret_cast<unsigned short*>(y) + (size_t)b * per) : y + (size_t)b * per;
    if (VEC) {
        // (8 channels per thread for the 16-bit-in / 16-bit-out build -- 16-byte accesses -- measured SLOWER: 24.9 vs 15.8 us at 8 x 32^3 x 64)
        // the host sizes the grid so that (gridDim.x * 1024) % C == 0: a thread keeps its 4 channels for the whole
        // loop and the per-(b,c) coefficients live in registers
        const size_t n4 = per >> 2, i0 = blockIdx.x * (size_t)256 + threadIdx.x;
        const int c = (int)((i0 * 4) % C);
        float A[4], Bc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) k.get(b, c + j, A[j], Bc[j]);
        const size_t st = (size_t)gridDim.x * 256;
        size_t i = i0;
        for (; i + 3 * st < n4; i += 4 * st) {          // 4 loads in flight
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = gn_load4<INH>(xb, i + u * st);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                v[u].x = act_fwd(A[0] * v[u].x + Bc[0], act);
                v[u].y = act_fwd(A[1] * v[u].y + Bc[1], act);
                v[u].z = act_fwd(A[2] * v[u].z + Bc[2], act);
                v[u].w = act_fwd(A[3] * v[u].w + Bc[3], act);
                gn_store4<OUTH>(yb, i + u * st, v[u]);
            }
        }
        for (; i < n4; i += st) {
            float4 v = gn_load4<INH>(xb, i);
            v.x = act_fwd(A[0] * v.x + Bc[0], act);
            v.y = act_fwd(A[1] * v.y + Bc[1], act);
            v.z = act_fwd(A[2] * v.z + Bc[2], act);
            v.w = act_fwd(A[3] * v.w + Bc[3], act);
            gn_store4<OUTH>(yb, i, v);
        }
    } else {
        for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < per; i += (size_t)gridDim.x * 256) {
            float A, Bc;
            k.get(b, (int)(i % C), A, Bc);
            yb[i] = act_fwd(A * xb[i] + Bc, act);
        }
    }
}

// pass 1 of the backward: S1[b][c] = sum dz, S2[b][c] = sum dz*xhat, dz = dy*act'(z)
struct GnBwdF {
    const float *x, *dy;
    GnCoef k;
    int act;
    int xty = 0;                           // x holds fp32 (0), fp16 (1) or bf16 (2) values (the 16-bit block output of a low-precision training step)
    int dyty = 0;                          // ... and so does dy (the 16-bit output of the backward-data conv in front of this pass)
    __device__ __forceinline__ void one(float xv, float dyv, int b, int c, float& s1, float& s2) const {
        float A, Bc;
        k.get(b, c, A, Bc);
        const int g = c / (k.C / k.G);
        const float xhat = (xv - k.mean[b * k.G + g]) * k.rstd[b * k.G + g];
        const float dz = dyv * act_grad(A * xv + Bc, act);
        s1 = dz; s2 = dz * xhat;
    }
    float cA[4], cB[4], cm[4], cr[4];      // filled by prep() on the device
    __device__ void prep(int b, int c) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            k.get(b, c + j, cA[j], cB[j]);
            const int g = (c + j) / (k.C / k.G);
            cm[j] = k.mean[b * k.G + g]; cr[j] = k.rstd[b * k.G + g];
        }
    }
    __device__ __forceinline__ void onej(float xv, float dyv, int j, float& s1, float& s2) const {
        const float xhat = (xv - cm[j]) * cr[j];
        const float dz = dyv * act_grad(cA[j] * xv + cB[j], act);
        s1 = dz; s2 = dz * xhat;
    }
    __device__ void vec4(size_t i, int, int, float (&o)[2][4]) const {
        const float4 xv = ld4_any(x, i, xty);
        const float4 dv = ld4_any(dy, i, dyty);
        onej(xv.x, dv.x, 0, o[0][0], o[1][0]);
        onej(xv.y, dv.y, 1, o[0][1], o[1][1]);
        onej(xv.z, dv.z, 2, o[0][2], o[1][2]);
        onej(xv.w, dv.w, 3, o[0][3], o[1][3]);
    }
    __device__ void scalar(size_t i, int b, int c, float (&o)[2]) const { one(x[i], dy[i], b, c, o[0], o[1]); }
};

// the same with the tensor types fixed at compile time (see ld4_t)
template <int XT, int DYT>
struct GnBwdFT : GnBwdF {
    __device__ __forceinline__ void onej(float xv, float dyv, int j, float& s1, float& s2) const {
        const float xhat = (xv - cm[j]) * cr[j];
        const float dz = dyv * act_grad_fast(cA[j] * xv + cB[j], act);     // as gn_act_bwd_dx_kernel<true, XT, DYT> evaluates it
        s1 = dz; s2 = dz * xhat;
    }
    __device__ void vec4(size_t i, int, int, float (&o)[2][4]) const {
        const float4 xv = ld4_t<XT>(x, i, xty);
        const float4 dv = ld4_t<DYT>(dy, i, dyty);
        onej(xv.x, dv.x, 0, o[0][0], o[1][0]);
        onej(xv.y, dv.y, 1, o[0][1], o[1][1]);
        onej(xv.z, dv.z, 2, o[0][2], o[1][2]);
        onej(xv.w, dv.w, 3, o[0][3], o[1][3]);
    }
};

// pass 2a: S[b][v][c] = sum_blk partial[b][blk][v][c]  (sum_partials_kernel, one wave per output)
// pass 2b: parameter gradients and per-(b,g) means from S (thread-parallel, loops of length B or C/G)
__global__ __launch_bounds__(256) void gn_bwd_final_kernel(const float* __restrict__ S, GnCoef k,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                           float* __restrict__ dscale, float* __restrict__ dshift,
                                                           float* __restrict__ m12, int B, float inv_count) {
    const int C = k.C, G = k.G, Cg = C / G;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < B * C) {
        const int b = i / C, c = i % C;
        const float s1 = S[(size_t)b * 2 * C + c], s2 = S[(size_t)b * 2 * C + C + c];
        const float ga = k.gamma ? k.gamma[c] : 1.f, be = k.beta ? k.beta[c] : 0.f;
        if (dscale) dscale[b * k.cs + c] = ga * s2 + be * s1;
        if (dshift) dshift[b * k.cs + c] = s1;
    }
    if (i < C) {
        float dg = 0.f, db = 0.f;
        for (int b = 0; b < B; ++b) {
            const float sc = k.scale ? k.scale[b * k.cs + i] + 1.f : 1.f;
            dg += sc * S[(size_t)b * 2 * C + C + i];
            db += sc * S[(size_t)b * 2 * C + i];
        }
        if (dgamma) dgamma[i] = dg;
        if (dbeta) dbeta[i] = db;
    }
    if (i < B * G) {
        const int b = i / G, g = i % G;
        float a1 = 0.f, a2 = 0.f;
        for (int c = g * Cg; c < (g + 1) * Cg; ++c) {
            const float ga = k.gamma ? k.gamma[c] : 1.f;
            const float sc = k.scale ? k.scale[b * k.cs + c] + 1.f : 1.f;
            a1 += ga * sc * S[(size_t)b * 2 * C + c];
            a2 += ga * sc * S[(size_t)b * 2 * C + C + c];
        }
        m12[2 * i] = a1 * inv_count;
        m12[2 * i + 1] = a2 * inv_count;
    }
}

// pass 3: dx = rstd * (gamma*(scale+1)*dz - m1 - xhat*m2)
template <bool VEC, int XT = -1, int DYT = -1>            // XT (x and dx) / DYT >= 0: the tensor type at compile time (ld4_t)
__global__ __launch_bounds__(256) void gn_act_bwd_dx_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                            float* __restrict__ dx, GnCoef k,
                                                            const float* __restrict__ m12, int rows, int act,
                                                            const float* __restrict__ add, int xty = 0, int dxty = 0, int dyty = 0) {
    // xty / dxty / dyty (VEC only): x, dy read / dx written as fp32 (0), fp16 (1) or bf16 (2) values
    // add (optional, kernel-uniform): the gradient that reaches x through its OTHER consumer (the residual branch of a ResnetBlock),
    // summed here instead of in a separate pass over three tensors
    const int b = blockIdx.y, C = k.C, G = k.G, Cg = C / G;
    const size_t per = (size_t)rows * C;
    const float* xb = xty ? reinterpret_cast<const float*>(reinterpret_cast<const unsigned short*>(x) + (size_t)b * per) : x + (size_t)b * per;
    const float* dyb = dyty ? reinterpret_cast<const float*>(reinterpret_cast<const unsigned short*>(dy) + (size_t)b * per) : dy + (size_t)b * per;
    const float* adb = add ? add + (size_t)b * per : nullptr;
    float* dxb = dxty ? reinterpret_cast<float*>(reinterpret_cast<unsigned short*>(dx) + (size_t)b * per) : dx + (size_t)b * per;
    auto one = [&](float xv, float dyv, int c) -> float {
        float A, Bc;
        k.get(b, c, A, Bc);
        const int g = c / Cg;
        const float r = k.rstd[b * G + g];
        const float xhat = (xv - k.mean[b * G + g]) * r;
        const float dz = dyv * act_grad(A * xv + Bc, act);
        // A = r*gamma*(scale+1)  ->  gamma*(scale+1)*dz*r = A*dz
        return A * dz - r * (m12[2 * (b * G + g)] + xhat * m12[2 * (b * G + g) + 1]);
    };
    if (VEC) {
        // grid sized so a thread keeps its 4 channels (see gn_act_fwd_kernel): coefficients hoisted into registers
        const size_t n4 = per >> 2, i0 = blockIdx.x * (size_t)256 + threadIdx.x;
        const int c = (int)((i0 * 4) % C);
        float A[4], Bc[4], r[4], m[4], m1[4], m2[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            k.get(b, c + j, A[j], Bc[j]);
            const int g = (c + j) / Cg;
            r[j] = k.rstd[b * G + g]; m[j] = k.mean[b * G + g];
            m1[j] = m12[2 * (b * G + g)]; m2[j] = m12[2 * (b * G + g) + 1];
        }
        auto onej = [&](float xv, float dyv, int j) -> float {
            const float xhat = (xv - m[j]) * r[j];
            const float z = A[j] * xv + Bc[j];
            const float dz = dyv * ((XT >= 0 || DYT >= 0) ? act_grad_fast(z, act) : act_grad(z, act));
            return A[j] * dz - r[j] * (m1[j] + xhat * m2[j]);
        };
        const size_t st = (size_t)gridDim.x * 256;
        size_t i = i0;
        for (; i + st < n4; i += 2 * st) {              // 4 (6) loads in flight
            float4 xv[2], dv[2], av[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                xv[u] = ld4_t<XT>(xb, (i + u * st) * 4, xty);
                dv[u] = ld4_t<DYT>(dyb, (i + u * st) * 4, dyty);
                av[u] = adb ? *reinterpret_cast<const float4*>(adb + (i + u * st) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                float4 o;
                o.x = onej(xv[u].x, dv[u].x, 0) + av[u].x; o.y = onej(xv[u].y, dv[u].y, 1) + av[u].y;
                o.z = onej(xv[u].z, dv[u].z, 2) + av[u].z; o.w = onej(xv[u].w, dv[u].w, 3) + av[u].w;
                st4_t<XT>(dxb, (i + u * st) * 4, o, dxty);
            }
        }
        for (; i < n4; i += st) {
            const float4 xv = ld4_t<XT>(xb, i * 4, xty);
            const float4 dv = ld4_t<DYT>(dyb, i * 4, dyty);
            const float4 av = adb ? *reinterpret_cast<const float4*>(adb + i * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            float4 o;
            o.x = onej(xv.x, dv.x, 0) + av.x; o.y = onej(xv.y, dv.y, 1) + av.y;
            o.z = onej(xv.z, dv.z, 2) + av.z; o.w = onej(xv.w, dv.w, 3) + av.w;
            st4_t<XT>(dxb, i * 4, o, dxty);
        }
    } else {
        for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < per; i += (size_t)gridDim.x * 256)
            dxb[i] = one(xb[i], dyb[i], (int)(i % C)) + (adb ? adb[i] : 0.f);
    }
}

// ---------------------------------------------------------------------------------------------
// channel LayerNorm (per row over C), gain only
// ---------------------------------------------------------------------------------------------
// one wave per row; C <= 64*16
__global__ __launch_bounds__(256) void chan_ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                          const float* __restrict__ bb, const float* __restrict__ res,
                                                          float* __restrict__ y, float* __restrict__ mean,
                                                          float* __restrict__ rstd, int rows, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const int wpb = blockDim.x >> 6;
    for (int r = blockIdx.x * wpb + (threadIdx.x >> 6); r < rows; r += gridDim.x * wpb) {
        const float* xr = x + (size_t)r * C;
        float s = 0.f;
        for (int c = lane; c < C; c += 64) s += xr[c];
        const float m = wave_sum(s) / C;
        float v = 0.f;
        for (int c = lane; c < C; c += 64) { const float d = xr[c] - m; v += d * d; }
        const float rs = rsqrtf(wave_sum(v) / C + eps);
        for (int c = lane; c < C; c += 64) {
            float v = (xr[c] - m) * rs * g[c] + (bb ? bb[c] : 0.f);
            if (res) v += res[(size_t)r * C + c];          // the caller's `LN(x) + residual` (attention blocks) in the same pass
            y[(size_t)r * C + c] = v;
        }
        if (lane == 0) { if (mean) mean[r] = m; if (rstd) rstd[r] = rs; }
    }
}

// dx = rstd*(dxhat - mean(dxhat) - xhat*mean(dxhat*xhat)), dxhat = dy*g ; dg partial via colreduce
__global__ __launch_bounds__(256) void chan_ln_bwd_dx_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                             const float* __restrict__ g, const float* __restrict__ mean,
                                                             const float* __restrict__ rstd, float* __restrict__ dx,
                                                             int rows, int C, const float* __restrict__ add) {
    // add (optional): the gradient reaching x through its other consumer (`LN(x) -> fn -> + x` blocks), summed in this pass
    const int lane = threadIdx.x & 63;
    const int wpb = blockDim.x >> 6;
    for (int r = blockIdx.x * wpb + (threadIdx.x >> 6); r < rows; r += gridDim.x * wpb) {
        const float* xr = x + (size_t)r * C;
        const float* dr = dy + (size_t)r * C;
        const float m = mean[r], rs = rstd[r];
        float a = 0.f, bsum = 0.f;
        for (int c = lane; c < C; c += 64) {
            const float dxh = dr[c] * g[c], xh = (xr[c] - m) * rs;
            a += dxh; bsum += dxh * xh;
        }
        a = wave_sum(a) / C; bsum = wave_sum(bsum) / C;
        for (int c = lane; c < C; c += 64) {
            const float dxh = dr[c] * g[c], xh = (xr[c] - m) * rs;
            float v = rs * (dxh - a - xh * bsum);
            if (add) v += add[(size_t)r * C + c];
            dx[(size_t)r * C + c] = v;
        }
    }
}
struct ChanLnDgF {
    const float *x, *dy, *mean, *rstd;
    int C;
    __device__ __forceinline__ float one(size_t i) const {
        const size_t r = i / C;
        return dy[i] * (x[i] - mean[r]) * rstd[r];
    }
    __device__ void prep(int, int) {}
    __device__ void vec4(size_t i, int, int, float (&o)[1][4]) const {
        o[0][0] = one(i); o[0][1] = one(i + 1); o[0][2] = one(i + 2); o[0][3] = one(i + 3);
    }
    __device__ void scalar(size_t i, int, int, float (&o)[1]) const { o[0] = one(i); }
};
__global__ __launch_bounds__(256) void sum_partials_kernel(const float* __restrict__ partial, float* __restrict__ out,
                                                           int nblk, int NVC, int total, float alpha) {
    // out[b][j] = alpha * sum_k partial[b][k][j]; one wave per output, lanes over k
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= total) return;
    const int b = i / NVC, j = i % NVC, lane = threadIdx.x & 63;
    float s = 0.f;
    for (int k = lane; k < nblk; k += 64) s += partial[((size_t)b * nblk + k) * NVC + j];
    s = wave_sum(s);
    if (lane == 0) out[i] = alpha * s;
}

// ---------------------------------------------------------------------------------------------
// elementwise activation
// ---------------------------------------------------------------------------------------------
template <bool VEC>
__global__ __launch_bounds__(256) void act_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n, int act) {
    if (VEC) {
        const size_t n4 = n >> 2;
        for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
            float4 v = *reinterpret_cast<const float4*>(x + i * 4);
            v.x = act_fwd(v.x, act); v.y = act_fwd(v.y, act); v.z = act_fwd(v.z, act); v.w = act_fwd(v.w, act);
            *reinterpret_cast<float4*>(y + i * 4) = v;
        }
        for (size_t i = (n4 << 2) + blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
            y[i] = act_fwd(x[i], act);
    } else {
        for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) y[i] = act_fwd(x[i], act);
    }
}
template <bool VEC>
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                      float* __restrict__ dx, size_t n, int act) {
    if (VEC) {
        const size_t n4 = n >> 2;
        for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
            const float4 v = *reinterpret_cast<const float4*>(x + i * 4);
            float4 d = *reinterpret_cast<const float4*>(dy + i * 4);
            d.x *= act_grad(v.x, act); d.y *= act_grad(v.y, act); d.z *= act_grad(v.z, act); d.w *= act_grad(v.w, act);
            *reinterpret_cast<float4*>(dx + i * 4) = d;
        }
        for (size_t i = (n4 << 2) + blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
            dx[i] = dy[i] * act_grad(x[i], act);
    } else {
        for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
            dx[i] = dy[i] * act_grad(x[i], act);
    }
}

// ---------------------------------------------------------------------------------------------
// squeeze-excite helpers
// ---------------------------------------------------------------------------------------------
struct IdentF {
    const float* x;
    __device__ void prep(int, int) {}
    __device__ void vec4(size_t i, int, int, float (&o)[1][4]) const {
        const float4 v = *reinterpret_cast<const float4*>(x + i);
        o[0][0] = v.x; o[0][1] = v.y; o[0][2] = v.z; o[0][3] = v.w;
    }
    __device__ void scalar(size_t i, int, int, float (&o)[1]) const { o[0] = x[i]; }
};
struct WeightedF {      // x[b][row][c] * w[b][row]  (GlobalContext pooling, imagen_video.py:975-979)
    const float *x, *w;
    int C;
    __device__ void prep(int, int) {}
    __device__ void vec4(size_t i, int, int, float (&o)[1][4]) const {
        const float4 v = *reinterpret_cast<const float4*>(x + i);
        const float ww = w[i / C];
        o[0][0] = v.x * ww; o[0][1] = v.y * ww; o[0][2] = v.z * ww; o[0][3] = v.w * ww;
    }
    __device__ void scalar(size_t i, int, int, float (&o)[1]) const { o[0] = x[i] * w[i / C]; }
};
struct ProdF {
    const float *a, *b;
    int bty = 0;                           // b holds fp32 (0), fp16 (1) or bf16 (2) values
    __device__ void prep(int, int) {}
    __device__ void vec4(size_t i, int, int, float (&o)[1][4]) const {
        const float4 u = *reinterpret_cast<const float4*>(a + i);
        const float4 v = ld4_any(b, i, bty);
        o[0][0] = u.x * v.x; o[0][1] = u.y * v.y; o[0][2] = u.z * v.z; o[0][3] = u.w * v.w;
    }
    __device__ void scalar(size_t i, int, int, float (&o)[1]) const { o[0] = a[i] * b[i]; }
};

// y = h*gate[b][c] (+ res)  |  mode 1: y = h*gate (dh = dy*gate uses the same kernel with res = NULL)
template <bool VEC>
__global__ __launch_bounds__(256) void gate_residual_kernel(const float* __restrict__ h, const float* __restrict__ gate,
                                                            const float* __restrict__ res, const float* __restrict__ addc,
                                                            float alpha, float* __restrict__ y, int rows, int C, int hty = 0, int yty = 0) {
    // hty / yty (VEC only): h read / y written as fp32 (0), fp16 (1) or bf16 (2) values
    const int b = blockIdx.y;
    const size_t per = (size_t)rows * C;
    const float* hb = hty ? reinterpret_cast<const float*>(reinterpret_cast<const unsigned short*>(h) + (size_t)b * per) : h + (size_t)b * per;
    const float* rb = res ? res + (size_t)b * per : nullptr;
    const float* gb = gate + (size_t)b * C;
    const float* ab = addc ? addc + (size_t)b * C : nullptr;
    float* yb = yty ? reinterpret_cast<float*>(reinterpret_cast<unsigned short*>(y) + (size_t)b * per) : y + (size_t)b * per;
    if (VEC) {
        const size_t n4 = per >> 2;
        for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
            const int c = (int)((i * 4) % C);
            float4 v = ld4_any(hb, i * 4, hty);
            const float4 gq = *reinterpret_cast<const float4*>(gb + c);
            v.x *= gq.x; v.y *= gq.y; v.z *= gq.z; v.w *= gq.w;
            if (ab) {
                const float4 a = *reinterpret_cast<const float4*>(ab + c);
                v.x += alpha * a.x; v.y += alpha * a.y; v.z += alpha * a.z; v.w += alpha * a.w;
            }
            if (rb) {
                const float4 r = *reinterpret_cast<const float4*>(rb + i * 4);
                v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
            }
            st4_any(yb, i * 4, v, yty);
        }
    } else {
        for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < per; i += (size_t)gridDim.x * 256) {
            float v = hb[i] * gb[i % C];
            if (ab) v += alpha * ab[i % C];
            if (rb) v += rb[i];
            yb[i] = v;
        }
    }
}

// y = h*gate[b][c] + res AND per-workgroup column sums (sum, sum of squares) of y for the consumer's GroupNorm statistics
// (ResnetBlock output -> the next block's first GroupNorm, imagen_pytorch3D.py:568-614): stats[b][blk][2][C].
// 1024 % C == 0, so a thread's channel quad is the same in every grid-stride iteration (stride = gridDim.x * 1024 floats) and
// the threads t, t + C/4, ... of a workgroup share it: fixed-order combine through LDS, one partial row per workgroup.
__global__ __launch_bounds__(256) void gate_residual_stats_kernel(const float* __restrict__ h, const float* __restrict__ gate,
                                                                  const float* __restrict__ res, float* __restrict__ y,
                                                                  float* __restrict__ stats, int rows, int C, int hty = 0) {
    __shared__ float red[256 * 8];
    const int b = blockIdx.y, t = threadIdx.x;
    const size_t per = (size_t)rows * C, n4 = per >> 2;
    const float* hb = hty ? reinterpret_cast<const float*>(reinterpret_cast<const unsigned short*>(h) + (size_t)b * per) : h + (size_t)b * per;
    const float* rb = res ? res + (size_t)b * per : nullptr;
    float* yb = y + (size_t)b * per;
    const int c = (t * 4) % C;
    const float4 gq = *reinterpret_cast<const float4*>(gate + (size_t)b * C + c);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, q0 = 0.f, q1 = 0.f, q2 = 0.f, q3 = 0.f;
    for (size_t i = blockIdx.x * (size_t)256 + t; i < n4; i += (size_t)gridDim.x * 256) {
        float4 v = ld4_any(hb, i * 4, hty);
        v.x *= gq.x; v.y *= gq.y; v.z *= gq.z; v.w *= gq.w;
        if (rb) {
            const float4 r = *reinterpret_cast<const float4*>(rb + i * 4);
            v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
        }
        *reinterpret_cast<float4*>(yb + i * 4) = v;
        s0 += v.x; s1 += v.y; s2 += v.z; s3 += v.w;
        q0 = fmaf(v.x, v.x, q0); q1 = fmaf(v.y, v.y, q1); q2 = fmaf(v.z, v.z, q2); q3 = fmaf(v.w, v.w, q3);
    }
    float* mine = red + t * 8;
    mine[0] = s0; mine[1] = s1; mine[2] = s2; mine[3] = s3; mine[4] = q0; mine[5] = q1; mine[6] = q2; mine[7] = q3;
    __syncthreads();
    const int nq = C / 4;                         // channel quads; nq divides 256
    if (t < nq) {
        float a[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] = red[t * 8 + k];
        for (int u = t + nq; u < 256; u += nq)
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] += red[u * 8 + k];
        float* o = stats + (((size_t)b * gridDim.x + blockIdx.x) * 2) * C + 4 * t;
        *reinterpret_cast<float4*>(o) = make_float4(a[0], a[1], a[2], a[3]);
        *reinterpret_cast<float4*>(o + C) = make_float4(a[4], a[5], a[6], a[7]);
    }
}

// SE3D.fc on [B][C] (single block: B*C*Cr MACs is a few 100k at most)
__global__ __launch_bounds__(256) void se_mlp_fwd_kernel(const float* __restrict__ pooled, const float* __restrict__ w1,
                                                         const float* __restrict__ w2, float* __restrict__ hidden,
                                                         float* __restrict__ gate, int B, int C, int Cr) {
    for (int i = threadIdx.x; i < B * Cr; i += 256) {
        const int b = i / Cr, r = i % Cr;
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += w1[r * C + c] * pooled[b * C + c];
        hidden[i] = fmaxf(s, 0.f);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < B * C; i += 256) {
        const int b = i / C, c = i % C;
        float s = 0.f;
        for (int r = 0; r < Cr; ++r) s += w2[c * Cr + r] * hidden[b * Cr + r];
        gate[i] = 1.f / (1.f + __expf(-s));
    }
}
// SE3D squeeze + excitation for ONE batch entry per workgroup (imagen_pytorch3D.py:617-632): channel means finished from the
// producer's per-tile column sums (or taken from `pooled_in`), then fc1 -> ReLU -> fc2 -> sigmoid.  The single-workgroup kernel
// above spends 10 us per call on 32 threads doing 64-long strided dot products; a sampler step has 19 of them.
__global__ __launch_bounds__(256) void se_pool_mlp_fwd_kernel(const float* __restrict__ partials, int nblk, float alpha,
                                                              const float* __restrict__ pooled_in, const float* __restrict__ w1,
                                                              const float* __restrict__ w2, float* __restrict__ pooled,
                                                              float* __restrict__ hidden, float* __restrict__ gate, int C, int Cr) {
    extern __shared__ float sm[];          // [256] scratch | [C] pooled | [Cr] hidden
    float* red = sm;
    float* pl = sm + 256;
    float* hid = pl + C;
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (partials) {                        // fixed order: row group rg of channel c sums tiles rg, rg + RG, ...; groups combined in order
        for (int c0 = 0; c0 < C; c0 += 256) {
            const int cw = min(256, C - c0), RG = 256 / cw;              // cw divides 256 whenever C does or is a multiple of 256
            const int c = c0 + t % cw, rg = t / cw;
            // eight rows in flight per thread (a 64^3 volume hands over 512-2048 tile rows: one dependent load at a time was 100 us per
            // call on C4, 19 calls per eval), combined in a fixed order
            float a8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (rg < RG) {
                int k = rg;
                for (; k + 7 * RG < nblk; k += 8 * RG) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) a8[u] += partials[(((size_t)b * nblk + k + u * RG) * 2) * C + c];
                }
                for (int u = 0; k < nblk; k += RG, ++u) a8[u] += partials[(((size_t)b * nblk + k) * 2) * C + c];
            }
            const float a = ((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]));
            red[t] = a;
            __syncthreads();
            if (t < cw) {
                float v = red[t];
                for (int r = 1; r < RG; ++r) v += red[r * cw + t];
                pl[c0 + t] = alpha * v;
                pooled[(size_t)b * C + c0 + t] = alpha * v;
            }
            __syncthreads();
        }
    } else {
        for (int c = t; c < C; c += 256) pl[c] = pooled_in[(size_t)b * C + c];
        __syncthreads();
    }
    for (int r = wave; r < Cr; r += 4) {   // one wave per hidden unit: lane-strided dot product, butterfly sum
        float a = 0.f;
        for (int c = lane; c < C; c += 64) a = fmaf(w1[(size_t)r * C + c], pl[c], a);
        a = wave_sum(a);
        if (lane == 0) { const float hv = fmaxf(a, 0.f); hid[r] = hv; hidden[(size_t)b * Cr + r] = hv; }
    }
    __syncthreads();
    for (int c = t; c < C; c += 256) {
        float a = 0.f;
        for (int r = 0; r < Cr; ++r) a = fmaf(w2[(size_t)c * Cr + r], hid[r], a);
        gate[(size_t)b * C + c] = 1.f / (1.f + __expf(-a));
    }
}
// Backward of the two-layer SE gate MLP.  One workgroup (the problem is B x C numbers); LDS = true stages every operand in LDS first:
// from global memory the dependent sums below are chains of up to C = 256 L2 round trips (22 us per launch, 19 launches per training
// micro-step of C2); from LDS the same loops take a third of that.
template <bool LDS>
__global__ __launch_bounds__(256) void se_mlp_bwd_kernel(const float* __restrict__ pooled, const float* __restrict__ w1,
                                                         const float* __restrict__ w2, const float* __restrict__ hidden,
                                                         const float* __restrict__ gate, const float* __restrict__ dgate,
                                                         float* __restrict__ dpooled, float* __restrict__ dw1,
                                                         float* __restrict__ dw2, float* __restrict__ scratch, int B,
                                                         int C, int Cr) {
    extern __shared__ __attribute__((aligned(16))) float se_sm[];
    float* dz2 = LDS ? se_sm : scratch;                       // [B][C]
    float* dhid = dz2 + B * C;                                // [B][Cr]
    const float *pl = pooled, *hd = hidden, *w1p = w1, *w2p = w2;
    if (LDS) {
        float* hs = dhid + B * Cr;                            // hidden [B][Cr], pooled [B][C], w1 [Cr][C], w2 [C][Cr]
        float* ps = hs + B * Cr;
        float* w1s = ps + B * C;
        float* w2s = w1s + Cr * C;
        for (int i = threadIdx.x; i < B * Cr; i += 256) hs[i] = hidden[i];
        for (int i = threadIdx.x; i < B * C; i += 256) ps[i] = pooled[i];
        for (int i = threadIdx.x; i < C * Cr; i += 256) { w1s[i] = w1[i]; w2s[i] = w2[i]; }
        pl = ps; hd = hs; w1p = w1s; w2p = w2s;
    }
    for (int i = threadIdx.x; i < B * C; i += 256) { const float g = gate[i]; dz2[i] = dgate[i] * g * (1.f - g); }
    __syncthreads();
    for (int i = threadIdx.x; i < C * Cr; i += 256) {
        const int c = i / Cr, r = i % Cr;
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += dz2[b * C + c] * hd[b * Cr + r];
        dw2[i] = s;
    }
    for (int i = threadIdx.x; i < B * Cr; i += 256) {
        const int b = i / Cr, r = i % Cr;
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += dz2[b * C + c] * w2p[c * Cr + r];
        dhid[i] = hd[i] > 0.f ? s : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < Cr * C; i += 256) {
        const int r = i / C, c = i % C;
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += dhid[b * Cr + r] * pl[b * C + c];
        dw1[i] = s;
    }
    for (int i = threadIdx.x; i < B * C; i += 256) {
        const int b = i / C, c = i % C;
        float s = 0.f;
        for (int r = 0; r < Cr; ++r) s += dhid[b * Cr + r] * w1p[r * C + c];
        dpooled[i] = s;
    }
}

__global__ __launch_bounds__(256) void add_channel_broadcast_kernel(float* __restrict__ x, const float* __restrict__ v,
                                                                    float alpha, int rows, int C) {
    const int b = blockIdx.y;
    const size_t per = (size_t)rows * C;
    float* xb = x + (size_t)b * per;
    const float* vb = v + (size_t)b * C;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < per; i += (size_t)gridDim.x * 256)
        xb[i] += alpha * vb[i % C];
}

// ---------------------------------------------------------------------------------------------
// data movement
// ---------------------------------------------------------------------------------------------
// toSpace = 0: y[b][d][h][w][c*8 + s1*4+s2*2+s3] = x[b][2d+s1][2h+s2][2w+s3][c]; toSpace = 1: inverse
__global__ __launch_bounds__(256) void shuffle2_kernel(const float* __restrict__ src, float* __restrict__ dst, int B,
                                                       int D, int H, int W, int C, int toSpace) {
    // iterate over the "space" side elements (B,2D,2H,2W,C), c fastest -> coalesced on that side
    const size_t total = (size_t)B * 8 * D * H * W * C;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        size_t r = i / C;
        const int w2 = (int)(r % (2 * W)); r /= 2 * W;
        const int h2 = (int)(r % (2 * H)); r /= 2 * H;
        const int d2 = (int)(r % (2 * D));
        const int b = (int)(r / (2 * D));
        const int s = ((d2 & 1) << 2) | ((h2 & 1) << 1) | (w2 & 1);
        const size_t j = ((((size_t)b * D + (d2 >> 1)) * H + (h2 >> 1)) * W + (w2 >> 1)) * (8 * (size_t)C) + c * 8 + s;
        if (toSpace) dst[i] = src[j]; else dst[j] = src[i];
    }
}

__global__ __launch_bounds__(256) void concat_kernel(const float* __restrict__ a, int Ca, const float* __restrict__ b,
                                                     int Cb, float* __restrict__ y, size_t rows) {
    const int C = Ca + Cb;
    const size_t total = rows * C;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t r = i / C;
        const int c = (int)(i % C);
        y[i] = c < Ca ? a[r * Ca + c] : b[r * Cb + (c - Ca)];
    }
}
// float4 form with a scalar factor per input (the skip connections of the pseudo-3D U-Net: cat(x, skip * 2^-0.5), imagen_video.py:1743)
__global__ __launch_bounds__(256) void concat4_kernel(const float4* __restrict__ a, int Ca4, const float4* __restrict__ b, int Cb4,
                                                      float sa, float sb, float4* __restrict__ y, size_t rows) {
    const int C4 = Ca4 + Cb4;
    const size_t total = rows * C4;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t r = i / C4;
        const int c = (int)(i - r * C4);
        const bool first = c < Ca4;
        float4 v = first ? a[r * Ca4 + c] : b[r * Cb4 + (c - Ca4)];
        const float f = first ? sa : sb;
        v.x *= f; v.y *= f; v.z *= f; v.w *= f;
        y[i] = v;
    }
}
// The same pass with the column sums (sum, sum of squares) of the rows it writes, per workgroup: [B][nblk][2][C] -- the concatenated
// tensor's consumer is a GroupNorm (the ResnetBlocks of the up path), whose statistics pass over the tensor disappears (15 launches of
// colreduce_kernel<MomentsF> per C5 stage-2 eval: 1.15 ms).  Workgroup (blockIdx.x, b) owns rows [blk * rpb, (blk + 1) * rpb) of batch
// entry b; thread t < RL * C4: channel quad t % C4, row lane t / C4; fixed-order combine of the row lanes through LDS: deterministic.
__global__ __launch_bounds__(256) void concat4_stats_kernel(const float4* __restrict__ a, int Ca4, const float4* __restrict__ b, int Cb4,
                                                            float sa, float sb, float4* __restrict__ y, int rowsPerBatch, int rpb,
                                                            float* __restrict__ stats, int nblk) {
    extern __shared__ float4 cs_sm[];          // [RL][C4] sums | [RL][C4] sums of squares
    const int C4 = Ca4 + Cb4, RL = 256 / C4;
    const int t = threadIdx.x, c = t % C4, rl = t / C4;
    const int bb = blockIdx.y, blk = blockIdx.x;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f), q = make_float4(0.f, 0.f, 0.f, 0.f);
    if (rl < RL) {
        const bool first = c < Ca4;
        const float f = first ? sa : sb;
        const int r1 = min(rowsPerBatch, (blk + 1) * rpb);
        for (int r = blk * rpb + rl; r < r1; r += RL) {
            const size_t row = (size_t)bb * rowsPerBatch + r;
            float4 v = first ? a[row * Ca4 + c] : b[row * Cb4 + (c - Ca4)];
            v.x *= f; v.y *= f; v.z *= f; v.w *= f;
            y[row * C4 + c] = v;
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            q.x = fmaf(v.x, v.x, q.x); q.y = fmaf(v.y, v.y, q.y); q.z = fmaf(v.z, v.z, q.z); q.w = fmaf(v.w, v.w, q.w);
        }
        cs_sm[rl * C4 + c] = s;
        cs_sm[(RL + rl) * C4 + c] = q;
    }
    __syncthreads();
    if (t < 2 * C4) {
        const int which = t / C4, cc = t % C4;
        float4 acc = cs_sm[(which * RL) * C4 + cc];
        for (int k = 1; k < RL; ++k) { const float4 v = cs_sm[(which * RL + k) * C4 + cc]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
        reinterpret_cast<float4*>(stats + (((size_t)bb * nblk + blk) * 2 + which) * (size_t)(4 * C4))[cc] = acc;
    }
}
__global__ __launch_bounds__(256) void split4_kernel(const float4* __restrict__ y, float4* __restrict__ a, int Ca4, float4* __restrict__ b,
                                                     int Cb4, float sa, float sb, size_t rows) {
    const int C4 = Ca4 + Cb4;
    const size_t total = rows * C4;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t r = i / C4;
        const int c = (int)(i - r * C4);
        const bool first = c < Ca4;
        if ((first && !a) || (!first && !b)) continue;
        float4 v = y[i];
        const float f = first ? sa : sb;
        v.x *= f; v.y *= f; v.z *= f; v.w *= f;
        if (first) a[r * Ca4 + c] = v; else b[r * Cb4 + (c - Ca4)] = v;
    }
}
__global__ __launch_bounds__(256) void concat_scaled_kernel(const float* __restrict__ a, int Ca, const float* __restrict__ b, int Cb,
                                                            float sa, float sb, float* __restrict__ y, size_t rows) {
    const int C = Ca + Cb;
    const size_t total = rows * C;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t r = i / C;
        const int c = (int)(i % C);
        y[i] = c < Ca ? sa * a[r * Ca + c] : sb * b[r * Cb + (c - Ca)];
    }
}
__global__ __launch_bounds__(256) void split_scaled_kernel(const float* __restrict__ y, float* __restrict__ a, int Ca, float* __restrict__ b,
                                                           int Cb, float sa, float sb, size_t rows) {
    const int C = Ca + Cb;
    const size_t total = rows * C;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t r = i / C;
        const int c = (int)(i % C);
        if (c < Ca) { if (a) a[r * Ca + c] = sa * y[i]; } else { if (b) b[r * Cb + (c - Ca)] = sb * y[i]; }
    }
}
__global__ __launch_bounds__(256) void split_kernel(const float* __restrict__ y, float* __restrict__ a, int Ca,
                                                    float* __restrict__ b, int Cb, size_t rows) {
    const int C = Ca + Cb;
    const size_t total = rows * C;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t r = i / C;
        const int c = (int)(i % C);
        if (c < Ca) { if (a) a[r * Ca + c] = y[i]; } else { if (b) b[r * Cb + (c - Ca)] = y[i]; }
    }
}

// trilinear up-sampling, align_corners=True: src coord = o * (I-1)/(O-1)
__global__ __launch_bounds__(256) void trilinear_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int D,
                                                        int H, int W, int C, int scale, int adjoint) {
    const int Do = D * scale, Ho = H * scale, Wo = W * scale;
    const float rd = Do > 1 ? (float)(D - 1) / (Do - 1) : 0.f, rh = Ho > 1 ? (float)(H - 1) / (Ho - 1) : 0.f,
                rw = Wo > 1 ? (float)(W - 1) / (Wo - 1) : 0.f;
    const size_t total = (size_t)B * Do * Ho * Wo * C;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        size_t r = i / C;
        const int ow = (int)(r % Wo); r /= Wo;
        const int oh = (int)(r % Ho); r /= Ho;
        const int od = (int)(r % Do);
        const int b = (int)(r / Do);
        const float fd = od * rd, fh = oh * rh, fw = ow * rw;
        const int d0 = (int)fd, h0 = (int)fh, w0 = (int)fw;
        const int d1 = d0 + (d0 < D - 1), h1 = h0 + (h0 < H - 1), w1 = w0 + (w0 < W - 1);
        const float ld = fd - d0, lh = fh - h0, lw = fw - w0;
        const float* sb = src + (size_t)b * D * H * W * C;
        float* db_ = dst + (size_t)b * D * H * W * C;
#define IDX(d, h, w) ((((size_t)(d) * H + (h)) * W + (w)) * C + c)
        if (!adjoint) {
            const float v = (1 - ld) * ((1 - lh) * ((1 - lw) * sb[IDX(d0, h0, w0)] + lw * sb[IDX(d0, h0, w1)]) +
                                        lh * ((1 - lw) * sb[IDX(d0, h1, w0)] + lw * sb[IDX(d0, h1, w1)])) +
                            ld * ((1 - lh) * ((1 - lw) * sb[IDX(d1, h0, w0)] + lw * sb[IDX(d1, h0, w1)]) +
                                  lh * ((1 - lw) * sb[IDX(d1, h1, w0)] + lw * sb[IDX(d1, h1, w1)]));
            dst[i] = v;
        } else {
            const float g = src[i];
            atomicAdd(db_ + IDX(d0, h0, w0), g * (1 - ld) * (1 - lh) * (1 - lw));
            atomicAdd(db_ + IDX(d0, h0, w1), g * (1 - ld) * (1 - lh) * lw);
            atomicAdd(db_ + IDX(d0, h1, w0), g * (1 - ld) * lh * (1 - lw));
            atomicAdd(db_ + IDX(d0, h1, w1), g * (1 - ld) * lh * lw);
            atomicAdd(db_ + IDX(d1, h0, w0), g * ld * (1 - lh) * (1 - lw));
            atomicAdd(db_ + IDX(d1, h0, w1), g * ld * (1 - lh) * lw);
            atomicAdd(db_ + IDX(d1, h1, w0), g * ld * lh * (1 - lw));
            atomicAdd(db_ + IDX(d1, h1, w1), g * ld * lh * lw);
        }
#undef IDX
    }
}

// sub[n][A'][A'][A'][C] <-> vol[fA][fA][fA][C], n = b2 + f*b3 + f*f*b4 (axis 2 fastest), A' = A + 2*halo
__global__ __launch_bounds__(256) void subvolume_kernel(const float* __restrict__ src, float* __restrict__ dst, int f,
                                                        int A, int C, int halo, int scatter, int accumulate) {
    const int Ap = A + 2 * halo, S = f * A;
    const size_t total = (size_t)f * f * f * Ap * Ap * Ap * C;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        size_t r = i / C;
        const int z = (int)(r % Ap); r /= Ap;
        const int yy = (int)(r % Ap); r /= Ap;
        const int xx = (int)(r % Ap);
        const int n = (int)(r / Ap);
        const int b2 = n % f, b3 = (n / f) % f, b4 = n / (f * f);
        const int gx = b2 * A + xx - halo, gy = b3 * A + yy - halo, gz = b4 * A + z - halo;
        const bool in = gx >= 0 && gx < S && gy >= 0 && gy < S && gz >= 0 && gz < S;
        const size_t j = (((size_t)gx * S + gy) * S + gz) * C + c;
        if (!scatter) dst[i] = in ? src[j] : 0.f;
        else if (in) {
            if (accumulate) atomicAdd(dst + j, src[i]); else dst[j] = src[i];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// diffusion step math
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void axpby3_kernel(const float* __restrict__ a, const float* __restrict__ b_,
                                                     const float* __restrict__ c_, const float* __restrict__ c0,
                                                     const float* __restrict__ c1, const float* __restrict__ c2,
                                                     float lo, float hi, int clamp_mode, float* __restrict__ out,
                                                     size_t per) {
    const int b = blockIdx.y;
    const float k0 = c0[b], k1 = (b_ && c1) ? c1[b] : 0.f, k2 = (c_ && c2) ? c2[b] : 0.f;
    const size_t o = (size_t)b * per;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < per; i += (size_t)gridDim.x * 256) {
        float v = k0 * a[o + i];
        if (b_) v += k1 * b_[o + i];
        if (c_) v += k2 * c_[o + i];
        if (clamp_mode == 1) v = fmaxf(v, lo);
        else if (clamp_mode == 2) v = fminf(fmaxf(v, lo), hi);
        out[o + i] = v;
    }
}

__global__ __launch_bounds__(256) void ddpm_step_kernel(const float* __restrict__ x_t, const float* __restrict__ pred,
                                                        const float* __restrict__ noise, const float* __restrict__ ca,
                                                        const float* __restrict__ cb, const float* __restrict__ cn,
                                                        float lo, float hi, int clamp_mode, float* __restrict__ x_next,
                                                        float* __restrict__ x0_out, size_t per) {
    const int b = blockIdx.y;
    const float ka = ca[b], kb = cb[b], kn = cn[b];
    const size_t o = (size_t)b * per;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < per; i += (size_t)gridDim.x * 256) {
        float x0 = pred[o + i];
        x0 = clamp_mode == 0 ? fmaxf(x0, lo) : fminf(fmaxf(x0, lo), hi);
        if (x0_out) x0_out[o + i] = x0;
        x_next[o + i] = ka * x_t[o + i] + kb * x0 + kn * noise[o + i];
    }
}

// kind: 0 = squared error (F.mse_loss), 1 = absolute error (F.l1_loss), 2 = Huber with beta 1 (F.smooth_l1_loss): the three
// `loss_type`s of Imagen.__init__ (imagen_pytorch3D.py:1785-1790)
__device__ __forceinline__ float loss_term(float d, int kind) {
    if (kind == 1) return fabsf(d);
    if (kind == 2) { const float a = fabsf(d); return a < 1.f ? 0.5f * d * d : a - 0.5f; }
    return d * d;
}
__device__ __forceinline__ float loss_term_grad(float d, int kind) {      // d loss_term / d d
    if (kind == 1) return d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
    if (kind == 2) return fabsf(d) < 1.f ? d : (d > 0.f ? 1.f : -1.f);
    return 2.f * d;
}
__global__ __launch_bounds__(256) void mse_clamp_fwd_kernel(const float* pred, float* pred_out, const float* __restrict__ target,
                                                            const float* __restrict__ w, float lo, int do_clamp,
                                                            float* __restrict__ partials, int B, size_t per, int kind) {
    __shared__ float sh[4];
    const size_t total = (size_t)B * per;
    float s = 0.f;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        float p = pred[i];
        if (do_clamp) p = fmaxf(p, lo);
        if (pred_out) pred_out[i] = p;
        const float d = p - target[i];
        s += (w ? w[i / per] : 1.f) * loss_term(d, kind);
    }
    s = block_sum256(s, sh);
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void mse_final_kernel(const float* __restrict__ partials, int n, float inv,
                                                        float* __restrict__ out) {
    __shared__ float sh[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += partials[i];
    s = block_sum256(s, sh);
    if (threadIdx.x == 0) *out = s * inv;
}
__global__ __launch_bounds__(256) void mse_clamp_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                            const float* __restrict__ w, float lo, int do_clamp,
                                                            float coef, float* __restrict__ dpred, int B, size_t per, int kind) {
    const size_t total = (size_t)B * per;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const float p = pred[i];
        // the reference clamps in place (imagen_pytorch3D.py:2362): no gradient where pred was floored
        // (kind 0: coef carries the 2 of d(d^2), and 2 * (0.5 d) is exact, so this is the old expression bit for bit)
        float gq = coef * (w ? w[i / per] : 1.f) * (0.5f * loss_term_grad(p - target[i], kind));
        if (do_clamp && !(p > lo)) gq = 0.f;
        dpred[i] = gq;
    }
}

// ---------------------------------------------------------------------------------------------
// optimiser
// ---------------------------------------------------------------------------------------------
// global gradient-norm clipping (accelerator.clip_grad_norm_ = torch.nn.utils.clip_grad_norm_, trainer.py:1054) over the flat gradient
// arena: stage 1 = per-workgroup sums of squares in a fixed order (fp32 per thread, fp64 across the workgroup), stage 2 = one workgroup
// sums the partials in fp64 and writes out[0] = total L2 norm, out[1] = min(1, max_norm / (norm + 1e-6)) -- the factor adam_kernel
// multiplies every gradient with (torch multiplies the gradients in place; they are consumed and zeroed by the same Adam pass here)
constexpr int kNormBlocks = 1024;
__global__ __launch_bounds__(256) void gradnorm_stage1_kernel(const float* __restrict__ g, size_t n, double* __restrict__ part) {
    __shared__ double sh[4];
    float acc = 0.f;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc = fmaf(g[i], g[i], acc);
    double d = (double)acc;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = d;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}
__global__ __launch_bounds__(256) void gradnorm_stage2_kernel(const double* __restrict__ part, int nb, float max_norm, float* __restrict__ out) {
    __shared__ double sh[4];
    double d = 0.0;
    for (int i = threadIdx.x; i < nb; i += 256) d += part[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = d;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float norm = (float)sqrt((sh[0] + sh[1]) + (sh[2] + sh[3]));
        out[0] = norm;
        out[1] = fminf(max_norm / (norm + 1e-6f), 1.f);
    }
}
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, size_t n, float lr, float b1, float b2,
                                                   float eps, float wd, float bc1, float bc2_sqrt, int zero_grad,
                                                   const float* __restrict__ gscale) {
    const float step = lr / bc1;
    const float gs = gscale ? gscale[0] : 1.f;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float gi = g[i];
        if (gscale) gi *= gs;
        const float pi = p[i];
        if (wd != 0.f) gi += wd * pi;
        const float mi = m[i] + (gi - m[i]) * (1.f - b1);          // lerp, as torch's exp_avg.lerp_
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        p[i] = pi - step * mi / (sqrtf(vi) / bc2_sqrt + eps);
        if (zero_grad) g[i] = 0.f;
    }
}
// gradient accumulation into the flat arena: dst[off_t + i] += src_t[i] for every tensor t of a table
// table[t] = {src pointer, dst offset (elements), n (elements)} as 3 x int64; grid = (blocks per tensor, tensors)
__global__ __launch_bounds__(256) void multi_accumulate_kernel(float* __restrict__ dst, const long long* __restrict__ table) {
    const long long* e = table + 3 * (size_t)blockIdx.y;
    const float* __restrict__ src = reinterpret_cast<const float*>(e[0]);
    float* __restrict__ d = dst + e[1];
    const size_t n = (size_t)e[2];
    if (((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(d)) & 15u) == 0) {
        const size_t n4 = n >> 2;
        for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
            float4 a = reinterpret_cast<float4*>(d)[i];
            const float4 b = reinterpret_cast<const float4*>(src)[i];
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
            reinterpret_cast<float4*>(d)[i] = a;
        }
        for (size_t i = (n4 << 2) + blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) d[i] += src[i];
    } else {
        for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) d[i] += src[i];
    }
}
// the same with the table passed BY VALUE in the kernel arguments (no table upload: the launch is self-contained, so a captured hipGraph
// of a training micro-step replays it as it stands -- the gradient tensors of a capture live at fixed addresses of the graph's pool)
constexpr int MA_ROWS = 120;
struct MaTable { const float* src[MA_ROWS]; long long off[MA_ROWS]; long long n[MA_ROWS]; };
static_assert(sizeof(MaTable) + 16 <= 4096, "kernel arguments are limited to 4 KiB");
__global__ __launch_bounds__(256) void multi_accumulate_args_kernel(float* __restrict__ dst, const MaTable t) {
    const float* __restrict__ src = t.src[blockIdx.y];
    float* __restrict__ d = dst + t.off[blockIdx.y];
    const size_t n = (size_t)t.n[blockIdx.y];
    if (((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(d)) & 15u) == 0) {
        const size_t n4 = n >> 2;
        for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
            float4 a = reinterpret_cast<float4*>(d)[i];
            const float4 b = reinterpret_cast<const float4*>(src)[i];
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
            reinterpret_cast<float4*>(d)[i] = a;
        }
        for (size_t i = (n4 << 2) + blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) d[i] += src[i];
    } else {
        for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) d[i] += src[i];
    }
}
__global__ __launch_bounds__(256) void ema_kernel(float* __restrict__ e, const float* __restrict__ p, size_t n, float w) {
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        e[i] += (p[i] - e[i]) * w;
}

// ---------------------------------------------------------------------------------------------
// softmax over the middle axis of [outer][n][inner]
// ---------------------------------------------------------------------------------------------
// inner == 1, few long rows (GlobalContext: 8 rows of 32768 positions): one 1024-thread workgroup per row
__global__ __launch_bounds__(1024) void softmax_longrow_kernel(const float* __restrict__ x, float* __restrict__ y, int n, float scale) {
    __shared__ float sh[16];
    const float* xr = x + (size_t)blockIdx.x * n;
    float* yr = y + (size_t)blockIdx.x * n;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    float mx = -INFINITY;
    for (int i = tid; i < n; i += 1024) mx = fmaxf(mx, xr[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if (lane == 0) sh[w] = mx;
    __syncthreads();
    mx = sh[0];
#pragma unroll
    for (int k = 1; k < 16; ++k) mx = fmaxf(mx, sh[k]);
    __syncthreads();
    float s = 0.f;
    for (int i = tid; i < n; i += 1024) s += __expf(xr[i] - mx);
    s = wave_sum(s);
    if (lane == 0) sh[w] = s;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) tot += sh[k];
    const float inv = scale / tot;
    for (int i = tid; i < n; i += 1024) yr[i] = __expf(xr[i] - mx) * inv;
}

// inner == 1: one wave per row
__global__ __launch_bounds__(256) void softmax_row_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                          size_t rows, int n, float scale) {
    const int lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    for (size_t r = blockIdx.x * (size_t)wpb + (threadIdx.x >> 6); r < rows; r += (size_t)gridDim.x * wpb) {
        const float* xr = x + r * n;
        float mx = -INFINITY;
        for (int i = lane; i < n; i += 64) mx = fmaxf(mx, xr[i]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        float s = 0.f;
        for (int i = lane; i < n; i += 64) s += __expf(xr[i] - mx);
        s = wave_sum(s);
        const float inv = scale / s;
        for (int i = lane; i < n; i += 64) y[r * n + i] = __expf(xr[i] - mx) * inv;
    }
}
__global__ __launch_bounds__(256) void softmax_row_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy,
                                                              float* __restrict__ dx, size_t rows, int n, float scale) {
    // y = scale*p ; dx = p*(scale*dy - sum(scale*dy*p)) = y*(dy - sum(dy*y)/scale)
    const int lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    for (size_t r = blockIdx.x * (size_t)wpb + (threadIdx.x >> 6); r < rows; r += (size_t)gridDim.x * wpb) {
        float s = 0.f;
        for (int i = lane; i < n; i += 64) s += y[r * n + i] * dy[r * n + i];
        s = wave_sum(s) / scale;
        for (int i = lane; i < n; i += 64) dx[r * n + i] = y[r * n + i] * (dy[r * n + i] - s);
    }
}
// few long rows (the backward of GlobalContext's softmax: 8 rows of 32768): one 1024-thread workgroup per row as in the forward -- a
// wave per row took 250 us per call, four calls per Unet3D training micro-step
__global__ __launch_bounds__(1024) void softmax_longrow_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy,
                                                                   float* __restrict__ dx, int n, float scale) {
    __shared__ float sh[16];
    const float* yr = y + (size_t)blockIdx.x * n;
    const float* dr = dy + (size_t)blockIdx.x * n;
    float* xr = dx + (size_t)blockIdx.x * n;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    float s = 0.f;
    for (int i = tid; i < n; i += 1024) s += yr[i] * dr[i];
    s = wave_sum(s);
    if (lane == 0) sh[w] = s;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) tot += sh[k];
    tot /= scale;
    for (int i = tid; i < n; i += 1024) xr[i] = yr[i] * (dr[i] - tot);
}
// inner > 1 with few columns (LinearAttention's k.softmax(dim=-2): [b*h, n, d], imagen_pytorch3D.py:926-1016): one 1024-thread
// workgroup per outer index, 1024/inner row groups stride over n with coalesced rows; column max / sum combined through LDS in a
// fixed order.  (One thread per column left 512 threads walking 512 strided rows three times: 267 us per call on C4.)
__global__ __launch_bounds__(1024) void softmax_col_wg_kernel(const float* __restrict__ x, float* __restrict__ y, int n, int inner,
                                                              float scale) {
    __shared__ float red[1024];
    __shared__ float colv[1024];
    const int t = threadIdx.x, j = t % inner, rg = t / inner, nrg = 1024 / inner;
    const float* xo = x + (size_t)blockIdx.x * n * inner;
    float* yo = y + (size_t)blockIdx.x * n * inner;
    float mx = -INFINITY;
    for (int i = rg; i < n; i += nrg) mx = fmaxf(mx, xo[(size_t)i * inner + j]);
    red[t] = mx;
    __syncthreads();
    if (t < inner) {
        float m = red[t];
        for (int r = 1; r < nrg; ++r) m = fmaxf(m, red[r * inner + t]);
        colv[t] = m;
    }
    __syncthreads();
    mx = colv[j];
    float sm = 0.f;
    for (int i = rg; i < n; i += nrg) sm += __expf(xo[(size_t)i * inner + j] - mx);
    __syncthreads();
    red[t] = sm;
    __syncthreads();
    if (t < inner) {
        float a = red[t];
        for (int r = 1; r < nrg; ++r) a += red[r * inner + t];
        colv[t] = scale / a;
    }
    __syncthreads();
    const float inv = colv[j];
    for (int i = rg; i < n; i += nrg) yo[(size_t)i * inner + j] = __expf(xo[(size_t)i * inner + j] - mx) * inv;
}
// inner > 1: one thread per (outer, inner) column
__global__ __launch_bounds__(256) void softmax_col_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                          size_t outer, int n, int inner, float scale) {
    const size_t total = outer * inner;
    for (size_t t = blockIdx.x * (size_t)256 + threadIdx.x; t < total; t += (size_t)gridDim.x * 256) {
        const size_t o = t / inner, j = t % inner;
        const float* xc = x + o * n * inner + j;
        float mx = -INFINITY;
        for (int i = 0; i < n; ++i) mx = fmaxf(mx, xc[(size_t)i * inner]);
        float s = 0.f;
        for (int i = 0; i < n; ++i) s += __expf(xc[(size_t)i * inner] - mx);
        const float inv = scale / s;
        float* yc = y + o * n * inner + j;
        for (int i = 0; i < n; ++i) yc[(size_t)i * inner] = __expf(xc[(size_t)i * inner] - mx) * inv;
    }
}
__global__ __launch_bounds__(256) void softmax_col_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy,
                                                              float* __restrict__ dx, size_t outer, int n, int inner,
                                                              float scale) {
    const size_t total = outer * inner;
    for (size_t t = blockIdx.x * (size_t)256 + threadIdx.x; t < total; t += (size_t)gridDim.x * 256) {
        const size_t base = (t / inner) * n * inner + (t % inner);
        float s = 0.f;
        for (int i = 0; i < n; ++i) s += y[base + (size_t)i * inner] * dy[base + (size_t)i * inner];
        s /= scale;
        for (int i = 0; i < n; ++i) {
            const size_t k = base + (size_t)i * inner;
            dx[k] = y[k] * (dy[k] - s);
        }
    }
}


// generic (sd,sh,sw) in {1,2}^3 pixel (un)shuffle; iterates the fine ("space") side, c fastest
__global__ __launch_bounds__(256) void shuffle_nd_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int D,
                                                         int H, int W, int C, int sd, int sh, int sw, int toSpace) {
    const int S = sd * sh * sw;
    const size_t total = (size_t)B * S * D * H * W * C;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        size_t r = i / C;
        const int w2 = (int)(r % (sw * W)); r /= sw * W;
        const int h2 = (int)(r % (sh * H)); r /= sh * H;
        const int d2 = (int)(r % (sd * D));
        const int b = (int)(r / (sd * D));
        const int s = ((d2 % sd) * sh + (h2 % sh)) * sw + (w2 % sw);
        const size_t j = ((((size_t)b * D + d2 / sd) * H + h2 / sh) * W + w2 / sw) * ((size_t)S * C) + (size_t)c * S + s;
        if (toSpace) dst[i] = src[j]; else dst[j] = src[i];
    }
}

// The same for S = sd sh sw in {2, 4, 8}, one thread per (coarse voxel, channel): the S values of a channel are contiguous on the depth
// side (one 8 / 16 / 32-byte access per thread, consecutive lanes consecutive channels) and land in S fine voxels on the space side
// (consecutive lanes consecutive floats).  The element-wise kernel above reads the depth side with a stride of S floats per lane:
// 2.5 TB/s on the 64^3 stage's (1,2,2) shuffles against 4.5 for this form.
template <int S>
__global__ __launch_bounds__(256) void shuffle_nd_vec_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int D, int H,
                                                             int W, int C, int sd, int sh, int sw, int toSpace) {
    const size_t total = (size_t)B * D * H * W * C;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        size_t r = i / C;
        const int w = (int)(r % W); r /= W;
        const int h = (int)(r % H); r /= H;
        const int d = (int)(r % D);
        const int b = (int)(r / D);
        const float* deep = (toSpace ? src : dst) + i * S;
        float v[S];
        if (toSpace) {
#pragma unroll
            for (int q = 0; q < S; q += (S >= 4 ? 4 : 2)) {
                if (S >= 4) { const float4 t = *reinterpret_cast<const float4*>(deep + q); v[q] = t.x; v[q + 1] = t.y; v[q + 2] = t.z; v[q + 3] = t.w; }
                else { const float2 t = *reinterpret_cast<const float2*>(deep + q); v[q] = t.x; v[q + 1] = t.y; }
            }
        }
#pragma unroll
        for (int q = 0; q < S; ++q) {
            const int qw = q % sw, qh = (q / sw) % sh, qd = q / (sw * sh);
            const size_t f = ((((size_t)b * (D * sd) + d * sd + qd) * (H * sh) + h * sh + qh) * (W * sw) + w * sw + qw) * C + c;
            if (toSpace) dst[f] = v[q]; else v[q] = src[f];
        }
        if (!toSpace) {
            float* dp = dst + i * S;
#pragma unroll
            for (int q = 0; q < S; q += (S >= 4 ? 4 : 2)) {
                if (S >= 4) *reinterpret_cast<float4*>(dp + q) = make_float4(v[q], v[q + 1], v[q + 2], v[q + 3]);
                else *reinterpret_cast<float2*>(dp + q) = make_float2(v[q], v[q + 1]);
            }
        }
    }
}

__global__ __launch_bounds__(256) void transpose_mid_kernel(const float* __restrict__ x, float* __restrict__ y, int A, int M,
                                                            int N, int C) {
    const size_t total = (size_t)A * M * N * C;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        size_t r = i / C;
        const int m = (int)(r % M); r /= M;       // output index order: [A][N][M][C]
        const int n = (int)(r % N);
        const int a = (int)(r / N);
        y[i] = x[(((size_t)a * M + m) * N + n) * C + c];
    }
}

__global__ __launch_bounds__(256) void nearest_resize_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int D,
                                                             int H, int W, int C, int Do, int Ho, int Wo) {
    const size_t total = (size_t)B * Do * Ho * Wo * C;
    const float fd = (float)D / Do, fh = (float)H / Ho, fw = (float)W / Wo;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        size_t r = i / C;
        const int ow = (int)(r % Wo); r /= Wo;
        const int oh = (int)(r % Ho); r /= Ho;
        const int od = (int)(r % Do);
        const int b = (int)(r / Do);
        const int id = min((int)floorf(od * fd), D - 1), ih = min((int)floorf(oh * fh), H - 1), iw = min((int)floorf(ow * fw), W - 1);
        y[i] = x[((((size_t)b * D + id) * H + ih) * W + iw) * C + c];
    }
}

// gradient of the nearest-neighbour resize for whole-number up-scaling factors (each source voxel sums its fd x fh x fw copies)
__global__ __launch_bounds__(256) void nearest_resize_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int B, int D,
                                                                 int H, int W, int C, int fd, int fh, int fw) {
    const size_t total = (size_t)B * D * H * W * C;
    const int Ho = H * fh, Wo = W * fw, Do = D * fd;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        size_t r = i / C;
        const int iw = (int)(r % W); r /= W;
        const int ih = (int)(r % H); r /= H;
        const int id = (int)(r % D);
        const int b = (int)(r / D);
        float s = 0.f;
        for (int a = 0; a < fd; ++a)
            for (int e = 0; e < fh; ++e)
                for (int f = 0; f < fw; ++f)
                    s += dy[((((size_t)b * Do + id * fd + a) * Ho + ih * fh + e) * Wo + iw * fw + f) * C + c];
        dx[i] = s;
    }
}

// y = x / max(||x||, eps) over rows of d floats that start `stride` floats apart (F.normalize(dim = -1): l2norm of the cosine-sim
// attention, imagen_video.py:118-119, 484-486): 16 lanes per row
__global__ __launch_bounds__(256) void l2norm_rows_kernel(const float* __restrict__ x, float* __restrict__ y, float* __restrict__ inv,
                                                          size_t rows, int d, int xs, int ys) {
    const int sub = threadIdx.x & 15;
    for (size_t r = (blockIdx.x * (size_t)256 + threadIdx.x) >> 4; r < ((rows + 15) >> 4 << 4); r += ((size_t)gridDim.x * 256) >> 4) {
        const bool ok = r < rows;
        const float* xr = x + (ok ? r : 0) * (size_t)xs;
        float s = 0.f;
        for (int e = sub; e < d; e += 16) { const float v = ok ? xr[e] : 0.f; s += v * v; }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 16);
        const float iv = 1.f / fmaxf(sqrtf(s), 1e-12f);
        if (ok) {
            for (int e = sub; e < d; e += 16) y[r * (size_t)ys + e] = xr[e] * iv;
            if (inv && sub == 0) inv[r] = iv;
        }
    }
}
// dx = (dy - y (y . dy)) * inv
__global__ __launch_bounds__(256) void l2norm_rows_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy,
                                                              const float* __restrict__ inv, float* __restrict__ dx, size_t rows, int d,
                                                              int ys, int dxs) {
    const int sub = threadIdx.x & 15;
    for (size_t r = (blockIdx.x * (size_t)256 + threadIdx.x) >> 4; r < ((rows + 15) >> 4 << 4); r += ((size_t)gridDim.x * 256) >> 4) {
        const bool ok = r < rows;
        const size_t ro = (ok ? r : 0) * (size_t)ys;
        float s = 0.f;
        for (int e = sub; e < d; e += 16) s += ok ? y[ro + e] * dy[ro + e] : 0.f;
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 16);
        if (ok) {
            const float iv = inv[r];
            for (int e = sub; e < d; e += 16) dx[r * (size_t)dxs + e] = (dy[ro + e] - y[ro + e] * s) * iv;
        }
    }
}

// multi-query attention soft-max: one wave per (g, i, head) row of length M = n_extra + n_self
__global__ __launch_bounds__(256) void attn_softmax_fwd_kernel(const float* __restrict__ sim, const float* __restrict__ rel,
                                                               const float* __restrict__ null_bias, float* __restrict__ p,
                                                               size_t rows, int n, int h, int E, int ns, int causal) {
    const int lane = threadIdx.x & 63, wpb = blockDim.x >> 6, M = E + ns;
    for (size_t r = blockIdx.x * (size_t)wpb + (threadIdx.x >> 6); r < rows; r += (size_t)gridDim.x * wpb) {
        const int hh = (int)(r % h), i = (int)((r / h) % n);
        const float* sr = sim + r * M;
        float* pr = p + r * M;
        auto score = [&](int j) -> float {
            float s = sr[j];
            if (j >= E) {
                const int jj = j - E;
                if (causal && jj > i) return -INFINITY;
                if (rel) s += rel[(size_t)(i - jj + ns - 1) * h + hh];
            } else if (j == E - 1 && null_bias) s += null_bias[hh];
            return s;
        };
        float mx = -INFINITY;
        for (int j = lane; j < M; j += 64) mx = fmaxf(mx, score(j));
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        float sum = 0.f;
        for (int j = lane; j < M; j += 64) sum += __expf(score(j) - mx);
        sum = wave_sum(sum);
        const float inv = 1.f / sum;
        for (int j = lane; j < M; j += 64) pr[j] = __expf(score(j) - mx) * inv;
    }
}
__global__ __launch_bounds__(256) void attn_softmax_bwd_kernel(const float* __restrict__ p, const float* __restrict__ dp,
                                                               float* __restrict__ dsim, float* __restrict__ drel,
                                                               float* __restrict__ dnull, size_t rows, int n, int h, int E,
                                                               int ns, int causal) {
    const int lane = threadIdx.x & 63, wpb = blockDim.x >> 6, M = E + ns;
    for (size_t r = blockIdx.x * (size_t)wpb + (threadIdx.x >> 6); r < rows; r += (size_t)gridDim.x * wpb) {
        const int hh = (int)(r % h), i = (int)((r / h) % n);
        float s = 0.f;
        for (int j = lane; j < M; j += 64) s += p[r * M + j] * dp[r * M + j];
        s = wave_sum(s);
        for (int j = lane; j < M; j += 64) {
            const float d = p[r * M + j] * (dp[r * M + j] - s);      // zero where masked (p == 0)
            dsim[r * M + j] = d;
            if (j >= E) {
                const int jj = j - E;
                if (drel && !(causal && jj > i)) atomicAdd(drel + (size_t)(i - jj + ns - 1) * h + hh, d);
            } else if (j == E - 1 && dnull) atomicAdd(dnull + hh, d);
        }
    }
}

__global__ void learned_sinu_fwd_kernel(const float* __restrict__ t, const float* __restrict__ w, float* __restrict__ out,
                                        int B, int half) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * half) return;
    const int b = i / half, j = i % half, D = 2 * half + 1;
    const float f = t[b] * w[j] * 6.283185307179586f;
    if (j == 0) out[b * D] = t[b];
    out[b * D + 1 + j] = sinf(f);
    out[b * D + 1 + half + j] = cosf(f);
}
__global__ void learned_sinu_bwd_kernel(const float* __restrict__ t, const float* __restrict__ w,
                                        const float* __restrict__ dout, float* __restrict__ dw, int B, int half) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= half) return;
    const int D = 2 * half + 1;
    float s = 0.f;
    for (int b = 0; b < B; ++b) {
        const float k = t[b] * 6.283185307179586f, f = k * w[j];
        s += k * (dout[b * D + 1 + j] * cosf(f) - dout[b * D + 1 + half + j] * sinf(f));
    }
    dw[j] = s;
}


// ---------------------------------------------------------------------------------------------
// skinny linear layers (time-conditioning MLPs: M = batch rows <= 64).  The MFMA conv kernel spends ~28 us
// of pipeline fill on these; here one wave owns one output column and streams its weight row once.
// ---------------------------------------------------------------------------------------------
constexpr int LS_MB = 8;      // rows per register block
template <bool VEC>
__global__ __launch_bounds__(256) void linear_small_fwd_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                               const float* __restrict__ bias, float* __restrict__ y,
                                                               int M, int K, int N) {
    const int lane = threadIdx.x & 63, n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const float* w = W + (size_t)n * K;
    for (int m0 = 0; m0 < M; m0 += LS_MB) {
        float acc[LS_MB];
#pragma unroll
        for (int j = 0; j < LS_MB; ++j) acc[j] = 0.f;
        if (VEC) {                  // K % 4 == 0, 16-byte aligned: K = 256 is one 16-byte load per lane and row
            for (int k4 = lane; k4 < (K >> 2); k4 += 64) {
                const float4 wv = reinterpret_cast<const float4*>(w)[k4];
                float4 xv[LS_MB];
#pragma unroll
                for (int j = 0; j < LS_MB; ++j)
                    xv[j] = reinterpret_cast<const float4*>(x + (size_t)min(m0 + j, M - 1) * K)[k4];
#pragma unroll
                for (int j = 0; j < LS_MB; ++j)      // explicit fma chain: hipcc otherwise contracts / packs the rows differently,
                    acc[j] = fmaf(xv[j].w, wv.w, fmaf(xv[j].z, wv.z, fmaf(xv[j].y, wv.y, fmaf(xv[j].x, wv.x, acc[j]))));   // and a row's bits would depend on its position
            }
        } else {
            for (int k = lane; k < K; k += 64) {
                const float wv = w[k];
#pragma unroll
                for (int j = 0; j < LS_MB; ++j) acc[j] = fmaf(x[(size_t)min(m0 + j, M - 1) * K + k], wv, acc[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < LS_MB; ++j) {
            const float s = wave_sum(acc[j]);
            if (lane == 0 && m0 + j < M) y[(size_t)(m0 + j) * N + n] = s + (bias ? bias[n] : 0.f);
        }
    }
}

// dW[n][k] = sum_m dy[m][n] x[m][k];  db[n] = sum_m dy[m][n]
__global__ __launch_bounds__(256) void linear_small_dw_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                              float* __restrict__ dw, float* __restrict__ db,
                                                              int M, int K, int N) {
    const size_t i = blockIdx.x * (size_t)256 + threadIdx.x;
    if (i >= (size_t)N * K) return;
    const int n = (int)(i / K), k = (int)(i % K);
    float s = 0.f, sb = 0.f;
    for (int m = 0; m < M; ++m) {
        const float d = dy[(size_t)m * N + n];
        s += d * x[(size_t)m * K + k];
        sb += d;
    }
    dw[i] = s;
    if (db && k == 0) db[n] = sb;
}

// dX partials: part[slice][m][k] = sum_{n in slice} dy[m][n] W[n][k]   (grid: (ceil(K/256), slices), M <= 64)
__global__ __launch_bounds__(256) void linear_small_dx_part_kernel(const float* __restrict__ W, const float* __restrict__ dy,
                                                                   float* __restrict__ part, int M, int K, int N, int nPer) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    const int n0 = blockIdx.y * nPer, n1 = min(N, n0 + nPer);
    for (int m0 = 0; m0 < M; m0 += LS_MB) {
        float acc[LS_MB];
#pragma unroll
        for (int j = 0; j < LS_MB; ++j) acc[j] = 0.f;
        if (k < K)
            for (int n = n0; n < n1; ++n) {
                const float wv = W[(size_t)n * K + k];
#pragma unroll
                for (int j = 0; j < LS_MB; ++j)
                    if (m0 + j < M) acc[j] = fmaf(dy[(size_t)(m0 + j) * N + n], wv, acc[j]);      // wave-uniform address: scalar load
            }
        if (k < K)
#pragma unroll
            for (int j = 0; j < LS_MB; ++j)
                if (m0 + j < M) part[((size_t)blockIdx.y * M + m0 + j) * K + k] = acc[j];
    }
}
__global__ __launch_bounds__(256) void linear_small_dx_sum_kernel(const float* __restrict__ part, float* __restrict__ dx,
                                                                  int MK, int slices) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= MK) return;
    float s = 0.f;
    for (int q = 0; q < slices; ++q) s += part[(size_t)q * MK + i];
    dx[i] = s;
}

// ---------------------------------------------------------------------------------------------
// dynamic thresholding (imagen_pytorch3D.py:2006-2021): s[b] = torch.quantile(|x0[b]|, p) with linear interpolation
// between the order statistics k and k+1 -- 4-pass MSB radix select on the bit patterns of |x| (monotone for
// non-negative floats), one workgroup per batch row, then x0 = clamp(x0, -s, s) / s.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void abs_quantile_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                            size_t per, unsigned k_lo, int need_hi, float weight) {
    __shared__ unsigned hist[256];
    __shared__ unsigned sh_prefix, sh_k, sh_below, sh_min;
    const float* xr = x + (size_t)blockIdx.x * per;
    const int tid = threadIdx.x;
    unsigned prefix = 0, k = k_lo, below = 0;      // below: elements strictly smaller than the selected prefix range
    for (int pass = 3; pass >= 0; --pass) {
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        const int shift = pass * 8;
        const unsigned himask = pass == 3 ? 0u : (0xFFFFFFFFu << (shift + 8));
        for (size_t i = tid; i < per; i += 1024) {
            const unsigned key = __float_as_uint(xr[i]) & 0x7FFFFFFFu;
            if ((key & himask) == (prefix & himask)) atomicAdd(&hist[(key >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (tid == 0) {
            unsigned cum = 0, bin = 0;
            for (; bin < 256; ++bin) {
                if (cum + hist[bin] > k) break;
                cum += hist[bin];
            }
            sh_prefix = prefix | (bin << shift);
            sh_k = k - cum;
            sh_below = below + cum;
        }
        __syncthreads();
        prefix = sh_prefix; k = sh_k; below = sh_below;
        __syncthreads();
    }
    // prefix = key of the k_lo-th smallest |x|; `below` = #elements < it; k = rank inside the run of equal keys
    float v_lo = __uint_as_float(prefix), v_hi = v_lo;
    if (need_hi) {
        // count of elements equal to the key = hist of the last pass at its bin; k < that count.  If another equal element
        // follows, v_hi == v_lo; otherwise v_hi is the smallest key above.
        const unsigned eq = hist[prefix & 255u];
        if (k + 1 >= eq) {
            if (tid == 0) sh_min = 0x7F800000u;      // +inf
            __syncthreads();
            unsigned m = 0x7F800000u;
            for (size_t i = tid; i < per; i += 1024) {
                const unsigned key = __float_as_uint(xr[i]) & 0x7FFFFFFFu;
                if (key > prefix && key < m) m = key;
            }
            atomicMin(&sh_min, m);
            __syncthreads();
            v_hi = __uint_as_float(sh_min);
        }
    }
    if (tid == 0)     // torch.lerp: a + w (b - a) for w < 0.5, b - (b - a)(1 - w) otherwise
        out[blockIdx.x] = weight < 0.5f ? v_lo + weight * (v_hi - v_lo) : v_hi - (v_hi - v_lo) * (1.f - weight);
}

__global__ __launch_bounds__(256) void dynamic_threshold_kernel(const float* __restrict__ x0, const float* __restrict__ s,
                                                                float* __restrict__ out, size_t per) {
    const int b = blockIdx.y;
    const float sb = s[b];
    const size_t o = (size_t)b * per;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < per; i += (size_t)gridDim.x * 256)
        out[o + i] = fminf(fmaxf(x0[o + i], -sb), sb) / sb;
}

// out = mask != 0 ? y : x     (inpainting: img * ~mask + noised * mask, imagen_pytorch3D.py:2121-2123)
__global__ __launch_bounds__(256) void mask_blend_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                         const float* __restrict__ mask, float* __restrict__ out, size_t n) {
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        out[i] = mask[i] != 0.f ? y[i] : x[i];
}

// ---------------------------------------------------------------------------------------------
// whole-volume inference (test_all.py:182-300, data.py:138-202): sliding-window patches of a resident [D][H][W] volume
// ---------------------------------------------------------------------------------------------
// out[n][P^3] = (vol[idx[n] + (i,j,k)] - mean) / std ; nz[n] = number of non-zero RAW voxels of the patch (5 % rejection rule)
__global__ __launch_bounds__(256) void patch_gather_kernel(const float* __restrict__ vol, const int* __restrict__ idx,
                                                           float* __restrict__ out, int* __restrict__ nz, int D, int H, int W,
                                                           int P, float mean, float stdv) {
    __shared__ int sh[4];
    const int n = blockIdx.y;
    const int i0 = idx[3 * n], j0 = idx[3 * n + 1], k0 = idx[3 * n + 2];
    const size_t per = (size_t)P * P * P;
    int cnt = 0;
    for (size_t e = blockIdx.x * (size_t)256 + threadIdx.x; e < per; e += (size_t)gridDim.x * 256) {
        const int k = (int)(e % P), j = (int)((e / P) % P), i = (int)(e / ((size_t)P * P));
        const float v = vol[((size_t)(i0 + i) * H + (j0 + j)) * W + (k0 + k)];
        cnt += v != 0.f;
        if (out) out[(size_t)n * per + e] = (v - mean) / stdv;
    }
    if (nz) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = cnt;
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(&nz[n], sh[0] + sh[1] + sh[2] + sh[3]);
    }
}
// pred[idx[n] + (i,j,k)] = patch[n][i][j][k] for lo[a] <= coordinate < P - hi[a]   (margins m[n] = {lo0,hi0,lo1,hi1,lo2,hi2})
__global__ __launch_bounds__(256) void patch_scatter_kernel(const float* __restrict__ patches, const int* __restrict__ idx,
                                                            const int* __restrict__ m, float* __restrict__ pred, int D, int H,
                                                            int W, int P) {
    const int n = blockIdx.y;
    const int i0 = idx[3 * n], j0 = idx[3 * n + 1], k0 = idx[3 * n + 2];
    const int* mm = m + 6 * n;
    const size_t per = (size_t)P * P * P;
    for (size_t e = blockIdx.x * (size_t)256 + threadIdx.x; e < per; e += (size_t)gridDim.x * 256) {
        const int k = (int)(e % P), j = (int)((e / P) % P), i = (int)(e / ((size_t)P * P));
        if (i < mm[0] || i >= P - mm[1] || j < mm[2] || j >= P - mm[3] || k < mm[4] || k >= P - mm[5]) continue;
        pred[((size_t)(i0 + i) * H + (j0 + j)) * W + (k0 + k)] = patches[(size_t)n * per + e];
    }
}
// pred[i] = min_val where the normalised low-res voxel equals min_val (test_all.py:300); vol is RAW
__global__ __launch_bounds__(256) void background_reset_kernel(float* __restrict__ pred, const float* __restrict__ vol, size_t n,
                                                               float mean, float stdv, float min_val) {
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        if ((vol[i] - mean) / stdv == min_val) pred[i] = min_val;
}
// two-stage minimum: partial[block] then out[0]
__global__ __launch_bounds__(256) void min_stage1_kernel(const float* __restrict__ x, float* __restrict__ partial, size_t n) {
    __shared__ float sh[4];
    float m = INFINITY;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) m = fminf(m, x[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fminf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = fminf(fminf(sh[0], sh[1]), fminf(sh[2], sh[3]));
}
__global__ __launch_bounds__(64) void min_stage2_kernel(const float* __restrict__ partial, int nb, float* __restrict__ out) {
    float m = INFINITY;
    for (int i = threadIdx.x; i < nb; i += 64) m = fminf(m, partial[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fminf(m, __shfl_xor(m, o, 64));
    if (threadIdx.x == 0) out[0] = m;
}
}  // namespace diqt

using namespace diqt;

// ================================================================================================
// C ABI
// ================================================================================================
#define STREAM ((hipStream_t)stream)

extern "C" size_t diqt_reduce_workspace_bytes(int B, int C) {
    if (B <= 0 || C <= 0) return 0;
    return ((size_t)B * RED_NBLK * 2 * C + (size_t)2 * B * C + (size_t)2 * B * C + 64) * sizeof(float);
}

// pooled[b][c] = sum_n softmax_n(x[b][n][:] . w)[n] x[b][n][c]; C in {64, 128, 256}; workspace: diqt_reduce_workspace_bytes(B, C)
extern "C" int diqt_softmax_pool_supported(int B, int rows, int C) {
    static const bool off = [] { const char* e = getenv("DIQT_NO_GCPOOL"); return e && e[0] == '1'; }();
    return !off && B > 0 && rows > 0 && (C == 64 || C == 128 || C == 256) ? 1 : 0;
}
extern "C" int diqt_softmax_pool(const float* x, const float* w, float* pooled, void* workspace, size_t workspace_bytes, int B, int rows,
                                 int C, void* stream) {
    DIQT_REQUIRE(x && w && pooled && workspace, DIQT_E_ALIGN, "softmax_pool: null pointer");
    DIQT_REQUIRE(diqt_softmax_pool_supported(B, rows, C), DIQT_E_UNSUPPORTED, "softmax_pool: C must be 64, 128 or 256 (got %d)", C);
    DIQT_REQUIRE(aligned16(x) && aligned16(w) && aligned16(workspace), DIQT_E_ALIGN, "softmax_pool: pointers must be 16-byte aligned");
    DIQT_REQUIRE(workspace_bytes >= diqt_reduce_workspace_bytes(B, C), DIQT_E_WORKSPACE, "softmax_pool: workspace too small");
    const int nblk = red_nblk(rows);
    float* partial = static_cast<float*>(workspace);
    hipLaunchKernelGGL(gc_pool_kernel, dim3(nblk, B), dim3(256), 0, STREAM, x, w, partial, rows, C);
    int rc = check_launch("softmax_pool/partial");
    if (rc) return rc;
    hipLaunchKernelGGL(gc_pool_final_kernel, dim3(B), dim3(256), 0, STREAM, partial, pooled, nblk, C);
    return check_launch("softmax_pool/final");
}

static bool vec_ok(const void* a, const void* b, const void* c, size_t per_batch_elems, int C) {
    return (C % 4 == 0) && (per_batch_elems % 4 == 0) && aligned16(a) && (!b || aligned16(b)) && (!c || aligned16(c));
}

extern "C" int diqt_groupnorm_stats(const float* x, float* mean, float* rstd, void* workspace, size_t workspace_bytes,
                                    int B, int rows, int C, int G, float eps, void* stream) {
    DIQT_REQUIRE(x && mean && rstd && workspace, DIQT_E_ALIGN, "groupnorm_stats: null pointer");
    DIQT_REQUIRE(B > 0 && rows > 0 && C > 0 && G > 0 && C % G == 0, DIQT_E_SHAPE, "groupnorm_stats: bad shape (C=%d,G=%d)", C, G);
    DIQT_REQUIRE(workspace_bytes >= diqt_reduce_workspace_bytes(B, C), DIQT_E_WORKSPACE, "groupnorm_stats: workspace too small");
    DIQT_REQUIRE(aligned16(workspace) && (C % 4 != 0 || aligned16(x)), DIQT_E_ALIGN, "groupnorm_stats: misaligned pointer");
    const int nblk = red_nblk(rows);
    float* partial = static_cast<float*>(workspace);
    MomentsF f{x};
    hipLaunchKernelGGL((colreduce_kernel<2, MomentsF>), dim3(nblk, B), dim3(256), 0, STREAM, f, partial, rows, C);
    int rc = check_launch("groupnorm_stats/reduce");
    if (rc) return rc;
    hipLaunchKernelGGL(gn_stats_final_kernel, dim3(B * G), dim3(512), 0, STREAM, partial, mean, rstd, B, C, G,
                       nblk, (double)rows * (C / G), eps, GnCoefOut{nullptr, nullptr, nullptr, nullptr, nullptr, 0});
    return check_launch("groupnorm_stats/final");
}

// diqt_groupnorm_stats that also writes the coefficients of diqt_conv3d_fwd_gn (see diqt_gn_coef_from_partials) from its finalisation
extern "C" int diqt_groupnorm_stats_coef(const float* x, const float* gamma, const float* beta, const float* scale, const float* shift,
                                         int cond_stride, float* mean, float* rstd, float* coef, void* workspace, size_t workspace_bytes,
                                         int B, int rows, int C, int G, float eps, void* stream) {
    DIQT_REQUIRE(x && mean && rstd && coef && workspace, DIQT_E_ALIGN, "groupnorm_stats_coef: null pointer");
    DIQT_REQUIRE(B > 0 && rows > 0 && C > 0 && G > 0 && C % G == 0, DIQT_E_SHAPE, "groupnorm_stats_coef: bad shape (C=%d,G=%d)", C, G);
    DIQT_REQUIRE(workspace_bytes >= diqt_reduce_workspace_bytes(B, C), DIQT_E_WORKSPACE, "groupnorm_stats_coef: workspace too small");
    DIQT_REQUIRE(aligned16(workspace) && (C % 4 != 0 || aligned16(x)), DIQT_E_ALIGN, "groupnorm_stats_coef: misaligned pointer");
    DIQT_REQUIRE((scale == nullptr) == (shift == nullptr) && (!scale || cond_stride >= C), DIQT_E_SHAPE, "groupnorm_stats_coef: scale / shift");
    const int nblk = red_nblk(rows);
    float* partial = static_cast<float*>(workspace);
    MomentsF f{x};
    hipLaunchKernelGGL((colreduce_kernel<2, MomentsF>), dim3(nblk, B), dim3(256), 0, STREAM, f, partial, rows, C);
    int rc = check_launch("groupnorm_stats_coef/reduce");
    if (rc) return rc;
    hipLaunchKernelGGL(gn_stats_final_kernel, dim3(B * G), dim3(512), 0, STREAM, partial, mean, rstd, B, C, G,
                       nblk, (double)rows * (C / G), eps, GnCoefOut{coef, gamma, beta, scale, shift, cond_stride});
    return check_launch("groupnorm_stats_coef/final");
}

// GroupNorm statistics / channel means from per-tile column sums that a producer kernel (the conv epilogue) already wrote:
// partial[b][blk][2 (sum, sum of squares)][C]
extern "C" int diqt_groupnorm_stats_from_partials(const float* partials, float* mean, float* rstd, int B, int nblk, int rows,
                                                  int C, int G, float eps, void* stream) {
    DIQT_REQUIRE(partials && mean && rstd, DIQT_E_ALIGN, "groupnorm_stats_from_partials: null pointer");
    DIQT_REQUIRE(B > 0 && nblk > 0 && rows > 0 && C > 0 && G > 0 && C % G == 0, DIQT_E_SHAPE, "groupnorm_stats_from_partials: bad shape");
    hipLaunchKernelGGL(gn_stats_final_kernel, dim3(B * G), dim3(512), 0, STREAM, partials, mean, rstd, B, C, G, nblk,
                       (double)rows * (C / G), eps, GnCoefOut{nullptr, nullptr, nullptr, nullptr, nullptr, 0});
    return check_launch("groupnorm_stats_from_partials");
}
// GroupNorm statistics from a producer's per-tile column sums AND, in the same launch, the coefficients of the fused
// GroupNorm -> (scale + 1) x + shift -> activation as y = act(A x + Bc): coef[2][B][C] (A, then Bc), what diqt_conv3d_fwd_gn applies to its
// input while staging it (Block.forward, imagen_pytorch3D.py:546-566 / imagen_video.py:680-697; sampling path).
extern "C" int diqt_gn_coef_from_partials(const float* partials, int nblk, int rows, const float* gamma, const float* beta,
                                          const float* scale, const float* shift, int cond_stride, float* mean, float* rstd, float* coef,
                                          int B, int C, int G, float eps, void* stream) {
    DIQT_REQUIRE(partials && mean && rstd && coef, DIQT_E_ALIGN, "gn_coef_from_partials: null pointer");
    DIQT_REQUIRE(B > 0 && nblk > 0 && rows > 0 && C > 0 && G > 0 && C % G == 0, DIQT_E_SHAPE, "gn_coef_from_partials: bad shape");
    DIQT_REQUIRE((scale == nullptr) == (shift == nullptr) && (!scale || cond_stride >= C), DIQT_E_SHAPE, "gn_coef_from_partials: scale / shift");
    hipLaunchKernelGGL(gn_stats_final_kernel, dim3(B * G), dim3(512), 0, STREAM, partials, mean, rstd, B, C, G, nblk,
                       (double)rows * (C / G), eps, GnCoefOut{coef, gamma, beta, scale, shift, cond_stride});
    return check_launch("gn_coef_from_partials");
}
// ... and from statistics that exist already (diqt_groupnorm_stats)
extern "C" int diqt_gn_coef(const float* mean, const float* rstd, const float* gamma, const float* beta, const float* scale,
                            const float* shift, int cond_stride, float* coef, int B, int C, int G, void* stream) {
    DIQT_REQUIRE(mean && rstd && coef, DIQT_E_ALIGN, "gn_coef: null pointer");
    DIQT_REQUIRE(B > 0 && C > 0 && G > 0 && C % G == 0, DIQT_E_SHAPE, "gn_coef: bad shape");
    DIQT_REQUIRE((scale == nullptr) == (shift == nullptr) && (!scale || cond_stride >= C), DIQT_E_SHAPE, "gn_coef: scale / shift");
    GnCoef k{mean, rstd, gamma, beta, scale, shift, C, G, cond_stride};
    hipLaunchKernelGGL(gn_coef_kernel, dim3((B * C + 255) / 256), dim3(256), 0, STREAM, k, coef, B);
    return check_launch("gn_coef");
}
__global__ __launch_bounds__(256) void mean_from_stat_partials_kernel(const float* __restrict__ partial, float* __restrict__ out,
                                                                      int nblk, int C, int total, float alpha) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;      // one wave per (b, c)
    if (i >= total) return;
    const int b = i / C, c = i % C;
    float s = 0.f;
    for (int k = lane; k < nblk; k += 64) s += partial[(((size_t)b * nblk + k) * 2) * C + c];
    s = wave_sum(s);
    if (lane == 0) out[i] = alpha * s;
}
extern "C" int diqt_channel_mean_from_partials(const float* partials, float* pooled, int B, int nblk, int rows, int C, void* stream) {
    DIQT_REQUIRE(partials && pooled, DIQT_E_ALIGN, "channel_mean_from_partials: null pointer");
    DIQT_REQUIRE(B > 0 && nblk > 0 && rows > 0 && C > 0, DIQT_E_SHAPE, "channel_mean_from_partials: bad shape");
    hipLaunchKernelGGL(mean_from_stat_partials_kernel, dim3((B * C + 3) / 4), dim3(256), 0, STREAM, partials, pooled, nblk, C, B * C,
                       1.f / (float)rows);
    return check_launch("channel_mean_from_partials");
}

// grid.x for the vectorised GN apply kernels: a multiple of C / gcd(C, 1024) so every thread's channel quad is loop-invariant
static unsigned gn_grid(size_t per, int C, int B) {
    int a = C, b = 1024;
    while (b) { const int t = a % b; a = b; b = t; }
    const unsigned m = (unsigned)(C / a);
    // one resident round of workgroups (256 CUs x 8 blocks) shared by the B batch slices: long-lived threads amortise
    // the coefficient prologue and keep several 16-byte loads in flight each
    unsigned want = (2048u + (unsigned)B - 1) / (unsigned)B;
    unsigned n = grid_for(per / 4 + 1, 256, want < 1 ? 1 : want);
    return (n + m - 1) / m * m;
}

extern "C" int diqt_gn_act_fwd(const float* x, const float* mean, const float* rstd, const float* gamma,
                               const float* beta, const float* scale, const float* shift, int cond_stride, float* y,
                               int B, int rows, int C, int G, int act, void* stream) {
    DIQT_REQUIRE(x && mean && rstd && y, DIQT_E_ALIGN, "gn_act_fwd: null pointer");
    DIQT_REQUIRE(B > 0 && rows > 0 && C > 0 && G > 0 && C % G == 0, DIQT_E_SHAPE, "gn_act_fwd: bad shape");
    DIQT_REQUIRE((scale == nullptr) == (shift == nullptr), DIQT_E_SHAPE, "gn_act_fwd: scale and shift go together");
    DIQT_REQUIRE(!scale || cond_stride >= C, DIQT_E_SHAPE, "gn_act_fwd: cond_stride < C");
    GnCoef k{mean, rstd, gamma, beta, scale, shift, C, G, cond_stride};
    const size_t per = (size_t)rows * C;
    const dim3 grid(gn_grid(per, C, B), B);
    if (vec_ok(x, y, nullptr, per, C))
        hipLaunchKernelGGL(gn_act_fwd_kernel<true>, grid, dim3(256), 0, STREAM, x, y, k, rows, act);
    else
        hipLaunchKernelGGL(gn_act_fwd_kernel<false>, grid, dim3(256), 0, STREAM, x, y, k, rows, act);
    return check_launch("gn_act_fwd");
}

extern "C" int diqt_gn_act_fwd_h(const void* x_, const float* mean, const float* rstd, const float* gamma, const float* beta,
                                 const float* scale, const float* shift, int cond_stride, void* y_h, int B, int rows, int C, int G,
                                 int act, int bf16, int x_half, void* stream) {
    const float* x = static_cast<const float*>(x_);
    DIQT_REQUIRE(x && mean && rstd && y_h, DIQT_E_ALIGN, "gn_act_fwd_h: null pointer");
    DIQT_REQUIRE(B > 0 && rows > 0 && C > 0 && G > 0 && C % G == 0, DIQT_E_SHAPE, "gn_act_fwd_h: bad shape");
    DIQT_REQUIRE((scale == nullptr) == (shift == nullptr), DIQT_E_SHAPE, "gn_act_fwd_h: scale and shift go together");
    DIQT_REQUIRE(!scale || cond_stride >= C, DIQT_E_SHAPE, "gn_act_fwd_h: cond_stride < C");
    const size_t per = (size_t)rows * C;
    DIQT_REQUIRE(vec_ok(x, static_cast<const float*>(y_h), nullptr, per, C), DIQT_E_UNSUPPORTED,
                 "gn_act_fwd_h: needs C %% 4 == 0 and 16-byte aligned tensors");
    GnCoef k{mean, rstd, gamma, beta, scale, shift, C, G, cond_stride};
    const dim3 grid(gn_grid(per, C, B), B);
    if (x_half) {
        if (bf16) hipLaunchKernelGGL((gn_act_fwd_kernel<true, 2, 2>), grid, dim3(256), 0, STREAM, x, static_cast<float*>(y_h), k, rows, act);
        else hipLaunchKernelGGL((gn_act_fwd_kernel<true, 1, 1>), grid, dim3(256), 0, STREAM, x, static_cast<float*>(y_h), k, rows, act);
    } else {
        if (bf16) hipLaunchKernelGGL((gn_act_fwd_kernel<true, 2>), grid, dim3(256), 0, STREAM, x, static_cast<float*>(y_h), k, rows, act);
        else hipLaunchKernelGGL((gn_act_fwd_kernel<true, 1>), grid, dim3(256), 0, STREAM, x, static_cast<float*>(y_h), k, rows, act);
    }
    return check_launch("gn_act_fwd_h");
}

// ext_partials: the reduction pass already done elsewhere (the epilogue of the conv that produced dy, diqt_conv3d_fwd_gnbwd):
// [B][ext_nblk][2][C] partial sums of (dz, dz xhat)
static int gn_act_bwd_impl(const float* x, const float* dy, const float* mean, const float* rstd,
                           const float* gamma, const float* beta, const float* scale, const float* shift,
                           int cond_stride, float* dx, float* dgamma, float* dbeta, float* dscale, float* dshift, void* workspace,
                           size_t workspace_bytes, int B, int rows, int C, int G, int act, void* stream, const float* ext_partials,
                           int ext_nblk, const float* dx_add = nullptr, int xty = 0, int dxty = 0, int dyty = 0) {
    DIQT_REQUIRE(x && dy && mean && rstd && dx && workspace, DIQT_E_ALIGN, "gn_act_bwd: null pointer");
    DIQT_REQUIRE(xty >= 0 && xty <= 2 && dxty >= 0 && dxty <= 2 && dyty >= 0 && dyty <= 2, DIQT_E_SHAPE,
                 "gn_act_bwd: x / dx / dy type is 0 (fp32), 1 (fp16) or 2 (bf16)");
    DIQT_REQUIRE(B > 0 && rows > 0 && C > 0 && G > 0 && C % G == 0, DIQT_E_SHAPE, "gn_act_bwd: bad shape");
    DIQT_REQUIRE(workspace_bytes >= diqt_reduce_workspace_bytes(B, C), DIQT_E_WORKSPACE, "gn_act_bwd: workspace too small");
    DIQT_REQUIRE(aligned16(workspace), DIQT_E_ALIGN, "gn_act_bwd: misaligned workspace");
    DIQT_REQUIRE(!scale || cond_stride >= C, DIQT_E_SHAPE, "gn_act_bwd: cond_stride < C");
    GnCoef k{mean, rstd, gamma, beta, scale, shift, C, G, cond_stride};
    const int nblk = red_nblk(rows);
    float* partial = static_cast<float*>(workspace);
    float* S = partial + (size_t)B * RED_NBLK * 2 * C;
    float* m12 = S + (size_t)2 * B * C;
    const size_t per = (size_t)rows * C;
    const bool vec = vec_ok(x, dy, dx, per, C) && (!dx_add || aligned16(dx_add));
    DIQT_REQUIRE(vec || (xty == 0 && dxty == 0 && dyty == 0), DIQT_E_UNSUPPORTED,
                 "gn_act_bwd: 16-bit x / dx / dy need C %% 4 == 0 and 16-byte aligned tensors");
    GnBwdF f{x, dy, k, act};
    f.xty = xty;
    f.dyty = dyty;
    int rc = DIQT_OK;
    if (ext_partials) {
        DIQT_REQUIRE(ext_nblk > 0, DIQT_E_SHAPE, "gn_act_bwd_from_partials: nblk");
    } else if (vec || C % 4 != 0 || C > 1024) {
        // the 16-bit combinations a low-precision training step produces get their types at compile time
#define DIQT_GNB_RED(XT, DYT) { GnBwdFT<XT, DYT> ft; static_cast<GnBwdF&>(ft) = f; \
        hipLaunchKernelGGL((colreduce_kernel<2, GnBwdFT<XT, DYT>>), dim3(nblk, B), dim3(256), 0, STREAM, ft, partial, rows, C); }
        if (xty == 0 && dyty == 0) hipLaunchKernelGGL((colreduce_kernel<2, GnBwdF>), dim3(nblk, B), dim3(256), 0, STREAM, f, partial, rows, C);
        else if (xty == 0 && dyty == 1) DIQT_GNB_RED(0, 1)
        else if (xty == 0 && dyty == 2) DIQT_GNB_RED(0, 2)
        else if (xty == 1 && dyty == 1) DIQT_GNB_RED(1, 1)
        else if (xty == 2 && dyty == 2) DIQT_GNB_RED(2, 2)
        else if (xty == 1 && dyty == 0) DIQT_GNB_RED(1, 0)
        else if (xty == 2 && dyty == 0) DIQT_GNB_RED(2, 0)
        else hipLaunchKernelGGL((colreduce_kernel<2, GnBwdF>), dim3(nblk, B), dim3(256), 0, STREAM, f, partial, rows, C);
#undef DIQT_GNB_RED
        rc = check_launch("gn_act_bwd/reduce");
        if (rc) return rc;
    } else {
        set_error("gn_act_bwd: C %% 4 == 0 requires 16-byte aligned x, dy, dx");
        return DIQT_E_ALIGN;
    }
    hipLaunchKernelGGL(sum_partials_kernel, dim3((B * 2 * C + 3) / 4), dim3(256), 0, STREAM, ext_partials ? ext_partials : partial, S,
                       ext_partials ? ext_nblk : nblk, 2 * C, B * 2 * C, 1.f);
    rc = check_launch("gn_act_bwd/sum");
    if (rc) return rc;
    hipLaunchKernelGGL(gn_bwd_final_kernel, dim3((B * C + 255) / 256), dim3(256), 0, STREAM, S, k, dgamma, dbeta, dscale,
                       dshift, m12, B, 1.f / ((float)rows * (C / G)));
    rc = check_launch("gn_act_bwd/final");
    if (rc) return rc;
    const dim3 grid(gn_grid(per, C, B), B);
#define DIQT_GNB_DX(XT, DYT) hipLaunchKernelGGL((gn_act_bwd_dx_kernel<true, XT, DYT>), grid, dim3(256), 0, STREAM, x, dy, dx, k, m12, rows, act, \
                                                dx_add, xty, dxty, dyty)
    if (vec && xty == dxty && xty == 0 && dyty == 1) DIQT_GNB_DX(0, 1);
    else if (vec && xty == dxty && xty == 0 && dyty == 2) DIQT_GNB_DX(0, 2);
    else if (vec && xty == dxty && xty == 1 && dyty == 1) DIQT_GNB_DX(1, 1);
    else if (vec && xty == dxty && xty == 2 && dyty == 2) DIQT_GNB_DX(2, 2);
    else if (vec && xty == dxty && xty == 1 && dyty == 0) DIQT_GNB_DX(1, 0);
    else if (vec && xty == dxty && xty == 2 && dyty == 0) DIQT_GNB_DX(2, 0);
    else if (vec) hipLaunchKernelGGL(gn_act_bwd_dx_kernel<true>, grid, dim3(256), 0, STREAM, x, dy, dx, k, m12, rows, act, dx_add, xty, dxty, dyty);
    else hipLaunchKernelGGL(gn_act_bwd_dx_kernel<false>, grid, dim3(256), 0, STREAM, x, dy, dx, k, m12, rows, act, dx_add);
#undef DIQT_GNB_DX
    return check_launch("gn_act_bwd/dx");
}

extern "C" int diqt_gn_act_bwd(const float* x, const float* dy, const float* mean, const float* rstd,
                               const float* gamma, const float* beta, const float* scale, const float* shift,
                               int cond_stride, float* dx, float* dgamma, float* dbeta, float* dscale, float* dshift, void* workspace,
                               size_t workspace_bytes, int B, int rows, int C, int G, int act, void* stream) {
    return gn_act_bwd_impl(x, dy, mean, rstd, gamma, beta, scale, shift, cond_stride, dx, dgamma, dbeta, dscale, dshift, workspace,
                           workspace_bytes, B, rows, C, G, act, stream, nullptr, 0);
}
extern "C" int diqt_gn_act_bwd_from_partials(const float* x, const float* dy, const float* partials, int nblk, const float* mean,
                                             const float* rstd, const float* gamma, const float* beta, const float* scale,
                                             const float* shift, int cond_stride, float* dx, float* dgamma, float* dbeta, float* dscale,
                                             float* dshift, void* workspace, size_t workspace_bytes, int B, int rows, int C, int G, int act,
                                             void* stream) {
    DIQT_REQUIRE(partials, DIQT_E_ALIGN, "gn_act_bwd_from_partials: null pointer");
    return gn_act_bwd_impl(x, dy, mean, rstd, gamma, beta, scale, shift, cond_stride, dx, dgamma, dbeta, dscale, dshift, workspace,
                           workspace_bytes, B, rows, C, G, act, stream, partials, nblk);
}

// partials (optional): as diqt_gn_act_bwd_from_partials; dx_add (optional, same shape as x): dx = (GroupNorm backward) + dx_add -- the
// gradient reaching x through its other consumer (ResnetBlock: x -> block1 AND x -> res_conv / identity, imagen_pytorch3D.py:601-614,
// imagen_video.py:745-770), so that autograd's separate sum over the two branches disappears.
extern "C" int diqt_gn_act_bwd_ex(const float* x, const float* dy, const float* partials, int nblk, const float* dx_add, const float* mean,
                                  const float* rstd, const float* gamma, const float* beta, const float* scale, const float* shift,
                                  int cond_stride, float* dx, float* dgamma, float* dbeta, float* dscale, float* dshift, void* workspace,
                                  size_t workspace_bytes, int B, int rows, int C, int G, int act, void* stream) {
    return gn_act_bwd_impl(x, dy, mean, rstd, gamma, beta, scale, shift, cond_stride, dx, dgamma, dbeta, dscale, dshift, workspace,
                           workspace_bytes, B, rows, C, G, act, stream, partials, nblk, dx_add);
}

// The same with x read and / or dx written in a 16-bit type (x_type, dx_type: 0 fp32, 1 fp16, 2 bf16): a low-precision training step
// keeps a ResnetBlock's block1 output -- a conv result, rounded to the operand type by autocast anyway -- in that type, and the gradient
// that flows back into it is only ever read by 16-bit-operand conv kernels (backward-data, weight gradient), which would round it too.
extern "C" int diqt_gn_act_bwd_h(const void* x, const void* dy, const float* partials, int nblk, const float* dx_add, const float* mean,
                                 const float* rstd, const float* gamma, const float* beta, const float* scale, const float* shift,
                                 int cond_stride, void* dx, float* dgamma, float* dbeta, float* dscale, float* dshift, void* workspace,
                                 size_t workspace_bytes, int B, int rows, int C, int G, int act, int x_type, int dx_type, int dy_type,
                                 void* stream) {
    return gn_act_bwd_impl(static_cast<const float*>(x), static_cast<const float*>(dy), mean, rstd, gamma, beta, scale, shift, cond_stride,
                           static_cast<float*>(dx), dgamma, dbeta, dscale, dshift, workspace, workspace_bytes, B, rows, C, G, act, stream, partials,
                           nblk, dx_add, x_type, dx_type, dy_type);
}

extern "C" int diqt_chan_layernorm_fwd_res(const float* x, const float* g, const float* b, const float* residual, float* y,
                                           float* mean, float* rstd, int rows, int C, float eps, void* stream) {
    DIQT_REQUIRE(x && g && y, DIQT_E_ALIGN, "chan_layernorm_fwd: null pointer");
    DIQT_REQUIRE(rows > 0 && C > 0, DIQT_E_SHAPE, "chan_layernorm_fwd: bad shape");
    hipLaunchKernelGGL(chan_ln_fwd_kernel, dim3(grid_for((size_t)rows, 4, 4096)), dim3(256), 0, STREAM, x, g, b, residual, y, mean,
                       rstd, rows, C, eps);
    return check_launch("chan_layernorm_fwd");
}
extern "C" int diqt_chan_layernorm_fwd(const float* x, const float* g, const float* b, float* y, float* mean,
                                       float* rstd, int rows, int C, float eps, void* stream) {
    return diqt_chan_layernorm_fwd_res(x, g, b, nullptr, y, mean, rstd, rows, C, eps, stream);
}

extern "C" int diqt_chan_layernorm_bwd(const float* x, const float* dy, const float* g, const float* mean,
                                       const float* rstd, float* dx, float* dg, float* db, void* workspace,
                                       size_t workspace_bytes, int rows, int C, void* stream) {
    return diqt_chan_layernorm_bwd_ex(x, dy, nullptr, g, mean, rstd, dx, dg, db, workspace, workspace_bytes, rows, C, stream);
}
// dx_add (optional, same shape as x): dx = (LayerNorm backward) + dx_add, the gradient of the residual branch of
// `fn(LN(x)) + x` (Attention / feed-forward blocks, imagen_video.py:410-525, 1004-1029)
extern "C" int diqt_chan_layernorm_bwd_ex(const float* x, const float* dy, const float* dx_add, const float* g, const float* mean,
                                          const float* rstd, float* dx, float* dg, float* db, void* workspace,
                                          size_t workspace_bytes, int rows, int C, void* stream) {
    DIQT_REQUIRE(x && dy && g && mean && rstd && dx, DIQT_E_ALIGN, "chan_layernorm_bwd: null pointer");
    DIQT_REQUIRE(rows > 0 && C > 0, DIQT_E_SHAPE, "chan_layernorm_bwd: bad shape");
    hipLaunchKernelGGL(chan_ln_bwd_dx_kernel, dim3(grid_for((size_t)rows, 4, 4096)), dim3(256), 0, STREAM, x, dy, g, mean,
                       rstd, dx, rows, C, dx_add);
    int rc = check_launch("chan_layernorm_bwd/dx");
    if (rc || !dg) return rc;
    DIQT_REQUIRE(workspace && workspace_bytes >= diqt_reduce_workspace_bytes(1, C), DIQT_E_WORKSPACE,
                 "chan_layernorm_bwd: workspace too small");
    const int nblk = red_nblk(rows);
    float* partial = static_cast<float*>(workspace);
    ChanLnDgF f{x, dy, mean, rstd, C};
    hipLaunchKernelGGL((colreduce_kernel<1, ChanLnDgF>), dim3(nblk, 1), dim3(256), 0, STREAM, f, partial, rows, C);
    rc = check_launch("chan_layernorm_bwd/dg");
    if (rc) return rc;
    hipLaunchKernelGGL(sum_partials_kernel, dim3((C + 3) / 4), dim3(256), 0, STREAM, partial, dg, nblk, C, C, 1.f);
    rc = check_launch("chan_layernorm_bwd/dg-final");
    if (rc || !db) return rc;
    hipLaunchKernelGGL((colreduce_kernel<1, IdentF>), dim3(nblk, 1), dim3(256), 0, STREAM, IdentF{dy}, partial, rows, C);
    rc = check_launch("chan_layernorm_bwd/db");
    if (rc) return rc;
    hipLaunchKernelGGL(sum_partials_kernel, dim3((C + 3) / 4), dim3(256), 0, STREAM, partial, db, nblk, C, C, 1.f);
    return check_launch("chan_layernorm_bwd/db-final");
}

// ---------------------------------------------------------------------------------------------
// depthwise temporal conv of the pseudo-3D U-Net's TemporalPEG: nn.Conv3d(C, C, (3,1,1), groups=C) after a causal / symmetric frame
// pad, inside a Residual (/root/reference/imagen_video.py:1340-1362).  An elementwise-class op: x is read once (1.25x with the frame
// halo of a chunk), y written once; channels-last x[B][F][P = H*W][C], w[C][KT], one thread per (b, 8-frame chunk, pixel, channel quad).
// ---------------------------------------------------------------------------------------------
constexpr int DWT_FCH = 8;

template <int KT>
__global__ __launch_bounds__(256) void dwconv_t_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ bias, const float* __restrict__ res,
                                                           float* __restrict__ y, int F, int P4, int C4, int left, int flip) {
    const int e = blockIdx.x * 256 + threadIdx.x;          // float4 index inside a frame: pixel * C4 + channel quad
    if (e >= P4) return;
    const int b = blockIdx.z, f0 = blockIdx.y * DWT_FCH, c = (e % C4) * 4;
    float4 wv[KT];                                         // wv[t] = taps of the 4 channels; flip: the backward-data form
#pragma unroll
    for (int t = 0; t < KT; ++t) {
        const int tt = flip ? KT - 1 - t : t;
        wv[t] = make_float4(w[(c + 0) * KT + tt], w[(c + 1) * KT + tt], w[(c + 2) * KT + tt], w[(c + 3) * KT + tt]);
    }
    const float4 bv = bias ? *reinterpret_cast<const float4*>(bias + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4* xb = reinterpret_cast<const float4*>(x) + (size_t)b * F * P4 + e;
    float4 v[DWT_FCH + KT - 1];                            // frames f0 - left .. f0 + 7 - left + KT - 1, all loads in flight together
#pragma unroll
    for (int u = 0; u < DWT_FCH + KT - 1; ++u) {
        const int f = f0 + u - left;
        v[u] = (f >= 0 && f < F) ? xb[(size_t)f * P4] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float4 rv[DWT_FCH];
    if (res) {
        const float4* rb = reinterpret_cast<const float4*>(res) + (size_t)b * F * P4 + e;
#pragma unroll
        for (int u = 0; u < DWT_FCH; ++u) rv[u] = f0 + u < F ? rb[(size_t)(f0 + u) * P4] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float4* yb = reinterpret_cast<float4*>(y) + (size_t)b * F * P4 + e;
#pragma unroll
    for (int u = 0; u < DWT_FCH; ++u) {
        if (f0 + u >= F) break;
        float4 o = bv;
#pragma unroll
        for (int t = 0; t < KT; ++t) {
            o.x = fmaf(wv[t].x, v[u + t].x, o.x); o.y = fmaf(wv[t].y, v[u + t].y, o.y);
            o.z = fmaf(wv[t].z, v[u + t].z, o.z); o.w = fmaf(wv[t].w, v[u + t].w, o.w);
        }
        if (res) { o.x += rv[u].x; o.y += rv[u].y; o.z += rv[u].z; o.w += rv[u].w; }
        yb[(size_t)(f0 + u) * P4] = o;
    }
}

// partial[blk][KT + 1][C]: rows 0..KT-1 = sum dy[f] x[f + t - left], row KT = sum dy (the bias gradient); blk = (b, frame chunk, pixel chunk)
template <int KT>
__global__ __launch_bounds__(256) void dwconv_t_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                             float* __restrict__ partial, int F, int P, int C, int left, int pch) {
    __shared__ float4 sh[KT + 1][256];
    const int C4 = C >> 2, rpar = 256 / C4;                // C4 divides 256 (host check)
    const int cq = threadIdx.x % C4, pr = threadIdx.x / C4;
    const int b = blockIdx.z, f0 = blockIdx.y * DWT_FCH, p0 = blockIdx.x * pch;
    const int p1 = min(P, p0 + pch);
    float4 acc[KT + 1];
#pragma unroll
    for (int t = 0; t <= KT; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    const size_t P4 = (size_t)P * C4;
    for (int p = p0 + pr; p < p1; p += rpar) {
        const float4* xb = reinterpret_cast<const float4*>(x) + (size_t)b * F * P4 + (size_t)p * C4 + cq;
        const float4* db = reinterpret_cast<const float4*>(dy) + (size_t)b * F * P4 + (size_t)p * C4 + cq;
        float4 v[DWT_FCH + KT - 1], d[DWT_FCH];
#pragma unroll
        for (int u = 0; u < DWT_FCH + KT - 1; ++u) {
            const int f = f0 + u - left;
            v[u] = (f >= 0 && f < F) ? xb[(size_t)f * P4] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < DWT_FCH; ++u) d[u] = f0 + u < F ? db[(size_t)(f0 + u) * P4] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < DWT_FCH; ++u) {
#pragma unroll
            for (int t = 0; t < KT; ++t) {
                acc[t].x = fmaf(d[u].x, v[u + t].x, acc[t].x); acc[t].y = fmaf(d[u].y, v[u + t].y, acc[t].y);
                acc[t].z = fmaf(d[u].z, v[u + t].z, acc[t].z); acc[t].w = fmaf(d[u].w, v[u + t].w, acc[t].w);
            }
            acc[KT].x += d[u].x; acc[KT].y += d[u].y; acc[KT].z += d[u].z; acc[KT].w += d[u].w;
        }
    }
#pragma unroll
    for (int t = 0; t <= KT; ++t) sh[t][threadIdx.x] = acc[t];
    __syncthreads();
    if (threadIdx.x < C4) {                                // fixed order over the parallel pixel rows
        const size_t blk = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
#pragma unroll
        for (int t = 0; t <= KT; ++t) {
            float4 s4 = sh[t][threadIdx.x];
            for (int k = 1; k < rpar; ++k) {
                const float4 o = sh[t][k * C4 + threadIdx.x];
                s4.x += o.x; s4.y += o.y; s4.z += o.z; s4.w += o.w;
            }
            *reinterpret_cast<float4*>(partial + (blk * (KT + 1) + t) * C + threadIdx.x * 4) = s4;
        }
    }
}

static bool dwt_ok(int F, int P, int C, int kt) { return kt == 3 && C % 4 == 0 && 256 % (C / 4) == 0 && F > 0 && P > 0; }
static int dwt_pch(int B, int F, int P) {                  // pixels per workgroup of the weight-gradient pass: ~2048 workgroups
    const int fc = (F + DWT_FCH - 1) / DWT_FCH;
    int pch = (int)(((long long)B * fc * P + 2047) / 2048);
    if (pch < 4) pch = 4;
    return pch;
}

// y = depthwise_conv_t(x) + bias (+ residual); flip = 1 with left = kt - 1 - left_of_forward is the gradient w.r.t. x.
extern "C" int diqt_dwconv_temporal_fwd(const float* x, const float* w, const float* bias, const float* residual, float* y, int B, int F,
                                        int P, int C, int kt, int left, int flip, void* stream) {
    DIQT_REQUIRE(x && w && y, DIQT_E_ALIGN, "dwconv_temporal_fwd: null pointer");
    DIQT_REQUIRE(B > 0 && dwt_ok(F, P, C, kt) && left >= 0 && left < kt, DIQT_E_UNSUPPORTED, "dwconv_temporal_fwd: kt = 3, C %% 4 == 0 only");
    DIQT_REQUIRE(aligned16(x) && aligned16(y) && (!bias || aligned16(bias)) && (!residual || aligned16(residual)), DIQT_E_ALIGN,
                 "dwconv_temporal_fwd: 16-byte alignment");
    const int P4 = P * (C / 4);
    const dim3 grid((P4 + 255) / 256, (F + DWT_FCH - 1) / DWT_FCH, B);
    hipLaunchKernelGGL(dwconv_t_fwd_kernel<3>, grid, dim3(256), 0, STREAM, x, w, bias, residual, y, F, P4, C / 4, left, flip);
    return check_launch("dwconv_temporal_fwd");
}
extern "C" size_t diqt_dwconv_temporal_bwd_weight_workspace_bytes(int B, int F, int P, int C, int kt) {
    if (!dwt_ok(F, P, C, kt)) return 0;
    const int pch = dwt_pch(B, F, P);
    const size_t nblk = (size_t)B * ((F + DWT_FCH - 1) / DWT_FCH) * ((P + pch - 1) / pch);
    return nblk * (kt + 1) * C * sizeof(float);
}
// dwb[(kt + 1)][C]: rows 0..kt-1 = d w[c][t] (tap-major: the caller transposes the kt x C numbers), row kt = d bias
extern "C" int diqt_dwconv_temporal_bwd_weight(const float* x, const float* dy, float* dwb, void* workspace, size_t workspace_bytes,
                                               int B, int F, int P, int C, int kt, int left, void* stream) {
    DIQT_REQUIRE(x && dy && dwb && workspace, DIQT_E_ALIGN, "dwconv_temporal_bwd_weight: null pointer");
    DIQT_REQUIRE(B > 0 && dwt_ok(F, P, C, kt) && left >= 0 && left < kt, DIQT_E_UNSUPPORTED, "dwconv_temporal_bwd_weight: kt = 3, C %% 4 == 0 only");
    DIQT_REQUIRE(workspace_bytes >= diqt_dwconv_temporal_bwd_weight_workspace_bytes(B, F, P, C, kt) && aligned16(workspace) &&
                     aligned16(x) && aligned16(dy), DIQT_E_WORKSPACE, "dwconv_temporal_bwd_weight: workspace / alignment");
    const int pch = dwt_pch(B, F, P);
    const dim3 grid((P + pch - 1) / pch, (F + DWT_FCH - 1) / DWT_FCH, B);
    float* partial = static_cast<float*>(workspace);
    hipLaunchKernelGGL(dwconv_t_wgrad_kernel<3>, grid, dim3(256), 0, STREAM, x, dy, partial, F, P, C, left, pch);
    int rc = check_launch("dwconv_temporal_bwd_weight");
    if (rc) return rc;
    const int nblk = (int)(grid.x * grid.y * grid.z), ncol = (kt + 1) * C;
    hipLaunchKernelGGL(sum_partials_kernel, dim3((ncol + 3) / 4), dim3(256), 0, STREAM, partial, dwb, nblk, ncol, ncol, 1.f);
    return check_launch("dwconv_temporal_bwd_weight/sum");
}

extern "C" int diqt_act_fwd(const float* x, float* y, size_t n, int act, void* stream) {
    DIQT_REQUIRE(x && y, DIQT_E_ALIGN, "act_fwd: null pointer");
    if (n == 0) return DIQT_OK;
    if (aligned16(x) && aligned16(y))
        hipLaunchKernelGGL(act_fwd_kernel<true>, dim3(grid_for(n / 4 + 1, 256)), dim3(256), 0, STREAM, x, y, n, act);
    else
        hipLaunchKernelGGL(act_fwd_kernel<false>, dim3(grid_for(n, 256)), dim3(256), 0, STREAM, x, y, n, act);
    return check_launch("act_fwd");
}
extern "C" int diqt_act_bwd(const float* x, const float* dy, float* dx, size_t n, int act, void* stream) {
    DIQT_REQUIRE(x && dy && dx, DIQT_E_ALIGN, "act_bwd: null pointer");
    if (n == 0) return DIQT_OK;
    if (aligned16(x) && aligned16(dy) && aligned16(dx))
        hipLaunchKernelGGL(act_bwd_kernel<true>, dim3(grid_for(n / 4 + 1, 256)), dim3(256), 0, STREAM, x, dy, dx, n, act);
    else
        hipLaunchKernelGGL(act_bwd_kernel<false>, dim3(grid_for(n, 256)), dim3(256), 0, STREAM, x, dy, dx, n, act);
    return check_launch("act_bwd");
}

template <class F>
static int colreduce1(F f, float* out, float alpha, void* workspace, size_t workspace_bytes, int B, int rows, int C,
                      void* stream, const char* what) {
    DIQT_REQUIRE(workspace && workspace_bytes >= diqt_reduce_workspace_bytes(B, C), DIQT_E_WORKSPACE, "%s: workspace too small", what);
    DIQT_REQUIRE(aligned16(workspace), DIQT_E_ALIGN, "%s: misaligned workspace", what);
    const int nblk = red_nblk(rows);
    float* partial = static_cast<float*>(workspace);
    hipLaunchKernelGGL((colreduce_kernel<1, F>), dim3(nblk, B), dim3(256), 0, STREAM, f, partial, rows, C);
    int rc = check_launch(what);
    if (rc) return rc;
    hipLaunchKernelGGL(sum_partials_kernel, dim3((B * C + 3) / 4), dim3(256), 0, STREAM, partial, out, nblk, C, B * C, alpha);
    return check_launch(what);
}

extern "C" int diqt_channel_mean(const float* x, float* pooled, void* workspace, size_t workspace_bytes, int B,
                                 int rows, int C, void* stream) {
    DIQT_REQUIRE(x && pooled, DIQT_E_ALIGN, "channel_mean: null pointer");
    DIQT_REQUIRE(B > 0 && rows > 0 && C > 0, DIQT_E_SHAPE, "channel_mean: bad shape");
    DIQT_REQUIRE(C % 4 != 0 || aligned16(x), DIQT_E_ALIGN, "channel_mean: misaligned x");
    return colreduce1(IdentF{x}, pooled, 1.f / rows, workspace, workspace_bytes, B, rows, C, stream, "channel_mean");
}

extern "C" int diqt_weighted_colsum(const float* x, const float* w, float* out, void* workspace, size_t workspace_bytes,
                                    int B, int rows, int C, void* stream) {
    DIQT_REQUIRE(x && w && out, DIQT_E_ALIGN, "weighted_colsum: null pointer");
    DIQT_REQUIRE(B > 0 && rows > 0 && C > 0, DIQT_E_SHAPE, "weighted_colsum: bad shape");
    DIQT_REQUIRE(C % 4 != 0 || aligned16(x), DIQT_E_ALIGN, "weighted_colsum: misaligned x");
    return colreduce1(WeightedF{x, w, C}, out, 1.f, workspace, workspace_bytes, B, rows, C, stream, "weighted_colsum");
}

extern "C" int diqt_gate_residual_fwd(const float* h, const float* gate, const float* res, const float* addc,
                                      float alpha, float* y, int B, int rows, int C, void* stream) {
    DIQT_REQUIRE(h && gate && y, DIQT_E_ALIGN, "gate_residual_fwd: null pointer");
    DIQT_REQUIRE(B > 0 && rows > 0 && C > 0, DIQT_E_SHAPE, "gate_residual_fwd: bad shape");
    const size_t per = (size_t)rows * C;
    const dim3 grid(grid_for(per / 4 + 1, 256, 1024), B);
    if (vec_ok(h, res, y, per, C) && aligned16(gate) && (!addc || aligned16(addc)))
        hipLaunchKernelGGL(gate_residual_kernel<true>, grid, dim3(256), 0, STREAM, h, gate, res, addc, alpha, y, rows, C);
    else
        hipLaunchKernelGGL(gate_residual_kernel<false>, grid, dim3(256), 0, STREAM, h, gate, res, addc, alpha, y, rows, C);
    return check_launch("gate_residual_fwd");
}

// The SE gate / its gradient with the conv output h -- a value autocast rounds to the operand type anyway -- and the gradient written back
// into it held in a 16-bit type (h_type / y_type: 0 fp32, 1 fp16, 2 bf16): block2's output of a ResnetBlock under autocast sampling or
// low-precision training lives in that type only (imagen_pytorch3D.py:601-632).  C % 4 == 0, 16-byte aligned tensors.
extern "C" int diqt_gate_residual_fwd_h(const void* h, const float* gate, const float* res, const float* addc, float alpha, void* y, int B,
                                        int rows, int C, int h_type, int y_type, void* stream) {
    DIQT_REQUIRE(h && gate && y, DIQT_E_ALIGN, "gate_residual_fwd_h: null pointer");
    DIQT_REQUIRE(B > 0 && rows > 0 && C > 0 && C % 4 == 0 && h_type >= 0 && h_type <= 2 && y_type >= 0 && y_type <= 2, DIQT_E_SHAPE,
                 "gate_residual_fwd_h: bad shape / type");
    DIQT_REQUIRE(aligned16(h) && aligned16(y) && aligned16(gate) && (!res || aligned16(res)) && (!addc || aligned16(addc)), DIQT_E_ALIGN,
                 "gate_residual_fwd_h: pointers must be 16-byte aligned");
    const size_t per = (size_t)rows * C;
    const dim3 grid(grid_for(per / 4 + 1, 256, 1024), B);
    hipLaunchKernelGGL(gate_residual_kernel<true>, grid, dim3(256), 0, STREAM, static_cast<const float*>(h), gate, res, addc, alpha,
                       static_cast<float*>(y), rows, C, h_type, y_type);
    return check_launch("gate_residual_fwd_h");
}
extern "C" int diqt_gate_residual_fwd_stats_h(const void* h, const float* gate, const float* res, float* y, float* stats, int B, int rows,
                                              int C, int h_type, void* stream) {
    DIQT_REQUIRE(h && gate && y && stats, DIQT_E_ALIGN, "gate_residual_fwd_stats_h: null pointer");
    DIQT_REQUIRE(B > 0 && rows > 0 && C > 0 && h_type >= 0 && h_type <= 2, DIQT_E_SHAPE, "gate_residual_fwd_stats_h: bad shape / type");
    const int nblk = diqt_gate_residual_stats_blocks(rows, C);
    DIQT_REQUIRE(nblk > 0, DIQT_E_UNSUPPORTED, "gate_residual_fwd_stats_h: C = %d must be a multiple of 4 dividing 1024", C);
    DIQT_REQUIRE(aligned16(h) && aligned16(gate) && aligned16(y) && aligned16(stats) && (!res || aligned16(res)), DIQT_E_ALIGN,
                 "gate_residual_fwd_stats_h: pointers must be 16-byte aligned");
    hipLaunchKernelGGL(gate_residual_stats_kernel, dim3(nblk, B), dim3(256), 0, STREAM, static_cast<const float*>(h), gate, res, y, stats, rows,
                       C, h_type);
    return check_launch("gate_residual_fwd_stats_h");
}
extern "C" int diqt_gate_residual_bwd_h(const void* h, const float* dy, float* dgate, void* workspace, size_t workspace_bytes, int B, int rows,
                                        int C, int h_type, void* stream) {
    DIQT_REQUIRE(h && dy && dgate, DIQT_E_ALIGN, "gate_residual_bwd_h: null pointer");
    DIQT_REQUIRE(B > 0 && rows > 0 && C > 0 && C % 4 == 0 && C <= 1024 && h_type >= 0 && h_type <= 2, DIQT_E_SHAPE, "gate_residual_bwd_h: bad shape / type");
    DIQT_REQUIRE(aligned16(h) && aligned16(dy), DIQT_E_ALIGN, "gate_residual_bwd_h: misaligned pointer");
    ProdF f{dy, static_cast<const float*>(h)};
    f.bty = h_type;
    return colreduce1(f, dgate, 1.f, workspace, workspace_bytes, B, rows, C, stream, "gate_residual_bwd_h");
}

// number of per-workgroup partial rows diqt_gate_residual_fwd_stats writes per batch entry (0: shape not supported -- C must divide
// 1024 and be a multiple of 4 -- use diqt_gate_residual_fwd and a statistics pass)
extern "C" int diqt_gate_residual_stats_blocks(int rows, int C) {
    if (rows <= 0 || C < 4 || C % 4 != 0 || 1024 % C != 0) return 0;
    const size_t per = (size_t)rows * C;
    return (int)grid_for(per / 4 + 1, 256, 512);
}

extern "C" int diqt_gate_residual_fwd_stats(const float* h, const float* gate, const float* res, float* y, float* stats, int B, int rows,
                                            int C, void* stream) {
    DIQT_REQUIRE(h && gate && y && stats, DIQT_E_ALIGN, "gate_residual_fwd_stats: null pointer");
    DIQT_REQUIRE(B > 0 && rows > 0 && C > 0, DIQT_E_SHAPE, "gate_residual_fwd_stats: bad shape");
    const int nblk = diqt_gate_residual_stats_blocks(rows, C);
    DIQT_REQUIRE(nblk > 0, DIQT_E_UNSUPPORTED, "gate_residual_fwd_stats: C = %d must be a multiple of 4 dividing 1024", C);
    DIQT_REQUIRE(aligned16(h) && aligned16(gate) && aligned16(y) && aligned16(stats) && (!res || aligned16(res)), DIQT_E_ALIGN,
                 "gate_residual_fwd_stats: pointers must be 16-byte aligned");
    hipLaunchKernelGGL(gate_residual_stats_kernel, dim3(nblk, B), dim3(256), 0, STREAM, h, gate, res, y, stats, rows, C);
    return check_launch("gate_residual_fwd_stats");
}

extern "C" int diqt_gate_residual_bwd(const float* h, const float* dy, float* dgate, void* workspace,
                                      size_t workspace_bytes, int B, int rows, int C, void* stream) {
    DIQT_REQUIRE(h && dy && dgate, DIQT_E_ALIGN, "gate_residual_bwd: null pointer");
    DIQT_REQUIRE(B > 0 && rows > 0 && C > 0, DIQT_E_SHAPE, "gate_residual_bwd: bad shape");
    DIQT_REQUIRE(C % 4 != 0 || (aligned16(h) && aligned16(dy)), DIQT_E_ALIGN, "gate_residual_bwd: misaligned pointer");
    return colreduce1(ProdF{dy, h}, dgate, 1.f, workspace, workspace_bytes, B, rows, C, stream, "gate_residual_bwd");
}

extern "C" int diqt_se_mlp_fwd(const float* pooled, const float* w1, const float* w2, float* hidden, float* gate, int B,
                               int C, int Cr, void* stream) {
    DIQT_REQUIRE(pooled && w1 && w2 && hidden && gate, DIQT_E_ALIGN, "se_mlp_fwd: null pointer");
    DIQT_REQUIRE(B > 0 && C > 0 && Cr > 0, DIQT_E_SHAPE, "se_mlp_fwd: bad shape");
    hipLaunchKernelGGL(se_mlp_fwd_kernel, dim3(1), dim3(256), 0, STREAM, pooled, w1, w2, hidden, gate, B, C, Cr);
    return check_launch("se_mlp_fwd");
}
// squeeze + excitation in one launch: `partials` (per-tile column sums of h from its producer, [B][nblk][2][C]) or, when NULL,
// `pooled` as input; writes pooled (when computed here), hidden and gate
extern "C" int diqt_se_pool_mlp_fwd(const float* partials, int nblk, int rows, float* pooled, const float* w1, const float* w2,
                                    float* hidden, float* gate, int B, int C, int Cr, void* stream) {
    DIQT_REQUIRE(pooled && w1 && w2 && hidden && gate, DIQT_E_ALIGN, "se_pool_mlp_fwd: null pointer");
    DIQT_REQUIRE(B > 0 && C > 0 && Cr > 0 && (!partials || (nblk > 0 && rows > 0)), DIQT_E_SHAPE, "se_pool_mlp_fwd: bad shape");
    DIQT_REQUIRE(!partials || 256 % (C < 256 ? C : 256) == 0 && (C <= 256 || C % 256 == 0), DIQT_E_UNSUPPORTED,
                 "se_pool_mlp_fwd: C = %d must divide 256 or be a multiple of it to finish the channel means here", C);
    const size_t lds = (size_t)(256 + C + Cr) * sizeof(float);
    DIQT_REQUIRE(lds <= 64 * 1024, DIQT_E_UNSUPPORTED, "se_pool_mlp_fwd: C too large");
    hipLaunchKernelGGL(se_pool_mlp_fwd_kernel, dim3(B), dim3(256), lds, STREAM, partials, nblk, partials ? 1.f / (float)rows : 1.f,
                       partials ? nullptr : pooled, w1, w2, pooled, hidden, gate, C, Cr);
    return check_launch("se_pool_mlp_fwd");
}
extern "C" int diqt_se_mlp_bwd(const float* pooled, const float* w1, const float* w2, const float* hidden,
                               const float* gate, const float* dgate, float* dpooled, float* dw1, float* dw2,
                               float* scratch, int B, int C, int Cr, void* stream) {
    DIQT_REQUIRE(pooled && w1 && w2 && hidden && gate && dgate && dpooled && dw1 && dw2 && scratch, DIQT_E_ALIGN,
                 "se_mlp_bwd: null pointer");
    DIQT_REQUIRE(B > 0 && C > 0 && Cr > 0, DIQT_E_SHAPE, "se_mlp_bwd: bad shape");
    const size_t lds = ((size_t)2 * B * C + (size_t)2 * B * Cr + (size_t)2 * C * Cr) * sizeof(float);
    if (lds <= 64 * 1024)
        hipLaunchKernelGGL(se_mlp_bwd_kernel<true>, dim3(1), dim3(256), lds, STREAM, pooled, w1, w2, hidden, gate, dgate, dpooled, dw1,
                           dw2, scratch, B, C, Cr);
    else
        hipLaunchKernelGGL(se_mlp_bwd_kernel<false>, dim3(1), dim3(256), 0, STREAM, pooled, w1, w2, hidden, gate, dgate, dpooled, dw1,
                           dw2, scratch, B, C, Cr);
    return check_launch("se_mlp_bwd");
}

extern "C" int diqt_add_channel_broadcast(float* x, const float* v, float alpha, int B, int rows, int C, void* stream) {
    DIQT_REQUIRE(x && v, DIQT_E_ALIGN, "add_channel_broadcast: null pointer");
    DIQT_REQUIRE(B > 0 && rows > 0 && C > 0, DIQT_E_SHAPE, "add_channel_broadcast: bad shape");
    hipLaunchKernelGGL(add_channel_broadcast_kernel, dim3(grid_for((size_t)rows * C, 256, 1024), B), dim3(256), 0, STREAM, x,
                       v, alpha, rows, C);
    return check_launch("add_channel_broadcast");
}

extern "C" int diqt_space_to_depth2(const float* x, float* y, int B, int D, int H, int W, int C, void* stream) {
    DIQT_REQUIRE(x && y, DIQT_E_ALIGN, "space_to_depth2: null pointer");
    DIQT_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0 && C > 0, DIQT_E_SHAPE, "space_to_depth2: bad shape");
    if (aligned16(y)) {      // one thread per (coarse voxel, channel): its 8 values are contiguous on the depth side
        hipLaunchKernelGGL(shuffle_nd_vec_kernel<8>, dim3(grid_for((size_t)B * D * H * W * C, 256)), dim3(256), 0, STREAM, x, y, B, D, H, W, C,
                           2, 2, 2, 0);
        return check_launch("space_to_depth2");
    }
    hipLaunchKernelGGL(shuffle2_kernel, dim3(grid_for((size_t)B * 8 * D * H * W * C, 256)), dim3(256), 0, STREAM, x, y, B, D,
                       H, W, C, 0);
    return check_launch("space_to_depth2");
}
extern "C" int diqt_depth_to_space2(const float* x, float* y, int B, int D, int H, int W, int C, void* stream) {
    DIQT_REQUIRE(x && y, DIQT_E_ALIGN, "depth_to_space2: null pointer");
    DIQT_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0 && C > 0, DIQT_E_SHAPE, "depth_to_space2: bad shape");
    if (aligned16(x)) {
        hipLaunchKernelGGL(shuffle_nd_vec_kernel<8>, dim3(grid_for((size_t)B * D * H * W * C, 256)), dim3(256), 0, STREAM, x, y, B, D, H, W, C,
                           2, 2, 2, 1);
        return check_launch("depth_to_space2");
    }
    hipLaunchKernelGGL(shuffle2_kernel, dim3(grid_for((size_t)B * 8 * D * H * W * C, 256)), dim3(256), 0, STREAM, x, y, B, D,
                       H, W, C, 1);
    return check_launch("depth_to_space2");
}
extern "C" int diqt_concat_channels(const float* a, int Ca, const float* b, int Cb, float* y, size_t rows, void* stream) {
    DIQT_REQUIRE(a && b && y, DIQT_E_ALIGN, "concat_channels: null pointer");
    DIQT_REQUIRE(Ca > 0 && Cb > 0, DIQT_E_SHAPE, "concat_channels: bad shape");
    if (rows == 0) return DIQT_OK;
    hipLaunchKernelGGL(concat_kernel, dim3(grid_for(rows * (Ca + Cb), 256)), dim3(256), 0, STREAM, a, Ca, b, Cb, y, rows);
    return check_launch("concat_channels");
}
extern "C" int diqt_concat_channels_scaled(const float* a, int Ca, const float* b, int Cb, float sa, float sb, float* y, size_t rows,
                                           void* stream) {
    DIQT_REQUIRE(a && b && y, DIQT_E_ALIGN, "concat_channels_scaled: null pointer");
    DIQT_REQUIRE(Ca > 0 && Cb > 0, DIQT_E_SHAPE, "concat_channels_scaled: bad shape");
    if (rows == 0) return DIQT_OK;
    if (Ca % 4 == 0 && Cb % 4 == 0 && aligned16(a) && aligned16(b) && aligned16(y)) {
        hipLaunchKernelGGL(concat4_kernel, dim3(grid_for(rows * ((Ca + Cb) / 4), 256)), dim3(256), 0, STREAM,
                           reinterpret_cast<const float4*>(a), Ca / 4, reinterpret_cast<const float4*>(b), Cb / 4, sa, sb,
                           reinterpret_cast<float4*>(y), rows);
    } else {
        hipLaunchKernelGGL(concat_scaled_kernel, dim3(grid_for(rows * (Ca + Cb), 256)), dim3(256), 0, STREAM, a, Ca, b, Cb, sa, sb, y, rows);
    }
    return check_launch("concat_channels_scaled");
}
// rows of column sums per batch entry diqt_concat_channels_stats writes (0: shape not taken -- channel counts that are not multiples of 4
// or more than 1024 channels)
extern "C" int diqt_concat_channels_stats_blocks(int Ca, int Cb, int rows_per_batch) {
    if (Ca <= 0 || Cb <= 0 || rows_per_batch <= 0 || Ca % 4 || Cb % 4 || (Ca + Cb) / 4 > 256) return 0;
    return (rows_per_batch + 255) / 256;
}
extern "C" int diqt_concat_channels_stats(const float* a, int Ca, const float* b, int Cb, float sa, float sb, float* y, int B,
                                          int rows_per_batch, float* stats, void* stream) {
    DIQT_REQUIRE(a && b && y && stats, DIQT_E_ALIGN, "concat_channels_stats: null pointer");
    const int nblk = diqt_concat_channels_stats_blocks(Ca, Cb, rows_per_batch);
    DIQT_REQUIRE(nblk > 0 && B > 0, DIQT_E_UNSUPPORTED, "concat_channels_stats: shape not taken (diqt_concat_channels_stats_blocks == 0)");
    DIQT_REQUIRE(aligned16(a) && aligned16(b) && aligned16(y) && aligned16(stats), DIQT_E_ALIGN, "concat_channels_stats: pointers must be 16-byte aligned");
    const int C4 = (Ca + Cb) / 4, RL = 256 / C4;
    hipLaunchKernelGGL(concat4_stats_kernel, dim3(nblk, B), dim3(256), (size_t)2 * RL * C4 * sizeof(float4), STREAM,
                       reinterpret_cast<const float4*>(a), Ca / 4, reinterpret_cast<const float4*>(b), Cb / 4, sa, sb,
                       reinterpret_cast<float4*>(y), rows_per_batch, 256, stats, nblk);
    return check_launch("concat_channels_stats");
}
extern "C" int diqt_split_channels_scaled(const float* y, float* a, int Ca, float* b, int Cb, float sa, float sb, size_t rows,
                                          void* stream) {
    DIQT_REQUIRE(y && (a || b), DIQT_E_ALIGN, "split_channels_scaled: null pointer");
    DIQT_REQUIRE(Ca > 0 && Cb > 0, DIQT_E_SHAPE, "split_channels_scaled: bad shape");
    if (rows == 0) return DIQT_OK;
    if (Ca % 4 == 0 && Cb % 4 == 0 && aligned16(y) && (!a || aligned16(a)) && (!b || aligned16(b))) {
        hipLaunchKernelGGL(split4_kernel, dim3(grid_for(rows * ((Ca + Cb) / 4), 256)), dim3(256), 0, STREAM,
                           reinterpret_cast<const float4*>(y), reinterpret_cast<float4*>(a), Ca / 4, reinterpret_cast<float4*>(b), Cb / 4,
                           sa, sb, rows);
    } else {
        hipLaunchKernelGGL(split_scaled_kernel, dim3(grid_for(rows * (Ca + Cb), 256)), dim3(256), 0, STREAM, y, a, Ca, b, Cb, sa, sb, rows);
    }
    return check_launch("split_channels_scaled");
}
extern "C" int diqt_split_channels(const float* y, float* a, int Ca, float* b, int Cb, size_t rows, void* stream) {
    DIQT_REQUIRE(y && (a || b), DIQT_E_ALIGN, "split_channels: null pointer");
    DIQT_REQUIRE(Ca > 0 && Cb > 0, DIQT_E_SHAPE, "split_channels: bad shape");
    if (rows == 0) return DIQT_OK;
    hipLaunchKernelGGL(split_kernel, dim3(grid_for(rows * (Ca + Cb), 256)), dim3(256), 0, STREAM, y, a, Ca, b, Cb, rows);
    return check_launch("split_channels");
}
extern "C" int diqt_subvolume_gather(const float* vol, float* sub, int f, int A, int C, int halo, void* stream) {
    DIQT_REQUIRE(vol && sub, DIQT_E_ALIGN, "subvolume_gather: null pointer");
    DIQT_REQUIRE(f > 0 && A > 0 && C > 0 && halo >= 0, DIQT_E_SHAPE, "subvolume_gather: bad shape");
    const size_t Ap = A + 2 * halo;
    hipLaunchKernelGGL(subvolume_kernel, dim3(grid_for((size_t)f * f * f * Ap * Ap * Ap * C, 256)), dim3(256), 0, STREAM, vol,
                       sub, f, A, C, halo, 0, 0);
    return check_launch("subvolume_gather");
}
extern "C" int diqt_subvolume_scatter(const float* sub, float* vol, int f, int A, int C, int halo, int accumulate,
                                      void* stream) {
    DIQT_REQUIRE(vol && sub, DIQT_E_ALIGN, "subvolume_scatter: null pointer");
    DIQT_REQUIRE(f > 0 && A > 0 && C > 0 && halo >= 0, DIQT_E_SHAPE, "subvolume_scatter: bad shape");
    DIQT_REQUIRE(halo == 0 || accumulate, DIQT_E_UNSUPPORTED, "subvolume_scatter: overlapping halo blocks need accumulate=1");
    const size_t Ap = A + 2 * halo;
    hipLaunchKernelGGL(subvolume_kernel, dim3(grid_for((size_t)f * f * f * Ap * Ap * Ap * C, 256)), dim3(256), 0, STREAM, sub,
                       vol, f, A, C, halo, 1, accumulate);
    return check_launch("subvolume_scatter");
}

extern "C" int diqt_axpby3(const float* a, const float* b_, const float* c_, const float* c0, const float* c1,
                           const float* c2, float lo, float hi, int clamp_mode, float* out, int B, size_t per,
                           void* stream) {
    DIQT_REQUIRE(a && c0 && out, DIQT_E_ALIGN, "axpby3: null pointer");
    DIQT_REQUIRE(B > 0 && per > 0, DIQT_E_SHAPE, "axpby3: bad shape");
    hipLaunchKernelGGL(axpby3_kernel, dim3(grid_for(per, 256, 1024), B), dim3(256), 0, STREAM, a, b_, c_, c0, c1, c2, lo, hi,
                       clamp_mode, out, per);
    return check_launch("axpby3");
}
extern "C" int diqt_q_sample(const float* x0, const float* noise, const float* alpha, const float* sigma, float* xt,
                             int B, size_t per, void* stream) {
    DIQT_REQUIRE(x0 && noise && alpha && sigma && xt, DIQT_E_ALIGN, "q_sample: null pointer");
    return diqt_axpby3(x0, noise, nullptr, alpha, sigma, nullptr, 0.f, 0.f, 0, xt, B, per, stream);
}
extern "C" int diqt_ddpm_step(const float* x_t, const float* pred, const float* noise, const float* ca, const float* cb,
                              const float* cn, float lo, float hi, int clamp_mode, float* x_next, float* x0_out, int B,
                              size_t per, void* stream) {
    DIQT_REQUIRE(x_t && pred && noise && ca && cb && cn && x_next, DIQT_E_ALIGN, "ddpm_step: null pointer");
    DIQT_REQUIRE(B > 0 && per > 0, DIQT_E_SHAPE, "ddpm_step: bad shape");
    hipLaunchKernelGGL(ddpm_step_kernel, dim3(grid_for(per, 256, 1024), B), dim3(256), 0, STREAM, x_t, pred, noise, ca, cb, cn,
                       lo, hi, clamp_mode, x_next, x0_out, per);
    return check_launch("ddpm_step");
}
extern "C" int diqt_loss_clamp_fwd(const float* pred, float* pred_clamped, const float* target, const float* w, float lo,
                                   int do_clamp, int kind, float* partials, float* loss_out, int B, size_t per, void* stream) {
    DIQT_REQUIRE(pred && target && partials && loss_out, DIQT_E_ALIGN, "loss_clamp_fwd: null pointer");
    DIQT_REQUIRE(B > 0 && per > 0 && kind >= 0 && kind <= 2, DIQT_E_SHAPE, "loss_clamp_fwd: bad shape / kind");
    const unsigned nblk = grid_for((size_t)B * per, 256, 1024);
    hipLaunchKernelGGL(mse_clamp_fwd_kernel, dim3(nblk), dim3(256), 0, STREAM, pred, pred_clamped, target, w, lo, do_clamp, partials, B, per, kind);
    int rc = check_launch("loss_clamp_fwd");
    if (rc) return rc;
    hipLaunchKernelGGL(mse_final_kernel, dim3(1), dim3(256), 0, STREAM, partials, (int)nblk, 1.f / ((float)B * (float)per),
                       loss_out);
    return check_launch("loss_final");
}
extern "C" int diqt_loss_clamp_bwd(const float* pred, const float* target, const float* w, float lo, int do_clamp, int kind,
                                   float gscale, float* dpred, int B, size_t per, void* stream) {
    DIQT_REQUIRE(pred && target && dpred, DIQT_E_ALIGN, "loss_clamp_bwd: null pointer");
    DIQT_REQUIRE(B > 0 && per > 0 && kind >= 0 && kind <= 2, DIQT_E_SHAPE, "loss_clamp_bwd: bad shape / kind");
    const float coef = gscale * 2.f / ((float)B * (float)per);
    hipLaunchKernelGGL(mse_clamp_bwd_kernel, dim3(grid_for((size_t)B * per, 256, 2048)), dim3(256), 0, STREAM, pred, target, w,
                       lo, do_clamp, coef, dpred, B, per, kind);
    return check_launch("loss_clamp_bwd");
}
extern "C" int diqt_mse_clamp_fwd(const float* pred, float* pred_clamped, const float* target, const float* w, float lo,
                                  int do_clamp, float* partials, float* loss_out, int B, size_t per, void* stream) {
    return diqt_loss_clamp_fwd(pred, pred_clamped, target, w, lo, do_clamp, 0, partials, loss_out, B, per, stream);
}
extern "C" int diqt_mse_clamp_bwd(const float* pred, const float* target, const float* w, float lo, int do_clamp,
                                  float gscale, float* dpred, int B, size_t per, void* stream) {
    return diqt_loss_clamp_bwd(pred, target, w, lo, do_clamp, 0, gscale, dpred, B, per, stream);
}

extern "C" int diqt_adam_step(float* param, float* grad, float* exp_avg, float* exp_avg_sq, size_t n, float lr,
                              float beta1, float beta2, float eps, float weight_decay, float bias_correction1,
                              float bias_correction2, int zero_grad, void* stream) {
    DIQT_REQUIRE(param && grad && exp_avg && exp_avg_sq, DIQT_E_ALIGN, "adam_step: null pointer");
    if (n == 0) return DIQT_OK;
    DIQT_REQUIRE(bias_correction1 > 0.f && bias_correction2 > 0.f, DIQT_E_SHAPE, "adam_step: bad bias correction");
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n, 256, 4096)), dim3(256), 0, STREAM, param, grad, exp_avg, exp_avg_sq, n,
                       lr, beta1, beta2, eps, weight_decay, bias_correction1, sqrtf(bias_correction2), zero_grad, (const float*)nullptr);
    return check_launch("adam_step");
}
extern "C" int diqt_adam_step_scaled(float* param, float* grad, float* exp_avg, float* exp_avg_sq, size_t n, float lr, float beta1,
                                     float beta2, float eps, float weight_decay, float bias_correction1, float bias_correction2,
                                     int zero_grad, const float* grad_scale, void* stream) {
    DIQT_REQUIRE(param && grad && exp_avg && exp_avg_sq && grad_scale, DIQT_E_ALIGN, "adam_step_scaled: null pointer");
    if (n == 0) return DIQT_OK;
    DIQT_REQUIRE(bias_correction1 > 0.f && bias_correction2 > 0.f, DIQT_E_SHAPE, "adam_step_scaled: bad bias correction");
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n, 256, 4096)), dim3(256), 0, STREAM, param, grad, exp_avg, exp_avg_sq, n,
                       lr, beta1, beta2, eps, weight_decay, bias_correction1, sqrtf(bias_correction2), zero_grad, grad_scale);
    return check_launch("adam_step(scaled)");
}
extern "C" size_t diqt_grad_norm_workspace_bytes(void) { return (size_t)kNormBlocks * sizeof(double); }
extern "C" int diqt_grad_norm_clip(const float* grad, size_t n, float max_norm, void* workspace, float* out2, void* stream) {
    DIQT_REQUIRE(grad && workspace && out2, DIQT_E_ALIGN, "grad_norm_clip: null pointer");
    DIQT_REQUIRE(max_norm > 0.f, DIQT_E_SHAPE, "grad_norm_clip: max_norm must be positive");
    const int nb = (int)grid_for(n ? n : 1, 256, kNormBlocks);
    hipLaunchKernelGGL(gradnorm_stage1_kernel, dim3(nb), dim3(256), 0, STREAM, grad, n, static_cast<double*>(workspace));
    hipLaunchKernelGGL(gradnorm_stage2_kernel, dim3(1), dim3(256), 0, STREAM, static_cast<const double*>(workspace), nb, max_norm, out2);
    return check_launch("grad_norm_clip");
}
extern "C" int diqt_multi_accumulate(float* dst, const long long* table, int count, int blocks_per_tensor, void* stream) {
    DIQT_REQUIRE(dst && table, DIQT_E_ALIGN, "multi_accumulate: null pointer");
    if (count <= 0) return DIQT_OK;
    DIQT_REQUIRE(count <= 65535 && blocks_per_tensor > 0, DIQT_E_SHAPE, "multi_accumulate: count %d out of range", count);
    hipLaunchKernelGGL(multi_accumulate_kernel, dim3(blocks_per_tensor, count), dim3(256), 0, STREAM, dst, table);
    return check_launch("multi_accumulate");
}
extern "C" int diqt_multi_accumulate_host(float* dst, const long long* host_table, int count, int blocks_per_tensor, void* stream) {
    DIQT_REQUIRE(dst && host_table, DIQT_E_ALIGN, "multi_accumulate_host: null pointer");
    DIQT_REQUIRE(count >= 0 && blocks_per_tensor > 0, DIQT_E_SHAPE, "multi_accumulate_host: count %d out of range", count);
    for (int lo = 0; lo < count; lo += MA_ROWS) {
        const int m = count - lo < MA_ROWS ? count - lo : MA_ROWS;
        MaTable t;
        for (int i = 0; i < MA_ROWS; ++i) {
            const long long* e = host_table + 3 * (size_t)(lo + (i < m ? i : 0));
            t.src[i] = reinterpret_cast<const float*>(e[0]);
            t.off[i] = e[1];
            t.n[i] = e[2];
            DIQT_REQUIRE(t.src[i] && t.off[i] >= 0 && t.n[i] >= 0, DIQT_E_SHAPE, "multi_accumulate_host: bad row %d", lo + i);
        }
        hipLaunchKernelGGL(multi_accumulate_args_kernel, dim3(blocks_per_tensor, m), dim3(256), 0, STREAM, dst, t);
        const int rc = check_launch("multi_accumulate");
        if (rc != DIQT_OK) return rc;
    }
    return DIQT_OK;
}
extern "C" int diqt_ema_lerp(float* ema, const float* param, size_t n, float one_minus_decay, void* stream) {
    DIQT_REQUIRE(ema && param, DIQT_E_ALIGN, "ema_lerp: null pointer");
    if (n == 0) return DIQT_OK;
    hipLaunchKernelGGL(ema_kernel, dim3(grid_for(n, 256, 4096)), dim3(256), 0, STREAM, ema, param, n, one_minus_decay);
    return check_launch("ema_lerp");
}

extern "C" int diqt_softmax_fwd(const float* x, float* y, size_t outer, int n, int inner, float scale, void* stream) {
    DIQT_REQUIRE(x && y, DIQT_E_ALIGN, "softmax_fwd: null pointer");
    DIQT_REQUIRE(n > 0 && inner > 0, DIQT_E_SHAPE, "softmax_fwd: bad shape");
    if (outer == 0) return DIQT_OK;
    if (inner == 1 && outer <= 1024 && n >= 4096)
        hipLaunchKernelGGL(softmax_longrow_kernel, dim3((unsigned)outer), dim3(1024), 0, STREAM, x, y, n, scale);
    else if (inner == 1)
        hipLaunchKernelGGL(softmax_row_kernel, dim3(grid_for(outer, 4, 8192)), dim3(256), 0, STREAM, x, y, outer, n, scale);
    else if (inner <= 1024 && 1024 % inner == 0 && n >= 64 && outer <= 65535)
        hipLaunchKernelGGL(softmax_col_wg_kernel, dim3((unsigned)outer), dim3(1024), 0, STREAM, x, y, n, inner, scale);
    else
        hipLaunchKernelGGL(softmax_col_kernel, dim3(grid_for(outer * inner, 256)), dim3(256), 0, STREAM, x, y, outer, n, inner, scale);
    return check_launch("softmax_fwd");
}
extern "C" int diqt_softmax_bwd(const float* y, const float* dy, float* dx, size_t outer, int n, int inner, float scale,
                                void* stream) {
    DIQT_REQUIRE(y && dy && dx, DIQT_E_ALIGN, "softmax_bwd: null pointer");
    DIQT_REQUIRE(n > 0 && inner > 0 && scale != 0.f, DIQT_E_SHAPE, "softmax_bwd: bad shape");
    if (outer == 0) return DIQT_OK;
    if (inner == 1 && outer <= 1024 && n >= 4096)
        hipLaunchKernelGGL(softmax_longrow_bwd_kernel, dim3((unsigned)outer), dim3(1024), 0, STREAM, y, dy, dx, n, scale);
    else if (inner == 1)
        hipLaunchKernelGGL(softmax_row_bwd_kernel, dim3(grid_for(outer, 4, 8192)), dim3(256), 0, STREAM, y, dy, dx, outer, n, scale);
    else
        hipLaunchKernelGGL(softmax_col_bwd_kernel, dim3(grid_for(outer * inner, 256)), dim3(256), 0, STREAM, y, dy, dx, outer, n, inner, scale);
    return check_launch("softmax_bwd");
}

extern "C" int diqt_learned_sinu_fwd(const float* t, const float* w, float* out, int B, int half, void* stream) {
    DIQT_REQUIRE(t && w && out, DIQT_E_ALIGN, "learned_sinu_fwd: null pointer");
    DIQT_REQUIRE(B > 0 && half > 0, DIQT_E_SHAPE, "learned_sinu_fwd: bad shape");
    hipLaunchKernelGGL(learned_sinu_fwd_kernel, dim3((B * half + 63) / 64), dim3(64), 0, STREAM, t, w, out, B, half);
    return check_launch("learned_sinu_fwd");
}
extern "C" int diqt_learned_sinu_bwd(const float* t, const float* w, const float* dout, float* dw, int B, int half,
                                     void* stream) {
    DIQT_REQUIRE(t && w && dout && dw, DIQT_E_ALIGN, "learned_sinu_bwd: null pointer");
    DIQT_REQUIRE(B > 0 && half > 0, DIQT_E_SHAPE, "learned_sinu_bwd: bad shape");
    hipLaunchKernelGGL(learned_sinu_bwd_kernel, dim3((half + 63) / 64), dim3(64), 0, STREAM, t, w, dout, dw, B, half);
    return check_launch("learned_sinu_bwd");
}

extern "C" int diqt_trilinear_up_fwd(const float* x, float* y, int B, int D, int H, int W, int C, int scale, void* stream) {
    DIQT_REQUIRE(x && y, DIQT_E_ALIGN, "trilinear_up_fwd: null pointer");
    DIQT_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0 && C > 0 && scale > 0, DIQT_E_SHAPE, "trilinear_up_fwd: bad shape");
    const size_t total = (size_t)B * D * H * W * C * scale * scale * scale;
    hipLaunchKernelGGL(trilinear_kernel, dim3(grid_for(total, 256)), dim3(256), 0, STREAM, x, y, B, D, H, W, C, scale, 0);
    return check_launch("trilinear_up_fwd");
}
extern "C" int diqt_trilinear_up_bwd(const float* dy, float* dx, int B, int D, int H, int W, int C, int scale, void* stream) {
    DIQT_REQUIRE(dy && dx, DIQT_E_ALIGN, "trilinear_up_bwd: null pointer");
    DIQT_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0 && C > 0 && scale > 0, DIQT_E_SHAPE, "trilinear_up_bwd: bad shape");
    const size_t total = (size_t)B * D * H * W * C * scale * scale * scale;
    hipLaunchKernelGGL(trilinear_kernel, dim3(grid_for(total, 256)), dim3(256), 0, STREAM, dy, dx, B, D, H, W, C, scale, 1);
    return check_launch("trilinear_up_bwd");
}

static int shuffle_nd(const float* x, float* y, int B, int D, int H, int W, int C, int sd, int sh, int sw, int toSpace,
                      void* stream, const char* what) {
    DIQT_REQUIRE(x && y, DIQT_E_ALIGN, "%s: null pointer", what);
    DIQT_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0 && C > 0, DIQT_E_SHAPE, "%s: bad shape", what);
    DIQT_REQUIRE(sd >= 1 && sd <= 8 && sh >= 1 && sh <= 8 && sw >= 1 && sw <= 8, DIQT_E_UNSUPPORTED, "%s: factors must be 1 .. 8", what);
    const size_t total = (size_t)B * sd * sh * sw * D * H * W * C;
    const int S = sd * sh * sw;
    const void* deepSide = toSpace ? (const void*)x : (const void*)y;
    if ((S == 2 || S == 4 || S == 8) && aligned16(deepSide)) {
        const dim3 grid(grid_for(total / S, 256));
        if (S == 2) hipLaunchKernelGGL(shuffle_nd_vec_kernel<2>, grid, dim3(256), 0, STREAM, x, y, B, D, H, W, C, sd, sh, sw, toSpace);
        else if (S == 4) hipLaunchKernelGGL(shuffle_nd_vec_kernel<4>, grid, dim3(256), 0, STREAM, x, y, B, D, H, W, C, sd, sh, sw, toSpace);
        else hipLaunchKernelGGL(shuffle_nd_vec_kernel<8>, grid, dim3(256), 0, STREAM, x, y, B, D, H, W, C, sd, sh, sw, toSpace);
        return check_launch(what);
    }
    hipLaunchKernelGGL(shuffle_nd_kernel, dim3(grid_for(total, 256)), dim3(256), 0, STREAM, x, y, B, D, H, W, C, sd, sh, sw, toSpace);
    return check_launch(what);
}
extern "C" int diqt_space_to_depth_nd(const float* x, float* y, int B, int D, int H, int W, int C, int sd, int sh, int sw,
                                      void* stream) {
    return shuffle_nd(x, y, B, D, H, W, C, sd, sh, sw, 0, stream, "space_to_depth_nd");
}
extern "C" int diqt_depth_to_space_nd(const float* x, float* y, int B, int D, int H, int W, int C, int sd, int sh, int sw,
                                      void* stream) {
    return shuffle_nd(x, y, B, D, H, W, C, sd, sh, sw, 1, stream, "depth_to_space_nd");
}
extern "C" int diqt_transpose_mid(const float* x, float* y, int A, int M, int N, int C, void* stream) {
    DIQT_REQUIRE(x && y, DIQT_E_ALIGN, "transpose_mid: null pointer");
    DIQT_REQUIRE(A > 0 && M > 0 && N > 0 && C > 0, DIQT_E_SHAPE, "transpose_mid: bad shape");
    hipLaunchKernelGGL(transpose_mid_kernel, dim3(grid_for((size_t)A * M * N * C, 256)), dim3(256), 0, STREAM, x, y, A, M, N, C);
    return check_launch("transpose_mid");
}
extern "C" int diqt_nearest_resize(const float* x, float* y, int B, int D, int H, int W, int C, int Do, int Ho, int Wo,
                                   void* stream) {
    DIQT_REQUIRE(x && y, DIQT_E_ALIGN, "nearest_resize: null pointer");
    DIQT_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0 && C > 0 && Do > 0 && Ho > 0 && Wo > 0, DIQT_E_SHAPE, "nearest_resize: bad shape");
    hipLaunchKernelGGL(nearest_resize_kernel, dim3(grid_for((size_t)B * Do * Ho * Wo * C, 256)), dim3(256), 0, STREAM, x, y, B,
                       D, H, W, C, Do, Ho, Wo);
    return check_launch("nearest_resize");
}
extern "C" int diqt_nearest_resize_bwd(const float* dy, float* dx, int B, int D, int H, int W, int C, int Do, int Ho, int Wo,
                                       void* stream) {
    DIQT_REQUIRE(dy && dx, DIQT_E_ALIGN, "nearest_resize_bwd: null pointer");
    DIQT_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0 && C > 0 && Do > 0 && Ho > 0 && Wo > 0, DIQT_E_SHAPE, "nearest_resize_bwd: bad shape");
    DIQT_REQUIRE(Do % D == 0 && Ho % H == 0 && Wo % W == 0, DIQT_E_UNSUPPORTED,
                 "nearest_resize_bwd: whole-number up-scaling factors only (%dx%dx%d -> %dx%dx%d)", D, H, W, Do, Ho, Wo);
    hipLaunchKernelGGL(nearest_resize_bwd_kernel, dim3(grid_for((size_t)B * D * H * W * C, 256)), dim3(256), 0, STREAM, dy, dx, B, D, H, W,
                       C, Do / D, Ho / H, Wo / W);
    return check_launch("nearest_resize_bwd");
}
// rows of d floats, x rows `x_stride` floats apart, y (and dy / dx of the backward) rows `y_stride` apart; inv[rows] (optional in the
// forward) keeps 1 / max(||x||, 1e-12) for the backward
extern "C" int diqt_l2norm_rows_fwd(const float* x, float* y, float* inv, size_t rows, int d, int x_stride, int y_stride, void* stream) {
    DIQT_REQUIRE(x && y, DIQT_E_ALIGN, "l2norm_rows_fwd: null pointer");
    DIQT_REQUIRE(d > 0 && x_stride >= d && y_stride >= d, DIQT_E_SHAPE, "l2norm_rows_fwd: bad shape");
    if (rows == 0) return DIQT_OK;
    hipLaunchKernelGGL(l2norm_rows_kernel, dim3(grid_for((rows + 15) / 16 * 16 * 16, 256)), dim3(256), 0, STREAM, x, y, inv, rows, d,
                       x_stride, y_stride);
    return check_launch("l2norm_rows_fwd");
}
extern "C" int diqt_l2norm_rows_bwd(const float* y, const float* dy, const float* inv, float* dx, size_t rows, int d, int y_stride,
                                    int dx_stride, void* stream) {
    DIQT_REQUIRE(y && dy && inv && dx, DIQT_E_ALIGN, "l2norm_rows_bwd: null pointer");
    DIQT_REQUIRE(d > 0 && y_stride >= d && dx_stride >= d, DIQT_E_SHAPE, "l2norm_rows_bwd: bad shape");
    if (rows == 0) return DIQT_OK;
    hipLaunchKernelGGL(l2norm_rows_bwd_kernel, dim3(grid_for((rows + 15) / 16 * 16 * 16, 256)), dim3(256), 0, STREAM, y, dy, inv, dx, rows,
                       d, y_stride, dx_stride);
    return check_launch("l2norm_rows_bwd");
}
extern "C" int diqt_attn_softmax_fwd(const float* sim, const float* rel, const float* null_bias, float* p, int G, int n,
                                     int h, int n_extra, int n_self, int causal, void* stream) {
    DIQT_REQUIRE(sim && p, DIQT_E_ALIGN, "attn_softmax_fwd: null pointer");
    DIQT_REQUIRE(G > 0 && n > 0 && h > 0 && n_extra >= 0 && n_self >= 0 && n_extra + n_self > 0, DIQT_E_SHAPE, "attn_softmax_fwd: bad shape");
    DIQT_REQUIRE(!(causal || rel) || n_self == n, DIQT_E_SHAPE, "attn_softmax_fwd: causal / relative bias need n_self == n");
    DIQT_REQUIRE(!null_bias || n_extra >= 1, DIQT_E_SHAPE, "attn_softmax_fwd: null bias without a null key");
    const size_t rows = (size_t)G * n * h;
    hipLaunchKernelGGL(attn_softmax_fwd_kernel, dim3(grid_for(rows, 4, 16384)), dim3(256), 0, STREAM, sim, rel, null_bias, p, rows,
                       n, h, n_extra, n_self, causal);
    return check_launch("attn_softmax_fwd");
}
extern "C" int diqt_attn_softmax_bwd(const float* p, const float* dp, float* dsim, float* drel, float* dnull_bias, int G,
                                     int n, int h, int n_extra, int n_self, int causal, void* stream);
// deterministic relative-bias gradient: every wave accumulates its rows into a private LDS table (for one row the keys map to
// distinct table entries, rows of a wave are sequential), the 4 tables of a block are summed in order into
// partial[block][(2ns-1)*h + h] and attn_rel_reduce_kernel sums the blocks in order.  (The plain kernel's global atomics took
// 6 ms per call on the temporal attention: 69 M atomics onto 504 addresses.)
__global__ __launch_bounds__(256) void attn_softmax_bwd_tbl_kernel(const float* __restrict__ p, const float* __restrict__ dp,
                                                                   float* __restrict__ dsim, float* __restrict__ partial,
                                                                   size_t rows, int n, int h, int E, int ns, int causal, int want_rel,
                                                                   int want_null) {
    extern __shared__ float tbl[];                       // [4 waves][T], T = (2ns-1)*h + h
    const int T = (2 * ns - 1) * h + h;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, M = E + ns;
    float* mine = tbl + (size_t)wv * T;
    for (int e = lane; e < T; e += 64) mine[e] = 0.f;
    for (size_t r = blockIdx.x * (size_t)4 + wv; r < rows; r += (size_t)gridDim.x * 4) {
        const int hh = (int)(r % h), i = (int)((r / h) % n);
        float s = 0.f;
        for (int j = lane; j < M; j += 64) s += p[r * M + j] * dp[r * M + j];
        s = wave_sum(s);
        for (int j = lane; j < M; j += 64) {
            const float d = p[r * M + j] * (dp[r * M + j] - s);      // zero where masked (p == 0)
            dsim[r * M + j] = d;
            if (j >= E) {
                const int jj = j - E;
                if (want_rel && !(causal && jj > i)) mine[(i - jj + ns - 1) * h + hh] += d;     // distinct entries within a row
            } else if (j == E - 1 && want_null) mine[(2 * ns - 1) * h + hh] += d;
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < T; e += 256)
        partial[(size_t)blockIdx.x * T + e] = ((tbl[e] + tbl[T + e]) + tbl[2 * T + e]) + tbl[3 * T + e];
}
// one wave per table entry: lanes stride over the block partials, fixed-order wave reduction
__global__ __launch_bounds__(256) void attn_rel_reduce_kernel(const float* __restrict__ partial, float* __restrict__ drel,
                                                              float* __restrict__ dnull, int nblk, int relElems, int h) {
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63, T = relElems + h;
    if (e >= T) return;
    float s = 0.f;
    for (int b = lane; b < nblk; b += 64) s += partial[(size_t)b * T + e];
    s = wave_sum(s);
    if (lane == 0) {
        if (e < relElems) { if (drel) drel[e] = s; }
        else if (dnull) dnull[e - relElems] = s;
    }
}

extern "C" size_t diqt_attn_softmax_bwd_workspace_bytes(int G, int n, int h, int n_extra, int n_self) {
    const size_t T = (size_t)(2 * n_self - 1) * h + h;
    if (T * 4 * sizeof(float) > 48 * 1024) return 0;              // table too large for LDS: the atomic kernel is used
    return (size_t)1024 * T * sizeof(float);
}

extern "C" int diqt_attn_softmax_bwd_ws(const float* p, const float* dp, float* dsim, float* drel, float* dnull_bias,
                                        void* workspace, size_t workspace_bytes, int G, int n, int h, int n_extra, int n_self,
                                        int causal, void* stream) {
    DIQT_REQUIRE(p && dp && dsim, DIQT_E_ALIGN, "attn_softmax_bwd: null pointer");
    DIQT_REQUIRE(G > 0 && n > 0 && h > 0 && n_extra >= 0 && n_self >= 0 && n_extra + n_self > 0, DIQT_E_SHAPE, "attn_softmax_bwd: bad shape");
    const size_t rows = (size_t)G * n * h;
    const size_t need = diqt_attn_softmax_bwd_workspace_bytes(G, n, h, n_extra, n_self);
    if (!(drel || dnull_bias) || need == 0 || !workspace || workspace_bytes < need)
        return diqt_attn_softmax_bwd(p, dp, dsim, drel, dnull_bias, G, n, h, n_extra, n_self, causal, stream);
    const int T = (2 * n_self - 1) * h + h;
    const unsigned nblk = grid_for(rows, 4, 1024);
    float* partial = static_cast<float*>(workspace);
    hipLaunchKernelGGL(attn_softmax_bwd_tbl_kernel, dim3(nblk), dim3(256), (size_t)4 * T * sizeof(float), STREAM, p, dp, dsim, partial,
                       rows, n, h, n_extra, n_self, causal, drel ? 1 : 0, dnull_bias ? 1 : 0);
    int rc = check_launch("attn_softmax_bwd(table)");
    if (rc) return rc;
    hipLaunchKernelGGL(attn_rel_reduce_kernel, dim3((T + 3) / 4), dim3(256), 0, STREAM, partial, drel, dnull_bias, (int)nblk,
                       T - h, h);
    return check_launch("attn_softmax_bwd(reduce)");
}

extern "C" int diqt_attn_softmax_bwd(const float* p, const float* dp, float* dsim, float* drel, float* dnull_bias, int G,
                                     int n, int h, int n_extra, int n_self, int causal, void* stream) {
    DIQT_REQUIRE(p && dp && dsim, DIQT_E_ALIGN, "attn_softmax_bwd: null pointer");
    DIQT_REQUIRE(G > 0 && n > 0 && h > 0 && n_extra >= 0 && n_self >= 0 && n_extra + n_self > 0, DIQT_E_SHAPE, "attn_softmax_bwd: bad shape");
    const size_t rows = (size_t)G * n * h;
    hipLaunchKernelGGL(attn_softmax_bwd_kernel, dim3(grid_for(rows, 4, 16384)), dim3(256), 0, STREAM, p, dp, dsim, drel, dnull_bias,
                       rows, n, h, n_extra, n_self, causal);
    return check_launch("attn_softmax_bwd");
}

static int ls_slices(int N) { int s = (N + 15) / 16; return s > 64 ? 64 : (s < 1 ? 1 : s); }

extern "C" size_t diqt_linear_small_workspace_bytes(int M, int K, int N) {
    return (size_t)ls_slices(N) * M * K * sizeof(float);
}

extern "C" int diqt_linear_small_fwd(const float* x, const float* W, const float* bias, float* y, int M, int K, int N,
                                     void* stream) {
    DIQT_REQUIRE(x && W && y, DIQT_E_ALIGN, "linear_small_fwd: null pointer");
    DIQT_REQUIRE(M > 0 && M <= 64 && K > 0 && N > 0, DIQT_E_SHAPE, "linear_small_fwd: needs 0 < M <= 64 (got %d)", M);
    if (K % 4 == 0 && aligned16(x) && aligned16(W))
        hipLaunchKernelGGL(linear_small_fwd_kernel<true>, dim3((N + 3) / 4), dim3(256), 0, STREAM, x, W, bias, y, M, K, N);
    else
        hipLaunchKernelGGL(linear_small_fwd_kernel<false>, dim3((N + 3) / 4), dim3(256), 0, STREAM, x, W, bias, y, M, K, N);
    return check_launch("linear_small_fwd");
}

extern "C" int diqt_linear_small_bwd(const float* x, const float* W, const float* dy, float* dx, float* dw, float* db,
                                     void* workspace, size_t workspace_bytes, int M, int K, int N, void* stream) {
    DIQT_REQUIRE(x && W && dy, DIQT_E_ALIGN, "linear_small_bwd: null pointer");
    DIQT_REQUIRE(M > 0 && M <= 64 && K > 0 && N > 0, DIQT_E_SHAPE, "linear_small_bwd: needs 0 < M <= 64 (got %d)", M);
    int rc = DIQT_OK;
    if (dw) {
        hipLaunchKernelGGL(linear_small_dw_kernel, dim3((unsigned)(((size_t)N * K + 255) / 256)), dim3(256), 0, STREAM, x, dy, dw,
                           db, M, K, N);
        rc = check_launch("linear_small_bwd/dw");
        if (rc) return rc;
    }
    if (dx) {
        const int slices = ls_slices(N), nPer = (N + slices - 1) / slices;
        DIQT_REQUIRE(workspace && workspace_bytes >= diqt_linear_small_workspace_bytes(M, K, N), DIQT_E_WORKSPACE,
                     "linear_small_bwd: workspace too small");
        float* part = static_cast<float*>(workspace);
        hipLaunchKernelGGL(linear_small_dx_part_kernel, dim3((K + 255) / 256, slices), dim3(256), 0, STREAM, W, dy, part, M, K,
                           N, nPer);
        rc = check_launch("linear_small_bwd/dx_part");
        if (rc) return rc;
        hipLaunchKernelGGL(linear_small_dx_sum_kernel, dim3((M * K + 255) / 256), dim3(256), 0, STREAM, part, dx, M * K, slices);
        rc = check_launch("linear_small_bwd/dx_sum");
    }
    return rc;
}

extern "C" int diqt_abs_quantile(const float* x, float* out, int B, size_t per, unsigned k_lo, float weight, void* stream) {
    DIQT_REQUIRE(x && out, DIQT_E_ALIGN, "abs_quantile: null pointer");
    DIQT_REQUIRE(B > 0 && per > 0 && k_lo < per && weight >= 0.f && weight < 1.f, DIQT_E_SHAPE, "abs_quantile: bad rank");
    const int need_hi = (weight > 0.f && (size_t)k_lo + 1 < per) ? 1 : 0;
    hipLaunchKernelGGL(abs_quantile_kernel, dim3(B), dim3(1024), 0, STREAM, x, out, per, k_lo, need_hi, weight);
    return check_launch("abs_quantile");
}
extern "C" int diqt_dynamic_threshold(const float* x0, const float* s, float* out, int B, size_t per, void* stream) {
    DIQT_REQUIRE(x0 && s && out, DIQT_E_ALIGN, "dynamic_threshold: null pointer");
    DIQT_REQUIRE(B > 0 && per > 0, DIQT_E_SHAPE, "dynamic_threshold: bad shape");
    hipLaunchKernelGGL(dynamic_threshold_kernel, dim3(grid_for(per, 256, 1024), B), dim3(256), 0, STREAM, x0, s, out, per);
    return check_launch("dynamic_threshold");
}
extern "C" int diqt_mask_blend(const float* x, const float* y, const float* mask, float* out, size_t n, void* stream) {
    DIQT_REQUIRE(x && y && mask && out, DIQT_E_ALIGN, "mask_blend: null pointer");
    if (n == 0) return DIQT_OK;
    hipLaunchKernelGGL(mask_blend_kernel, dim3(grid_for(n, 256, 4096)), dim3(256), 0, STREAM, x, y, mask, out, n);
    return check_launch("mask_blend");
}

extern "C" int diqt_patch_gather(const float* vol, const int* idx, float* out, int* nonzero, int n_patches, int D, int H, int W,
                                 int P, float mean, float stdv, void* stream) {
    DIQT_REQUIRE(vol && idx && (out || nonzero), DIQT_E_ALIGN, "patch_gather: null pointer");
    DIQT_REQUIRE(D > 0 && H > 0 && W > 0 && P > 0 && P <= D && P <= H && P <= W, DIQT_E_SHAPE, "patch_gather: bad shape");
    if (n_patches <= 0) return DIQT_OK;
    DIQT_REQUIRE(n_patches <= 65535, DIQT_E_SHAPE, "patch_gather: at most 65535 patches per call");
    if (nonzero) {
        hipError_t e = hipMemsetAsync(nonzero, 0, (size_t)n_patches * sizeof(int), STREAM);
        DIQT_REQUIRE(e == hipSuccess, DIQT_E_LAUNCH, "patch_gather: hipMemsetAsync: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(patch_gather_kernel, dim3(grid_for((size_t)P * P * P, 256, 64), n_patches), dim3(256), 0, STREAM, vol, idx, out,
                       nonzero, D, H, W, P, mean, stdv);
    return check_launch("patch_gather");
}
extern "C" int diqt_patch_scatter(const float* patches, const int* idx, const int* margins, float* pred, int n_patches, int D,
                                  int H, int W, int P, void* stream) {
    DIQT_REQUIRE(patches && idx && margins && pred, DIQT_E_ALIGN, "patch_scatter: null pointer");
    DIQT_REQUIRE(D > 0 && H > 0 && W > 0 && P > 0 && P <= D && P <= H && P <= W, DIQT_E_SHAPE, "patch_scatter: bad shape");
    if (n_patches <= 0) return DIQT_OK;
    DIQT_REQUIRE(n_patches <= 65535, DIQT_E_SHAPE, "patch_scatter: at most 65535 patches per call");
    hipLaunchKernelGGL(patch_scatter_kernel, dim3(grid_for((size_t)P * P * P, 256, 64), n_patches), dim3(256), 0, STREAM, patches, idx,
                       margins, pred, D, H, W, P);
    return check_launch("patch_scatter");
}
extern "C" int diqt_background_reset(float* pred, const float* vol, size_t n, float mean, float stdv, float min_val, void* stream) {
    DIQT_REQUIRE(pred && vol, DIQT_E_ALIGN, "background_reset: null pointer");
    if (n == 0) return DIQT_OK;
    hipLaunchKernelGGL(background_reset_kernel, dim3(grid_for(n, 256, 4096)), dim3(256), 0, STREAM, pred, vol, n, mean, stdv, min_val);
    return check_launch("background_reset");
}
extern "C" int diqt_min_value(const float* x, size_t n, float* workspace_1024, float* out, void* stream) {
    DIQT_REQUIRE(x && workspace_1024 && out && n > 0, DIQT_E_ALIGN, "min_value: null pointer / empty input");
    const unsigned nb = grid_for(n, 256, 1024);
    hipLaunchKernelGGL(min_stage1_kernel, dim3(nb), dim3(256), 0, STREAM, x, workspace_1024, n);
    int rc = check_launch("min_value/stage1");
    if (rc) return rc;
    hipLaunchKernelGGL(min_stage2_kernel, dim3(1), dim3(64), 0, STREAM, workspace_1024, (int)nb, out);
    return check_launch("min_value/stage2");
}
