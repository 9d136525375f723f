// Fused multi-query attention forward (imagen_video.py:410-525 Attention.forward; sampling / no-grad path):
//   out[g, i, hh, :] = softmax_j( scale * q[g,i,hh,:] . k[g,j,:] + bias(i, j, hh) ) @ v[g,j,:]
// with ONE key/value head shared by all query heads, `E` extra keys in front (conditioning tokens ..., learned null key last),
// an optional T5-style relative position bias table on the self keys (+ null_bias[hh] on the null key) and an optional causal
// mask -- without materialising the [G, n*h, E+n] score tensor (1 GB per call at the C5 mid level).
//
// Transposed flash formulation on v_mfma_f32_32x32x2_f32: a wave owns 32 query rows (a row = (token, head)) as the COLUMNS of
//   S^T[key][query] = K Q^T          (A = K tile from LDS as b128 fragments, B = Q^T kept in registers for the whole kernel)
// so the softmax statistics of a query live in one lane (16 accumulator registers + one cross-half shuffle), and the
// probabilities P^T sit in the accumulator registers in exactly the lane layout the next MFMA wants for its B operand:
//   O^T[d][query] += V^T P^T         (A = V tile from LDS, B = P^T straight from the S^T registers, no LDS round trip).
// Key pair of MFMA step s: rows (s&3) + 8*(s>>2) + 4*half of the 32-key tile, for both operands.
#include "common.h"

namespace diqt {
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int AQ = 128;     // query rows per workgroup (4 waves x 32)
constexpr int AKT = 32;     // keys per tile

// score of (query row, key j) after the additive terms and masks of Attention.forward (imagen_video.py:490-518): T5-style relative
// bias on the self keys, the null-key bias on the last extra key, -inf for causal-masked and out-of-range keys.  Branch-free on
// purpose (clamped unconditional gather + selects): a guarded form compiles to four exec-mask branches per element, each gather
// behind its own s_waitcnt vmcnt(0) -- which also drains the next tile's prefetch.
__device__ __forceinline__ float attn_bias_mask(float v, int j, int M, int E, int ns, int h, int qi, int qh, int causal,
                                                const float* __restrict__ rel, float nbv) {
    const int jj = j - E;
    const bool self = j >= E;
    float b = (j == E - 1) ? nbv : 0.f;
    if (rel) {                                          // kernel-uniform
        const int idx = max(0, min(qi - jj + ns - 1, 2 * ns - 2));
        const float rv = rel[(size_t)idx * h + qh];
        b = self ? rv : b;
    }
    const bool dead = j >= M || (causal && self && jj > qi);
    return dead ? -INFINITY : v + b;
}

template <int ND>           // dim_head = 32 * ND
__global__ __launch_bounds__(256, 2) void mqa_flash_fwd_kernel(const float* __restrict__ q, const float* __restrict__ kv,
                                                               const float* __restrict__ rel, const float* __restrict__ null_bias,
                                                               float* __restrict__ out, int n, int h, int E, int ns, int causal,
                                                               float scale) {
    constexpr int D = 32 * ND, ROW = D + 4, NPF = AKT * (2 * D / 4) / 256;    // float4 pieces of a K|V tile per thread
    __shared__ __attribute__((aligned(16))) float KVs[2][2][AKT * ROW];       // [buffer][K | V][key][ROW]
    const int g = blockIdx.y;
    const int M = E + ns, R = n * h;                       // keys, query rows of this batch entry
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hf = lane >> 5;
    const int r = blockIdx.x * AQ + wave * 32 + l31;       // this lane's query row (column of S^T)
    const bool rvalid = r < R;
    const int rc = rvalid ? r : R - 1;
    const int qi = rc / h, qh = rc % h;                    // token index, head
    const float* qg = q + ((size_t)g * R + rc) * D;
    const float* kvg = kv + (size_t)g * M * 2 * D;

    // Q^T operand: MFMA step (group gq, element e) of a key tile uses k-index dd = 8*gq + 4*hf + e  (same order as the K fragments)
    float qreg[D / 2];
#pragma unroll
    for (int gq = 0; gq < D / 8; ++gq) {
        const float4 v = *reinterpret_cast<const float4*>(qg + 8 * gq + 4 * hf);
        qreg[4 * gq] = v.x * scale; qreg[4 * gq + 1] = v.y * scale; qreg[4 * gq + 2] = v.z * scale; qreg[4 * gq + 3] = v.w * scale;
    }
    f32x16 o[ND];
#pragma unroll
    for (int c = 0; c < ND; ++c)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[c][i] = 0.f;
    float mrun = -INFINITY, lrun = 0.f;
    const float nbv = null_bias ? null_bias[qh] : 0.f;

    // K | V tile of keys [32t, 32t+32): kv row = [k(D) | v(D)].  The next tile is loaded global -> registers BEFORE this tile's
    // MFMAs and written to the other LDS buffer after them: one barrier per tile and no memory round trip between tiles.
    // Loads are unconditional (clamped key, zero-selected): a guarded load becomes an exec-masked branch with its own wait.
    float4 pre[NPF];
    auto load_tile = [&](int t) {
#pragma unroll
        for (int u = 0; u < NPF; ++u) {
            const int e = u * 256 + tid;
            const int key = e / (2 * D / 4), c4 = (e % (2 * D / 4)) * 4;
            const int j = t * AKT + key;
            const float4 v = *reinterpret_cast<const float4*>(kvg + (size_t)min(j, M - 1) * 2 * D + c4);
            pre[u] = j < M ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int u = 0; u < NPF; ++u) {
            const int e = u * 256 + tid;
            const int key = e / (2 * D / 4), c4 = (e % (2 * D / 4)) * 4;
            if (c4 < D) *reinterpret_cast<float4*>(&KVs[buf][0][key * ROW + c4]) = pre[u];
            else *reinterpret_cast<float4*>(&KVs[buf][1][key * ROW + (c4 - D)]) = pre[u];
        }
    };
    const int ntiles = (M + AKT - 1) / AKT;
    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
        const float* Ks = KVs[t & 1][0];
        const float* Vs = KVs[t & 1][1];
        if (t + 1 < ntiles) load_tile(t + 1);
        // ---- S^T = K Q^T ----
        f32x16 s;
#pragma unroll
        for (int i = 0; i < 16; ++i) s[i] = 0.f;
        const float* kp = Ks + l31 * ROW + 4 * hf;
#pragma unroll
        for (int gq = 0; gq < D / 8; ++gq) {
            const float4 a = *reinterpret_cast<const float4*>(kp + 8 * gq);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, qreg[4 * gq], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, qreg[4 * gq + 1], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, qreg[4 * gq + 2], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, qreg[4 * gq + 3], s, 0, 0, 0);
        }
        // ---- bias, mask, online softmax (per lane = per query; rows of s are keys) ----
        float tmax = -INFINITY;
        const bool plain = !causal && !rel && t * AKT >= E && (t + 1) * AKT <= M;     // every key of the tile is an unbiased self key
        if (!plain) {
#pragma unroll
            for (int i = 0; i < 16; ++i)
                s[i] = attn_bias_mask(s[i], t * AKT + (i & 3) + 8 * (i >> 2) + 4 * hf, M, E, ns, h, qi, qh, causal, rel, nbv);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) tmax = fmaxf(tmax, s[i]);
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float mnew = fmaxf(mrun, tmax);
        const float corr = (mnew == -INFINITY) ? 1.f : __expf(mrun - mnew);      // mrun = -inf on the first tile -> 0
        float psum = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float p = (mnew == -INFINITY) ? 0.f : __expf(s[i] - mnew);
            s[i] = p;
            psum += p;
        }
        psum += __shfl_xor(psum, 32, 64);
        lrun = lrun * corr + psum;
        mrun = mnew;
        if (!__all(corr == 1.f)) {                        // wave-uniform: the running maxima settle after the first tiles
#pragma unroll
            for (int c = 0; c < ND; ++c)
#pragma unroll
                for (int i = 0; i < 16; ++i) o[c][i] *= corr;
        }
        // ---- O^T += V^T P^T : step i uses the key pair held in register i of the two lane halves ----
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int key = (i & 3) + 8 * (i >> 2) + 4 * hf;
            const float* vp = Vs + key * ROW + l31;
#pragma unroll
            for (int c = 0; c < ND; ++c) o[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(vp[32 * c], s[i], o[c], 0, 0, 0);
        }
        if (t + 1 < ntiles) store_tile((t + 1) & 1);       // that buffer was last read in tile t-1, retired by its barrier
        __syncthreads();
    }
    // ---- epilogue: O^T rows are head-dim indices, columns are queries: out[g, r, dd] = o / l ----
    if (rvalid) {
        const float inv = 1.f / lrun;
        float* og = out + ((size_t)g * R + r) * D;
#pragma unroll
        for (int c = 0; c < ND; ++c)
#pragma unroll
            for (int i = 0; i < 16; i += 4) {
                // registers i..i+3 of a lane are 4 consecutive head-dim rows: (i&3)=0..3 -> dd = 8*(i>>2) + 4*hf + 0..3
                const int dd = 32 * c + 8 * (i >> 2) + 4 * hf;
                *reinterpret_cast<float4*>(og + dd) = make_float4(o[c][i] * inv, o[c][i + 1] * inv, o[c][i + 2] * inv, o[c][i + 3] * inv);
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Mixed-precision variant (torch.autocast: the reference's q k^T and attn v einsums run in fp16 / bf16, its soft-max in fp32):
// the same transposed flash formulation on v_mfma_f32_32x32x16_{f16,bf16}.  64 keys per tile:
//   S^T[key][query] = K Q^T : A = K rows from LDS (8 halves of d per lane and k-step), B = Q^T, scaled and rounded once, in registers;
//   soft-max statistics in fp32 on the accumulator registers (a query = a lane);
//   O^T[d][query] += V^T P^T : B = P^T packed to 16 bit straight from the accumulator registers.  A 32x32 accumulator holds key
//   16s + 8(j>>2) + 4*half + (j&3) in register 8(s&1) + j, so the V tile is written TRANSPOSED into LDS with its keys in that order
//   (position 16s + 8*half + j) and a lane reads its 8 keys of a k-step with one ds_read_b128.
// q, kv, out stay fp32 in HBM (cast while staging), like the mixed-precision conv.
// ---------------------------------------------------------------------------------------------------------------------------------
typedef unsigned u32x4a __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8a __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8a __attribute__((ext_vector_type(8)));
constexpr int HKT = 64;      // keys per tile

template <bool BF>
__device__ __forceinline__ unsigned apack2(float a, float b) {
    if (BF) {
        typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
        bf2 v = {(__bf16)a, (__bf16)b};
        return __builtin_bit_cast(unsigned, v);
    } else {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        h2 v = {(_Float16)a, (_Float16)b};
        return __builtin_bit_cast(unsigned, v);
    }
}
template <bool BF>
__device__ __forceinline__ unsigned short apack1(float a) { return (unsigned short)(apack2<BF>(a, 0.f) & 0xffffu); }
template <bool BF>
__device__ __forceinline__ f32x16 amfma16(u32x4a a, u32x4a b, f32x16 c) {
    if (BF) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8a, a), __builtin_bit_cast(bf16x8a, b), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8a, a), __builtin_bit_cast(f16x8a, b), c, 0, 0, 0);
}

constexpr int AQH = 256;    // query rows per workgroup of the mixed-precision kernel (8 waves x 32): every workgroup streams ALL keys of
                            // its batch entry, so K|V traffic is (query rows / AQH) x the K|V size -- the bound once the MFMAs are 16x faster

// fp32 -> 16-bit copy of the K|V rows (read once per call instead of once per workgroup as fp32)
template <bool BF>
__global__ __launch_bounds__(256) void cast_to_h_kernel(const float* __restrict__ x, unsigned* __restrict__ y, size_t npairs) {
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < npairs; i += (size_t)gridDim.x * 256) {
        const float2 v = *reinterpret_cast<const float2*>(x + 2 * i);
        y[i] = apack2<BF>(v.x, v.y);
    }
}

template <int ND, bool BF>   // dim_head = 32 * ND
__global__ __launch_bounds__(512, 1) void mqa_flash_fwd_h_kernel(const float* __restrict__ q, const unsigned short* __restrict__ kv,
                                                                 const float* __restrict__ rel, const float* __restrict__ null_bias,
                                                                 float* __restrict__ out, int n, int h, int E, int ns, int causal,
                                                                 float scale, int round_out) {
    constexpr int D = 32 * ND, KROWB = 2 * D + 16, VROWB = 2 * HKT + 16, NPF = HKT * (2 * D / 8) / 512;   // 16-byte pieces per thread
    static_assert(NPF >= 1, "tile too small for 512 threads");
    __shared__ __attribute__((aligned(16))) unsigned char Ksm[2][HKT * KROWB];    // [buffer][key][d] 16-bit, padded rows
    __shared__ __attribute__((aligned(16))) unsigned char Vsm[2][D * VROWB];      // [buffer][d][key position] 16-bit, padded rows
    const int g = blockIdx.y;
    const int M = E + ns, R = n * h;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hf = lane >> 5;
    const int r = blockIdx.x * AQH + wave * 32 + l31;
    const bool rvalid = r < R;
    const int rc = rvalid ? r : R - 1;
    const int qi = rc / h, qh = rc % h;
    const float* qg = q + ((size_t)g * R + rc) * D;
    const unsigned short* kvg = kv + (size_t)g * M * 2 * D;

    // Q^T operand of k-step s: d = 16 s + 8 hf + j, j = 0..7
    u32x4a qreg[D / 16];
#pragma unroll
    for (int sx = 0; sx < D / 16; ++sx) {
        const float4 v0 = *reinterpret_cast<const float4*>(qg + 16 * sx + 8 * hf);
        const float4 v1 = *reinterpret_cast<const float4*>(qg + 16 * sx + 8 * hf + 4);
        qreg[sx].x = apack2<BF>(v0.x * scale, v0.y * scale); qreg[sx].y = apack2<BF>(v0.z * scale, v0.w * scale);
        qreg[sx].z = apack2<BF>(v1.x * scale, v1.y * scale); qreg[sx].w = apack2<BF>(v1.z * scale, v1.w * scale);
    }
    f32x16 o[ND];
#pragma unroll
    for (int c = 0; c < ND; ++c)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[c][i] = 0.f;
    float mrun = -INFINITY, lrun = 0.f;
    const float nbv = null_bias ? null_bias[qh] : 0.f;

    u32x4a pre[NPF];
    auto load_tile = [&](int t) {
#pragma unroll
        for (int u = 0; u < NPF; ++u) {
            const int e = u * 512 + tid;
            const int key = e / (2 * D / 8), c8 = (e % (2 * D / 8)) * 8;
            const int j = t * HKT + key;
            const u32x4a v = *reinterpret_cast<const u32x4a*>(kvg + (size_t)min(j, M - 1) * 2 * D + c8);
            const u32x4a z = {0u, 0u, 0u, 0u};
            pre[u] = j < M ? v : z;
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int u = 0; u < NPF; ++u) {
            const int e = u * 512 + tid;
            const int key = e / (2 * D / 8), c8 = (e % (2 * D / 8)) * 8;
            if (c8 < D) {
                *reinterpret_cast<u32x4a*>(&Ksm[buf][key * KROWB + c8 * 2]) = pre[u];
            } else {
                // key = 16 s + 8 a + 4 half + b  ->  position 16 s + 8 half + 4 a + b
                const int kk = key & 15, pos = (key & ~15) + 8 * ((kk >> 2) & 1) + 4 * (kk >> 3) + (kk & 3);
                unsigned char* vb = &Vsm[buf][(c8 - D) * VROWB + pos * 2];
                const unsigned w0 = pre[u].x, w1 = pre[u].y, w2 = pre[u].z, w3 = pre[u].w;      // by value (component bit-casts, see conv_half.hip)
                *reinterpret_cast<unsigned short*>(vb) = (unsigned short)(w0 & 0xffffu);
                *reinterpret_cast<unsigned short*>(vb + VROWB) = (unsigned short)(w0 >> 16);
                *reinterpret_cast<unsigned short*>(vb + 2 * VROWB) = (unsigned short)(w1 & 0xffffu);
                *reinterpret_cast<unsigned short*>(vb + 3 * VROWB) = (unsigned short)(w1 >> 16);
                *reinterpret_cast<unsigned short*>(vb + 4 * VROWB) = (unsigned short)(w2 & 0xffffu);
                *reinterpret_cast<unsigned short*>(vb + 5 * VROWB) = (unsigned short)(w2 >> 16);
                *reinterpret_cast<unsigned short*>(vb + 6 * VROWB) = (unsigned short)(w3 & 0xffffu);
                *reinterpret_cast<unsigned short*>(vb + 7 * VROWB) = (unsigned short)(w3 >> 16);
            }
        }
    };
    const int ntiles = (M + HKT - 1) / HKT;
    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
        const unsigned char* Kb = Ksm[t & 1];
        const unsigned char* Vb = Vsm[t & 1];
        if (t + 1 < ntiles) load_tile(t + 1);
        // ---- S^T = K Q^T for the two 32-key blocks ----
        f32x16 sacc[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int i = 0; i < 16; ++i) sacc[kb][i] = 0.f;
            const unsigned char* kp = Kb + (kb * 32 + l31) * KROWB + hf * 16;
#pragma unroll
            for (int sx = 0; sx < D / 16; ++sx)
                sacc[kb] = amfma16<BF>(*reinterpret_cast<const u32x4a*>(kp + sx * 32), qreg[sx], sacc[kb]);
        }
        // ---- bias, mask, online soft-max in fp32 (rows of sacc are keys, a lane is a query) ----
        float tmax = -INFINITY;
        const bool plain = !causal && !rel && t * HKT >= E && (t + 1) * HKT <= M;
        if (!plain) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    sacc[kb][i] = attn_bias_mask(sacc[kb][i], t * HKT + kb * 32 + (i & 3) + 8 * (i >> 2) + 4 * hf, M, E, ns, h, qi, qh,
                                                 causal, rel, nbv);
        }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 16; ++i) tmax = fmaxf(tmax, sacc[kb][i]);
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float mnew = fmaxf(mrun, tmax);
        const float corr = (mnew == -INFINITY) ? 1.f : __expf(mrun - mnew);
        float psum = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float pv = (mnew == -INFINITY) ? 0.f : __expf(sacc[kb][i] - mnew);
                sacc[kb][i] = pv;
                psum += pv;
            }
        psum += __shfl_xor(psum, 32, 64);
        lrun = lrun * corr + psum;
        mrun = mnew;
        if (!__all(corr == 1.f)) {                        // wave-uniform: the running maxima settle after the first tiles
#pragma unroll
            for (int c = 0; c < ND; ++c)
#pragma unroll
                for (int i = 0; i < 16; ++i) o[c][i] *= corr;
        }
        // ---- O^T += V^T P^T : k-step s = 16 keys, P^T packed from registers 8(s&1)..+7 of block s>>1 ----
#pragma unroll
        for (int sx = 0; sx < HKT / 16; ++sx) {
            const f32x16& pb = sacc[sx >> 1];
            const int i0 = 8 * (sx & 1);
            u32x4a pk;
            pk.x = apack2<BF>(pb[i0], pb[i0 + 1]); pk.y = apack2<BF>(pb[i0 + 2], pb[i0 + 3]);
            pk.z = apack2<BF>(pb[i0 + 4], pb[i0 + 5]); pk.w = apack2<BF>(pb[i0 + 6], pb[i0 + 7]);
#pragma unroll
            for (int c = 0; c < ND; ++c)
                o[c] = amfma16<BF>(*reinterpret_cast<const u32x4a*>(Vb + (32 * c + l31) * VROWB + sx * 32 + hf * 16), pk, o[c]);
        }
        if (t + 1 < ntiles) store_tile((t + 1) & 1);
        __syncthreads();
    }
    if (rvalid) {
        const float inv = 1.f / lrun;
        float* og = out + ((size_t)g * R + r) * D;
#pragma unroll
        for (int c = 0; c < ND; ++c)
#pragma unroll
            for (int i = 0; i < 16; i += 4) {
                const int dd = 32 * c + 8 * (i >> 2) + 4 * hf;
                float4 v = make_float4(o[c][i] * inv, o[c][i + 1] * inv, o[c][i + 2] * inv, o[c][i + 3] * inv);
                if (round_out) {
                    if (BF) { v.x = (float)(__bf16)v.x; v.y = (float)(__bf16)v.y; v.z = (float)(__bf16)v.z; v.w = (float)(__bf16)v.w; }
                    else { v.x = (float)(_Float16)v.x; v.y = (float)(_Float16)v.y; v.z = (float)(_Float16)v.z; v.w = (float)(_Float16)v.w; }
                }
                *reinterpret_cast<float4*>(og + dd) = v;
            }
    }
}
}  // namespace diqt

using namespace diqt;

extern "C" int diqt_mqa_attention_fwd(const float* q, const float* kv, const float* rel, const float* null_bias, float* out, int G,
                                      int n, int h, int d, int n_extra, int n_self, int causal, float scale, void* stream) {
    DIQT_REQUIRE(q && kv && out, DIQT_E_ALIGN, "mqa_attention_fwd: null pointer");
    DIQT_REQUIRE(G > 0 && n > 0 && h > 0 && n_extra >= 0 && n_self >= 0 && n_extra + n_self > 0, DIQT_E_SHAPE, "mqa_attention_fwd: bad shape");
    DIQT_REQUIRE(d == 32 || d == 64, DIQT_E_UNSUPPORTED, "mqa_attention_fwd: dim_head %d (32 or 64 are built)", d);
    DIQT_REQUIRE(!(causal || rel) || n_self == n, DIQT_E_SHAPE, "mqa_attention_fwd: causal / relative bias need n_self == n");
    DIQT_REQUIRE(!null_bias || n_extra >= 1, DIQT_E_SHAPE, "mqa_attention_fwd: null bias without a null key");
    DIQT_REQUIRE(aligned16(q) && aligned16(kv) && aligned16(out), DIQT_E_ALIGN, "mqa_attention_fwd: pointers must be 16-byte aligned");
    DIQT_REQUIRE(G <= 65535, DIQT_E_SHAPE, "mqa_attention_fwd: G > 65535");
    const dim3 grid((unsigned)(((long long)n * h + AQ - 1) / AQ), G);
    if (d == 64)
        hipLaunchKernelGGL(mqa_flash_fwd_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, q, kv, rel, null_bias, out, n, h, n_extra,
                           n_self, causal, scale);
    else
        hipLaunchKernelGGL(mqa_flash_fwd_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, q, kv, rel, null_bias, out, n, h, n_extra,
                           n_self, causal, scale);
    return check_launch("mqa_attention_fwd");
}

extern "C" int diqt_cast_to_h(const float* x, void* y, size_t n, int bf16, void* stream) {
    DIQT_REQUIRE(x && y, DIQT_E_ALIGN, "cast_to_h: null pointer");
    DIQT_REQUIRE(n % 2 == 0 && (reinterpret_cast<uintptr_t>(x) & 7u) == 0 && (reinterpret_cast<uintptr_t>(y) & 3u) == 0, DIQT_E_ALIGN,
                 "cast_to_h: even element count, 8-byte aligned input and 4-byte aligned output required");
    if (n == 0) return DIQT_OK;
    auto k = bf16 ? cast_to_h_kernel<true> : cast_to_h_kernel<false>;
    hipLaunchKernelGGL(k, dim3(grid_for(n / 2, 256)), dim3(256), 0, (hipStream_t)stream, x, static_cast<unsigned*>(y), n / 2);
    return check_launch("cast_to_h");
}

extern "C" int diqt_mqa_attention_fwd_h(const float* q, const void* kv, const float* rel, const float* null_bias, float* out, int G,
                                        int n, int h, int d, int n_extra, int n_self, int causal, float scale, int bf16, int round_out,
                                        void* stream) {
    DIQT_REQUIRE(q && kv && out, DIQT_E_ALIGN, "mqa_attention_fwd_h: null pointer");
    DIQT_REQUIRE(G > 0 && n > 0 && h > 0 && n_extra >= 0 && n_self >= 0 && n_extra + n_self > 0, DIQT_E_SHAPE, "mqa_attention_fwd_h: bad shape");
    DIQT_REQUIRE(d == 32 || d == 64, DIQT_E_UNSUPPORTED, "mqa_attention_fwd_h: dim_head %d (32 or 64 are built)", d);
    DIQT_REQUIRE(!(causal || rel) || n_self == n, DIQT_E_SHAPE, "mqa_attention_fwd_h: causal / relative bias need n_self == n");
    DIQT_REQUIRE(!null_bias || n_extra >= 1, DIQT_E_SHAPE, "mqa_attention_fwd_h: null bias without a null key");
    DIQT_REQUIRE(aligned16(q) && aligned16(kv) && aligned16(out), DIQT_E_ALIGN, "mqa_attention_fwd_h: pointers must be 16-byte aligned");
    DIQT_REQUIRE(G <= 65535, DIQT_E_SHAPE, "mqa_attention_fwd_h: G > 65535");
    const dim3 grid((unsigned)(((long long)n * h + AQH - 1) / AQH), G);
    void (*k)(const float*, const unsigned short*, const float*, const float*, float*, int, int, int, int, int, float, int) =
        d == 64 ? (bf16 ? mqa_flash_fwd_h_kernel<2, true> : mqa_flash_fwd_h_kernel<2, false>)
                : (bf16 ? mqa_flash_fwd_h_kernel<1, true> : mqa_flash_fwd_h_kernel<1, false>);
    hipLaunchKernelGGL(k, grid, dim3(512), 0, (hipStream_t)stream, q, static_cast<const unsigned short*>(kv), rel, null_bias, out, n, h,
                       n_extra, n_self, causal, scale, round_out ? 1 : 0);
    return check_launch("mqa_attention_fwd_h");
}
