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
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int j = t * AKT + (i & 3) + 8 * (i >> 2) + 4 * hf;
            float v = s[i];
            if (j >= M) v = -INFINITY;
            else if (j >= E) {
                const int jj = j - E;
                if (causal && jj > qi) v = -INFINITY;
                else if (rel) v += rel[(size_t)(qi - jj + ns - 1) * h + qh];
            } else if (j == E - 1 && null_bias) v += null_bias[qh];
            s[i] = v;
            tmax = fmaxf(tmax, v);
        }
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
#pragma unroll
        for (int c = 0; c < ND; ++c)
#pragma unroll
            for (int i = 0; i < 16; ++i) o[c][i] *= corr;
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
