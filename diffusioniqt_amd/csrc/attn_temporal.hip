// The temporal attention block of the pseudo-3D U-Net as ONE kernel (sampling path under autocast: fp16 / bf16 MFMA operands, fp32
// accumulation and statistics):
//     y = LayerNorm_out( Attention( LayerNorm(x) ) W_o ) + x            per sequence = the F frames of one (batch entry, pixel)
// (imagen_video.py:410-525 Attention.forward inside Residual(EinopsToAndFrom('b c f h w', '(b h w) f c', ...)), :1351-1354).
//
// Why one kernel: with heads x dim_head = 512 the queries and the attention output are 8x wider than x (4.3 GB each per level-0 call of
// the 64^3 stage), so the unfused chain -- transpose, LayerNorm, to_q, to_kv, concat + cast, attention, to_out, LayerNorm + residual,
// transpose -- moves ~20 GB through HBM for a block whose input and output are 0.5 GB each.  Here a workgroup owns a whole sequence:
// x rows are gathered in place (frame stride P*C, no transposes), q, k, v, the scores and the head outputs never leave the CU.
//
// All products run transposed on v_mfma_f32_32x32x16_{f16,bf16} so that a token stays on a LANE from the projection to the output:
//     q^T  = W_q  xn^T        A = W_q rows (registers / L2), B = xn^T from the LDS image of the normalised sequence
//     S^T  = k   q^T + bias   A = k rows from LDS, B = q^T straight from the accumulator registers; the accumulator starts at the bias
//     o^T  = v^T P^T          A = v^T rows from LDS, B = P^T from the S^T registers (soft-max statistics: one lane per query)
//     y^T += W_o^T o^T        A = W_o^T rows (registers / L2), B = o^T from the accumulator registers
// An accumulator tile used as the next B operand presents the reduction index in the order 4 hf + 8 g + e; the A side of that product
// stores its reduction axis in the same order (LDS positions of k's channels and v^T's keys, host-packed W_o).
// Wave w handles head w (8 waves, two per SIMD) or heads w, w + 4 (4 waves); the weights of its heads live in registers for the whole
// kernel when C == 64; the per-wave
// partial y^T go through LDS and are summed in a fixed order (deterministic), then LayerNorm + residual + store.
// The null key / value (imagen_video.py:471-481) is one extra score per query, handled on the VALU.
#include "common.h"
#include <stdlib.h>

namespace diqt {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4t __attribute__((ext_vector_type(4)));
typedef unsigned u32x2t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8t __attribute__((ext_vector_type(8)));

template <bool BF>
__device__ __forceinline__ unsigned tpack2(float a, float b) {
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
__device__ __forceinline__ float tround(float a) { return BF ? (float)(__bf16)a : (float)(_Float16)a; }
template <bool BF>
__device__ __forceinline__ f32x16 tmfma(u32x4t a, u32x4t b, f32x16 c) {
    if (BF) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8t, a), __builtin_bit_cast(bf16x8t, b), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8t, a), __builtin_bit_cast(f16x8t, b), c, 0, 0, 0);
}
// registers 8 s .. 8 s + 7 of an accumulator tile as the B operand of k-step s of the next product
template <bool BF>
__device__ __forceinline__ u32x4t tpack8(const f32x16& a, int s, float m) {
    const int i = 8 * s;
    u32x4t r;
    r.x = tpack2<BF>(a[i] * m, a[i + 1] * m); r.y = tpack2<BF>(a[i + 2] * m, a[i + 3] * m);
    r.z = tpack2<BF>(a[i + 4] * m, a[i + 5] * m); r.w = tpack2<BF>(a[i + 6] * m, a[i + 7] * m);
    return r;
}

struct TAGeom {
    int B, F, P, h, causal, round_out, has_rel;
    float eps;
    int nseq;
};

constexpr int TD = 64;          // dim_head

template <int C, int N, int NW = 4>
struct TACfg {
    static constexpr int TT = N / 32;                 // token tiles
    static constexpr int XROWB = C * 2 + 16;          // xn row: C 16-bit channels + pad
    static constexpr int KROWB = TD * 2 + 16;         // k row: 64 channels (permuted positions) + pad
    static constexpr int VROWB = N * 2 + 16;          // v^T row: N key positions (permuted) + pad
    static constexpr int PROW = 64 + 4;               // floats per row of a y partial (one 64-channel block)
    static constexpr int XN_OFF = 0;
    static constexpr int K_OFF = XN_OFF + N * XROWB;
    static constexpr int V_OFF = K_OFF + N * KROWB;
    static constexpr int REL_OFF = V_OFF + TD * VROWB;            // [h <= 8][2N] floats
    static constexpr int NULL_OFF = REL_OFF + 8 * 2 * N * 4;      // k_null[64], v_null[64] floats
    static constexpr int PART_OFF = NULL_OFF + 2 * TD * 4;        // [4 waves][N][PROW] floats
    static constexpr int LDS = PART_OFF + 4 * N * PROW * 4;
    static constexpr int TPR = 64 * NW / N;           // threads per row in the staging / final passes
    static constexpr int QPT = (C / 4) / TPR;         // float4 quads per thread
    static constexpr bool WREG = C == 64;             // weights of a wave's heads in registers
    static_assert(N == 32 || N == 64, "32 or 64 frames");
    static_assert(C % 64 == 0 && C <= 256, "channels");
    static_assert((C / 4) % TPR == 0, "whole float4 quads per thread");
};

// NW waves, HPW heads per wave (heads = NW * HPW): 4 x 2 is one wave per SIMD with both heads' weights in registers; 8 x 1 puts two
// waves on a SIMD, so one wave's soft-max (vector ALU) runs under the other's MFMAs -- and with half the weights per wave the
// accumulators stay in arch registers (the 4 x 2 build spends 550 v_accvgpr_read per sequence moving them)
template <int C, int N, bool BF, int HPW, int NW>
__global__ __launch_bounds__(64 * NW, NW / 4) void temporal_attn_h_kernel(const float* __restrict__ x, const float* __restrict__ g1,
                                                                 const unsigned short* __restrict__ wq, const unsigned short* __restrict__ wkv,
                                                                 const unsigned short* __restrict__ wo, const float* __restrict__ g2,
                                                                 const float* __restrict__ nullkv, const float* __restrict__ rel,
                                                                 const float* __restrict__ null_bias, float* __restrict__ y, TAGeom g) {
    using Cf = TACfg<C, N, NW>;
    constexpr int NT = 64 * NW;
    constexpr int TT = Cf::TT, KB = C / 16, CT = C / 32, CH = C / 64, QPT = Cf::QPT, TPR = Cf::TPR;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_ta[];
    unsigned char* xnS = smem_ta + Cf::XN_OFF;
    unsigned char* kS = smem_ta + Cf::K_OFF;
    unsigned char* vS = smem_ta + Cf::V_OFF;
    float* relS = reinterpret_cast<float*>(smem_ta + Cf::REL_OFF);
    float* nullS = reinterpret_cast<float*>(smem_ta + Cf::NULL_OFF);
    float* partS = reinterpret_cast<float*>(smem_ta + Cf::PART_OFF);

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hf = lane >> 5;
    const int h = g.h;                 // = NW * HPW: wave w owns heads w, w + NW
    const int row = tid / TPR, part = tid % TPR;

    // ---- once per workgroup: bias tables, null key / value, gains ----
    for (int e = tid; e < h * 2 * N; e += NT) {
        const int hd = e / (2 * N), idx = e % (2 * N);
        // entry idx = query - key + N - 1: keys behind the query (idx < N - 1) are masked when causal
        const float b = (g.has_rel && idx < 2 * N - 1) ? rel[(size_t)idx * h + hd] : 0.f;
        relS[e] = (g.causal && idx < N - 1) ? -INFINITY : b;
    }
    // the null key / value as MFMA operands (constant for the whole kernel): A rows of "K": row 0 = k_null in the channel order of the
    // k image, the other 31 rows zero; A rows of "V^T": column 0 = v_null
    u32x4t knA[4], vnA[2];
    {
        const u32x4t z = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) {
            // position p = 16 k4 + 8 hf + 4 a + b holds channel 16 k4 + 8 a + 4 hf + b
            const float* kn = nullkv + 16 * k4 + 4 * hf;
            u32x4t v;
            v.x = tpack2<BF>(kn[0], kn[1]); v.y = tpack2<BF>(kn[2], kn[3]); v.z = tpack2<BF>(kn[8], kn[9]); v.w = tpack2<BF>(kn[10], kn[11]);
            knA[k4] = l31 == 0 ? v : z;
        }
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            u32x4t v = z;
            v.x = tpack2<BF>(nullkv[TD + 32 * dt + l31], 0.f);
            vnA[dt] = hf == 0 ? v : z;
        }
    }
    (void)nullS;
    constexpr bool LEAN = C > 128;        // wide rows: gains re-read at their use and no cross-sequence prefetch (registers)
    float4 g1v[LEAN ? 1 : QPT], g2v[LEAN ? 1 : QPT];
    if (!LEAN) {
#pragma unroll
        for (int u = 0; u < QPT; ++u) {
            g1v[u] = *reinterpret_cast<const float4*>(g1 + 4 * (part + TPR * u));
            g2v[u] = *reinterpret_cast<const float4*>(g2 + 4 * (part + TPR * u));
        }
    }
    auto gain1 = [&](int u) { return LEAN ? *reinterpret_cast<const float4*>(g1 + 4 * (part + TPR * u)) : g1v[LEAN ? 0 : u]; };
    auto gain2 = [&](int u) { return LEAN ? *reinterpret_cast<const float4*>(g2 + 4 * (part + TPR * u)) : g2v[LEAN ? 0 : u]; };

    // ---- weights of this wave's heads (registers when they fit; re-read from L2 per sequence otherwise) ----
    auto load_wq = [&](int hd, int dt, int kb) {
        return *reinterpret_cast<const u32x4t*>(wq + ((size_t)(hd * TD + 32 * dt + l31) * C + 16 * kb + 8 * hf));
    };
    auto load_wo = [&](int hd, int ct, int kb4) {
        return *reinterpret_cast<const u32x4t*>(wo + (((size_t)hd * C + 32 * ct + l31) * TD + 16 * kb4 + 8 * hf));
    };
    auto load_wkv = [&](int kind, int dt, int kb) {
        return *reinterpret_cast<const u32x4t*>(wkv + ((size_t)(kind * TD + 32 * dt + l31) * C + 16 * kb + 8 * hf));
    };
    constexpr int WR = Cf::WREG ? 1 : 0;
    u32x4t wqR[WR ? HPW : 1][2][WR ? KB : 1], woR[WR ? HPW : 1][WR ? CT : 1][4];
    // k / v projection tiles: t = (kind, tt, dt) in [0, 4 TT); wave w takes t = w, w + NW, ...
    constexpr int KVT = (4 * TT + NW - 1) / NW;
    u32x4t wkvR[KVT][WR ? KB : 1];
    if (WR) {
#pragma unroll
        for (int u = 0; u < HPW; ++u) {
            const int hd = wave + NW * u;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int kb = 0; kb < KB; ++kb) wqR[u][dt][kb] = load_wq(hd, dt, kb);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int k4 = 0; k4 < 4; ++k4) woR[u][ct][k4] = load_wo(hd, ct, k4);
        }
#pragma unroll
        for (int u = 0; u < KVT; ++u) {
            const int t = min(wave + NW * u, 4 * TT - 1);
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) wkvR[u][kb] = load_wkv(t / (2 * TT), t & 1, kb);
        }
    }
    __syncthreads();

    const size_t frameStride = (size_t)g.P * C;
    auto seq_base = [&](int s) { return ((size_t)(s / g.P) * g.F * g.P + (size_t)(s % g.P)) * C; };

    float4 xv[QPT], xnext[LEAN ? 1 : QPT];
    int seq = blockIdx.x;
    if (!LEAN && seq < g.nseq) {
        const float* xp = x + seq_base(seq) + (size_t)row * frameStride;
#pragma unroll
        for (int u = 0; u < QPT; ++u) xv[u] = *reinterpret_cast<const float4*>(xp + 4 * (part + TPR * u));
    }
    for (; seq < g.nseq; seq += gridDim.x) {
        const size_t base = seq_base(seq);
        if (LEAN) {
            const float* xp = x + base + (size_t)row * frameStride;
#pragma unroll
            for (int u = 0; u < QPT; ++u) xv[u] = *reinterpret_cast<const float4*>(xp + 4 * (part + TPR * u));
        }
        // ================= phase A: LayerNorm of the sequence rows -> xn (16-bit) in LDS =================
        {
            float s1 = 0.f;
#pragma unroll
            for (int u = 0; u < QPT; ++u) s1 += (xv[u].x + xv[u].y) + (xv[u].z + xv[u].w);
#pragma unroll
            for (int m = 1; m < TPR; m <<= 1) s1 += __shfl_xor(s1, m, 64);
            const float mean = s1 * (1.f / C);
            float s2 = 0.f;
#pragma unroll
            for (int u = 0; u < QPT; ++u) {
                const float a = xv[u].x - mean, b = xv[u].y - mean, c = xv[u].z - mean, d = xv[u].w - mean;
                s2 += (a * a + b * b) + (c * c + d * d);
            }
#pragma unroll
            for (int m = 1; m < TPR; m <<= 1) s2 += __shfl_xor(s2, m, 64);
            const float rstd = rsqrtf(s2 * (1.f / C) + g.eps);
#pragma unroll
            for (int u = 0; u < QPT; ++u) {
                u32x2t w;
                const float4 gg = gain1(u);
                w.x = tpack2<BF>((xv[u].x - mean) * rstd * gg.x, (xv[u].y - mean) * rstd * gg.y);
                w.y = tpack2<BF>((xv[u].z - mean) * rstd * gg.z, (xv[u].w - mean) * rstd * gg.w);
                *reinterpret_cast<u32x2t*>(xnS + row * Cf::XROWB + 8 * (part + TPR * u)) = w;
            }
        }
        // the next sequence's rows: in flight during the whole of phases B / C
        {
            const int ns = seq + gridDim.x;
            if (!LEAN && ns < g.nseq) {
                const float* xp = x + seq_base(ns) + (size_t)row * frameStride;
#pragma unroll
                for (int u = 0; u < QPT; ++u) xnext[u] = *reinterpret_cast<const float4*>(xp + 4 * (part + TPR * u));
            }
        }
        __syncthreads();
        // ================= phase B: k (rows in LDS, channel positions permuted) and v^T (key positions permuted) =================
#pragma unroll
        for (int u = 0; u < KVT; ++u) {
            const int t = wave + NW * u;
            if (t >= 4 * TT) continue;                                 // wave-uniform (8 waves, 32 frames: waves 4 .. 7 have no tile)
            const int kind = t / (2 * TT), kvDt = t & 1, kvTt = (t % (2 * TT)) >> 1;       // 0: k (orientation W x xn^T), 1: v (xn x W^T)
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
            const unsigned char* xr = xnS + (32 * kvTt + l31) * Cf::XROWB + 16 * hf;
            if (kind == 0) {
#pragma unroll
                for (int kb = 0; kb < KB; ++kb) {
                    const u32x4t xf = *reinterpret_cast<const u32x4t*>(xr + 32 * kb);
                    acc = tmfma<BF>(WR ? wkvR[u][WR ? kb : 0] : load_wkv(0, kvDt, kb), xf, acc);
                }
            } else {
#pragma unroll
                for (int kb = 0; kb < KB; ++kb) {
                    const u32x4t xf = *reinterpret_cast<const u32x4t*>(xr + 32 * kb);
                    acc = tmfma<BF>(xf, WR ? wkvR[u][WR ? kb : 0] : load_wkv(1, kvDt, kb), acc);
                }
            }
            if (kind == 0) {        // acc: rows = channels 32 dt + (4 hf + 8 g4 + e), column = token 32 tt + l31
                unsigned char* kr = kS + (32 * kvTt + l31) * Cf::KROWB;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int pos = 32 * kvDt + 16 * (g4 >> 1) + 8 * hf + 4 * (g4 & 1);
                    u32x2t w;
                    w.x = tpack2<BF>(acc[4 * g4], acc[4 * g4 + 1]); w.y = tpack2<BF>(acc[4 * g4 + 2], acc[4 * g4 + 3]);
                    *reinterpret_cast<u32x2t*>(kr + 2 * pos) = w;
                }
            } else {                // acc: rows = tokens 32 tt + (4 hf + 8 g4 + e), column = channel 32 dt + l31
                unsigned char* vr = vS + (32 * kvDt + l31) * Cf::VROWB;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int pos = 32 * kvTt + 16 * (g4 >> 1) + 8 * hf + 4 * (g4 & 1);
                    u32x2t w;
                    w.x = tpack2<BF>(acc[4 * g4], acc[4 * g4 + 1]); w.y = tpack2<BF>(acc[4 * g4 + 2], acc[4 * g4 + 3]);
                    *reinterpret_cast<u32x2t*>(vr + 2 * pos) = w;
                }
            }
        }
        __syncthreads();
        // ================= phase C: the heads of this wave =================
        u32x4t oB[HPW][TT][4];           // the heads' outputs o^T as B operands of the to_out product
#pragma unroll
        for (int u = 0; u < HPW; ++u) {
            {
                const int hd = wave + NW * u;
                if (!WR) __builtin_amdgcn_sched_barrier(0);          // weights come from L2: keep one head's loads in flight, not both
                // ---- q^T = W_q xn^T (softmax scale folded into W_q) ----
                f32x16 qacc[2][TT];
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int tt = 0; tt < TT; ++tt)
#pragma unroll
                        for (int i = 0; i < 16; ++i) qacc[dt][tt][i] = 0.f;
#pragma unroll(WR ? KB : 4)
                for (int kb = 0; kb < KB; ++kb) {
                    u32x4t xf[TT];
#pragma unroll
                    for (int tt = 0; tt < TT; ++tt)
                        xf[tt] = *reinterpret_cast<const u32x4t*>(xnS + (32 * tt + l31) * Cf::XROWB + 16 * hf + 32 * kb);
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) {
                        const u32x4t wf = WR ? wqR[WR ? u : 0][dt][WR ? kb : 0] : load_wq(hd, dt, kb);
#pragma unroll
                        for (int tt = 0; tt < TT; ++tt) qacc[dt][tt] = tmfma<BF>(wf, xf[tt], qacc[dt][tt]);
                    }
                }
                u32x4t qB[TT][4];
                float snull[TT];
                const float nb = (g.has_rel && null_bias) ? null_bias[hd] : 0.f;
#pragma unroll
                for (int tt = 0; tt < TT; ++tt) {
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                        for (int s = 0; s < 2; ++s) qB[tt][2 * dt + s] = tpack8<BF>(qacc[dt][tt], s, 1.f);
                    // the null key's score: row 0 of (k_null | 0 ...) q^T
                    f32x16 na;
#pragma unroll
                    for (int i = 0; i < 16; ++i) na[i] = 0.f;
#pragma unroll
                    for (int k4 = 0; k4 < 4; ++k4) na = tmfma<BF>(knA[k4], qB[tt][k4], na);
                    const float own = na[0], other = __shfl_xor(own, 32, 64);
                    snull[tt] = (hf ? other : own) + nb;
                }
                // ---- S^T = k q^T + bias (the accumulator starts at the relative-position bias / the causal mask) ----
                f32x16 sacc[TT][TT];       // [key tile][query tile]
                const float* relH = relS + hd * 2 * N;
#pragma unroll
                for (int kt = 0; kt < TT; ++kt)
#pragma unroll
                    for (int tt = 0; tt < TT; ++tt)
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            const int j = 32 * kt + (i & 3) + 8 * (i >> 2) + 4 * hf, qi = 32 * tt + l31;
                            sacc[kt][tt][i] = relH[qi - j + N - 1];
                        }
#pragma unroll
                for (int kt = 0; kt < TT; ++kt)
#pragma unroll
                    for (int k4 = 0; k4 < 4; ++k4) {
                        const u32x4t kf = *reinterpret_cast<const u32x4t*>(kS + (32 * kt + l31) * Cf::KROWB + 32 * k4 + 16 * hf);
#pragma unroll
                        for (int tt = 0; tt < TT; ++tt) sacc[kt][tt] = tmfma<BF>(kf, qB[tt][k4], sacc[kt][tt]);
                    }
                // ---- soft-max over the keys of a query: registers of a lane + the other half-wave + the null key ----
                u32x4t pB[TT][TT][2];      // [query tile][key tile][k-step]
                float pnull[TT], inv[TT];
#pragma unroll
                for (int tt = 0; tt < TT; ++tt) {
                    float m = -INFINITY;
#pragma unroll
                    for (int kt = 0; kt < TT; ++kt)
#pragma unroll
                        for (int i = 0; i < 16; ++i) m = fmaxf(m, sacc[kt][tt][i]);
                    m = fmaxf(m, __shfl_xor(m, 32, 64));
                    m = fmaxf(m, snull[tt]);
                    float l = 0.f;
                    constexpr float LOG2E = 1.4426950408889634f;
                    const float mneg = -m * LOG2E;              // exp(s - m) = 2^(s log2e - m log2e): one fma + v_exp_f32 per score
#pragma unroll
                    for (int kt = 0; kt < TT; ++kt)
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            const float p = __builtin_amdgcn_exp2f(fmaf(sacc[kt][tt][i], LOG2E, mneg));
                            sacc[kt][tt][i] = p;
                            l += p;
                        }
                    l += __shfl_xor(l, 32, 64);
                    pnull[tt] = __builtin_amdgcn_exp2f(fmaf(snull[tt], LOG2E, mneg));
                    l += pnull[tt];
                    inv[tt] = 1.f / l;
#pragma unroll
                    for (int kt = 0; kt < TT; ++kt)
#pragma unroll
                        for (int s = 0; s < 2; ++s) pB[tt][kt][s] = tpack8<BF>(sacc[kt][tt], s, 1.f);
                }
                // ---- o^T = v^T P^T (+ the null value), normalised ----
                f32x16 oacc[2][TT];
#pragma unroll
                for (int tt = 0; tt < TT; ++tt) {
                    u32x4t pn = {0u, 0u, 0u, 0u};
                    pn.x = hf == 0 ? tpack2<BF>(pnull[tt], 0.f) : 0u;            // B[k = 0][query] = p_null, the rest of the k-step zero
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) {
#pragma unroll
                        for (int i = 0; i < 16; ++i) oacc[dt][tt][i] = 0.f;
                        oacc[dt][tt] = tmfma<BF>(vnA[dt], pn, oacc[dt][tt]);
                    }
                }
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int kt = 0; kt < TT; ++kt)
#pragma unroll
                        for (int s = 0; s < 2; ++s) {
                            const u32x4t vf = *reinterpret_cast<const u32x4t*>(vS + (32 * dt + l31) * Cf::VROWB + 2 * (32 * kt + 16 * s + 8 * hf));
#pragma unroll
                            for (int tt = 0; tt < TT; ++tt) oacc[dt][tt] = tmfma<BF>(vf, pB[tt][kt][s], oacc[dt][tt]);
                        }
#pragma unroll
                for (int tt = 0; tt < TT; ++tt)
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                        for (int s = 0; s < 2; ++s) oB[u][tt][2 * dt + s] = tpack8<BF>(oacc[dt][tt], s, inv[tt]);
            }
        }
        // ================= phase D: sum the waves' partials (fixed order), LayerNorm, residual, store =================
        float4 yv[QPT];
#pragma unroll(LEAN ? 1 : CH)
        for (int cb = 0; cb < CH; ++cb) {
            // ---- y^T (64 channels) = sum over this wave's heads of W_o^T o^T ----
            f32x16 yacc[2][TT];
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
                for (int tt = 0; tt < TT; ++tt)
#pragma unroll
                    for (int i = 0; i < 16; ++i) yacc[c2][tt][i] = 0.f;
#pragma unroll
            for (int u = 0; u < HPW; ++u) {
                {
                    const int hd = wave + NW * u;
#pragma unroll
                    for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
                        for (int k4 = 0; k4 < 4; ++k4) {
                            const u32x4t wf = WR ? woR[WR ? u : 0][WR ? 2 * cb + c2 : 0][k4] : load_wo(hd, 2 * cb + c2, k4);
#pragma unroll
                            for (int tt = 0; tt < TT; ++tt) yacc[c2][tt] = tmfma<BF>(wf, oB[u][tt][k4], yacc[c2][tt]);
                        }
                }
            }
            if (cb) __syncthreads();                     // the previous block's partials have been read
            // 4 slots of [N][PROW]: with 8 waves, waves 4 .. 7 write theirs first, waves 0 .. 3 add those to their own and write the sums
            auto slot_rw = [&](int slot, bool add, bool store) {
#pragma unroll
                for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
                    for (int tt = 0; tt < TT; ++tt) {
                        float* pr = partS + ((size_t)slot * N + 32 * tt + l31) * Cf::PROW + 32 * c2 + 4 * hf;
#pragma unroll
                        for (int g4 = 0; g4 < 4; ++g4) {
                            f32x16& a = yacc[c2][tt];
                            if (add) {
                                const float4 t = *reinterpret_cast<const float4*>(pr + 8 * g4);
                                a[4 * g4] += t.x; a[4 * g4 + 1] += t.y; a[4 * g4 + 2] += t.z; a[4 * g4 + 3] += t.w;
                            }
                            if (store) *reinterpret_cast<float4*>(pr + 8 * g4) = make_float4(a[4 * g4], a[4 * g4 + 1], a[4 * g4 + 2], a[4 * g4 + 3]);
                        }
                    }
            };
            if (NW == 8) {
                if (wave >= 4) slot_rw(wave - 4, false, true);
                __syncthreads();
                if (wave < 4) slot_rw(wave, true, true);
            } else {
                slot_rw(wave, false, true);
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < QPT; ++u) {
                const int quad = part + TPR * u;             // quads 16 cb .. 16 cb + 15 belong to this block
                if (quad / 16 == cb) {
                    const float* pr = partS + (size_t)row * Cf::PROW + 4 * (quad % 16);
                    float4 s = *reinterpret_cast<const float4*>(pr);
#pragma unroll
                    for (int w = 1; w < 4; ++w) {
                        const float4 t = *reinterpret_cast<const float4*>(pr + (size_t)w * N * Cf::PROW);
                        s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
                    }
                    yv[u] = s;
                }
            }
        }
        {
            if (g.round_out) {
#pragma unroll
                for (int u = 0; u < QPT; ++u) {
                    yv[u].x = tround<BF>(yv[u].x); yv[u].y = tround<BF>(yv[u].y); yv[u].z = tround<BF>(yv[u].z); yv[u].w = tround<BF>(yv[u].w);
                }
            }
            float s1 = 0.f;
#pragma unroll
            for (int u = 0; u < QPT; ++u) s1 += (yv[u].x + yv[u].y) + (yv[u].z + yv[u].w);
#pragma unroll
            for (int m = 1; m < TPR; m <<= 1) s1 += __shfl_xor(s1, m, 64);
            const float mean = s1 * (1.f / C);
            float s2 = 0.f;
#pragma unroll
            for (int u = 0; u < QPT; ++u) {
                const float a = yv[u].x - mean, b = yv[u].y - mean, c = yv[u].z - mean, d = yv[u].w - mean;
                s2 += (a * a + b * b) + (c * c + d * d);
            }
#pragma unroll
            for (int m = 1; m < TPR; m <<= 1) s2 += __shfl_xor(s2, m, 64);
            const float rstd = rsqrtf(s2 * (1.f / C) + g.eps);
            float* yp = y + base + (size_t)row * frameStride;
#pragma unroll
            for (int u = 0; u < QPT; ++u) {
                float4 o;
                const float4 gg = gain2(u);
                if (LEAN) xv[u] = *reinterpret_cast<const float4*>(x + base + (size_t)row * frameStride + 4 * (part + TPR * u));
                o.x = (yv[u].x - mean) * rstd * gg.x + xv[u].x; o.y = (yv[u].y - mean) * rstd * gg.y + xv[u].y;
                o.z = (yv[u].z - mean) * rstd * gg.z + xv[u].z; o.w = (yv[u].w - mean) * rstd * gg.w + xv[u].w;
                *reinterpret_cast<float4*>(yp + 4 * (part + TPR * u)) = o;
            }
        }
        if (!LEAN) {
#pragma unroll
            for (int u = 0; u < QPT; ++u) xv[u] = xnext[LEAN ? 0 : u];
        }
        __syncthreads();                                 // xn / k / v^T / partials are rewritten by the next sequence
    }
}

template <int C, int N>
int ta_launch(const float* x, const float* g1, const void* wq, const void* wkv, const void* wo, const float* g2, const float* nullkv,
              const float* rel, const float* null_bias, float* y, const TAGeom& g, int bf16, hipStream_t s) {
    // 8 heads: one head per wave, 8 waves (two per SIMD); DIQT_TATTN_W4=1: the first build, 4 waves x 2 heads.  4 heads: 4 waves x 1 head.
    static const bool w4 = [] { const char* e = getenv("DIQT_TATTN_W4"); return e && e[0] == '1'; }();
    static const int wgs = [] { const char* e = getenv("DIQT_TATTN_WGS"); return e ? atoi(e) : 256; }();
    const int grid = g.nseq < wgs ? g.nseq : wgs;
    auto go = [&](auto kern, int threads, int lds) -> int {
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            DIQT_REQUIRE(e == hipSuccess, DIQT_E_LAUNCH, "temporal_attention_h: hipFuncSetAttribute: %s", hipGetErrorString(e));
        }
        hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, s, x, g1, static_cast<const unsigned short*>(wq),
                           static_cast<const unsigned short*>(wkv), static_cast<const unsigned short*>(wo), g2, nullkv, rel, null_bias, y, g);
        return check_launch("temporal_attention_h");
    };
    static_assert(TACfg<C, N, 4>::LDS <= 160 * 1024 && TACfg<C, N, 8>::LDS <= 160 * 1024, "LDS");
    // (C = 128, weights re-read from L2: the 4-wave build measures faster, 604 vs 641 us on 8192 sequences of 64 frames)
    if (g.h == 8 && !w4 && C != 128)
        return bf16 ? go(temporal_attn_h_kernel<C, N, true, 1, 8>, 512, TACfg<C, N, 8>::LDS) : go(temporal_attn_h_kernel<C, N, false, 1, 8>, 512, TACfg<C, N, 8>::LDS);
    if (g.h == 8)
        return bf16 ? go(temporal_attn_h_kernel<C, N, true, 2, 4>, 256, TACfg<C, N, 4>::LDS) : go(temporal_attn_h_kernel<C, N, false, 2, 4>, 256, TACfg<C, N, 4>::LDS);
    return bf16 ? go(temporal_attn_h_kernel<C, N, true, 1, 4>, 256, TACfg<C, N, 4>::LDS) : go(temporal_attn_h_kernel<C, N, false, 1, 4>, 256, TACfg<C, N, 4>::LDS);
}


}  // namespace
}  // namespace diqt

using namespace diqt;

extern "C" int diqt_temporal_attention_h_supported(int B, int F, int P, int C, int h, int d) {
    static const bool off = [] { const char* e = getenv("DIQT_NO_TATTN"); return e && e[0] == '1'; }();
    if (off || d != TD || (h != 4 && h != 8) || (F != 32 && F != 64) || (C != 64 && C != 128 && C != 256)) return 0;
    if (B < 1 || P < 1 || (long long)B * P >= (1ll << 30)) return 0;
    return 1;
}

extern "C" int diqt_temporal_attention_h(const float* x, const float* norm_g, const void* wq_h, const void* wkv_h, const void* wo_h,
                                         const float* out_g, const float* null_kv, const float* rel, const float* null_bias, float* y,
                                         int B, int F, int P, int C, int h, int d, int causal, float eps, int bf16, int round_out,
                                         void* stream) {
    DIQT_REQUIRE(diqt_temporal_attention_h_supported(B, F, P, C, h, d), DIQT_E_UNSUPPORTED,
                 "temporal_attention_h: unsupported shape B=%d F=%d P=%d C=%d h=%d d=%d", B, F, P, C, h, d);
    DIQT_REQUIRE(x && norm_g && wq_h && wkv_h && wo_h && out_g && null_kv && y, DIQT_E_ALIGN, "temporal_attention_h: null pointer");
    DIQT_REQUIRE(aligned16(x) && aligned16(y) && aligned16(wq_h) && aligned16(wkv_h) && aligned16(wo_h) && aligned16(norm_g) && aligned16(out_g),
                 DIQT_E_ALIGN, "temporal_attention_h: pointers must be 16-byte aligned");
    TAGeom g;
    g.B = B; g.F = F; g.P = P; g.h = h; g.causal = causal; g.round_out = round_out; g.has_rel = rel != nullptr; g.eps = eps;
    g.nseq = B * P;
    hipStream_t s = (hipStream_t)stream;
    if (C == 64) {
        return F == 64 ? ta_launch<64, 64>(x, norm_g, wq_h, wkv_h, wo_h, out_g, null_kv, rel, null_bias, y, g, bf16, s)
                       : ta_launch<64, 32>(x, norm_g, wq_h, wkv_h, wo_h, out_g, null_kv, rel, null_bias, y, g, bf16, s);
    }
    if (C == 128) {
        return F == 64 ? ta_launch<128, 64>(x, norm_g, wq_h, wkv_h, wo_h, out_g, null_kv, rel, null_bias, y, g, bf16, s)
                       : ta_launch<128, 32>(x, norm_g, wq_h, wkv_h, wo_h, out_g, null_kv, rel, null_bias, y, g, bf16, s);
    }
    return F == 64 ? ta_launch<256, 64>(x, norm_g, wq_h, wkv_h, wo_h, out_g, null_kv, rel, null_bias, y, g, bf16, s)
                   : ta_launch<256, 32>(x, norm_g, wq_h, wkv_h, wo_h, out_g, null_kv, rel, null_bias, y, g, bf16, s);
}
