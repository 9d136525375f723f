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
#ifndef DIQT_DQ_OCC
// mqa_flash_bwd_dq_kernel: workgroups per CU.  At 2 the d = 64 build is held to 256 registers per wave and spills 92 B per lane; at 1 it
// takes 255 + 52 registers and nothing spills -- and runs SLOWER: 1139 vs 935 us per call of the joint 2048-token attention backward of
// Unet3D dim 64 @ 32^3 (rocprofv3 kernel trace, same box, round 4): the second workgroup's latency hiding is worth more than the
// spilled values cost (they are re-read outside the key-tile loop).  Measured, kept at 2.
#define DIQT_DQ_OCC 2
#endif

namespace diqt {

typedef float f32x4a __attribute__((ext_vector_type(4)));       // LDS operand quads (plain vector registers)
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int AQ = 128;     // query rows per workgroup (4 waves x 32)
constexpr int AKT = 32;     // keys per tile

// score of (query row, key j) after the additive terms and masks of Attention.forward (imagen_video.py:490-518): T5-style relative
// bias on the self keys, the null-key bias on the last extra key, -inf for causal-masked and out-of-range keys.  Branch-free on
// purpose (clamped unconditional gather + selects): a guarded form compiles to four exec-mask branches per element, each gather
// behind its own s_waitcnt vmcnt(0) -- which also drains the next tile's prefetch.
// The two halves of attn_bias_mask for callers that apply it to the 16 accumulator rows of a tile: FIRST all 16 table gathers (one
// kernel-uniform `if (rel)` around the loop, the loads back to back), THEN the branch-free combine.  Calling attn_bias_mask per element
// compiles to a branch + global load + `s_waitcnt vmcnt(0)` per element: 16 serialised L2 round trips per tile (~10k cycles beside
// 8k cycles of MFMA on the temporal attentions), and each wait also drains the next tile's prefetch.
__device__ __forceinline__ int attn_rel_index(int j, int E, int ns, int h, int qi, int qh) {
    return max(0, min(qi - (j - E) + ns - 1, 2 * ns - 2)) * h + qh;
}
__device__ __forceinline__ float attn_bias_apply(float v, int j, int M, int E, int qi, int causal, bool hasRel, float rv, float nbv) {
    const int jj = j - E;
    const bool self = j >= E;
    float b = (j == E - 1) ? nbv : 0.f;
    b = (hasRel && self) ? rv : b;
    const bool dead = j >= M || (causal && self && jj > qi);
    return dead ? -INFINITY : v + b;
}

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

// HASREL: a relative-position bias table is given -- a template parameter, see mqa_flash_fwd_h_kernel (the table-index arithmetic of a
// tile is hoisted above the `plain tile` branch: vector instructions that share the f32 MFMA's lanes, spent even without a table)
template <int ND, bool HASREL = true>           // dim_head = 32 * ND
__global__ __launch_bounds__(256, 2) void mqa_flash_fwd_kernel(const float* __restrict__ q, const float* __restrict__ kv,
                                                               const float* __restrict__ rel_, const float* __restrict__ null_bias,
                                                               float* __restrict__ out, float* __restrict__ lse, int n, int h, int E,
                                                               int ns, int causal, float scale, int P, const float* __restrict__ nullkv) {
    // P > 0: the sequences are the FRAME axis of channels-last tensors q[B][n][P][h D], kv[B][n][P][2 D], out like q (sequence
    // g = (b, pixel p), token i = frame: row (b n + i) P + p), read and written in place -- no transposed copies around the temporal
    // attentions -- and the lone extra key / value (E == 1, the learned null row) comes from nullkv[2 D] instead of a concatenated copy
    // of kv.  P == 0: q[G][n][h D], kv[G][E + ns][2 D] as documented above.
    constexpr int D = 32 * ND, ROW = D + 4, NPF = AKT * (2 * D / 4) / 256;    // float4 pieces of a K|V tile per thread
    const float* __restrict__ rel = HASREL ? rel_ : nullptr;
    __shared__ __attribute__((aligned(16))) float KVs[2][2][AKT * ROW];       // [buffer][K | V][key][ROW]
    const int g = blockIdx.y;
    const int M = E + ns, R = n * h;                       // keys, query rows of this batch entry
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hf = lane >> 5;
    const int r = blockIdx.x * AQ + wave * 32 + l31;       // this lane's query row (column of S^T)
    const bool rvalid = r < R;
    const int rc = rvalid ? r : R - 1;
    const int qi = rc / h, qh = rc % h;                    // token index, head
    const size_t tokStride = P > 0 ? (size_t)P : 1;        // rows between consecutive tokens of a sequence
    const size_t seqRow0 = P > 0 ? (size_t)(g / P) * n * P + (size_t)(g % P) : (size_t)g * n;      // row of token 0 in q / out
    const float* qg = q + ((seqRow0 + (size_t)qi * tokStride) * h + qh) * D;
    // kv rows: key j of the sequence (P == 0: row j of its [E + ns] block; P > 0: the null row for j == 0, frame j - 1 otherwise)
    const float* kvg = P > 0 ? nullkv : kv + (size_t)g * M * 2 * D;
    const float* kvSelf = P > 0 ? kv + ((size_t)(g / P) * ns * P + (size_t)(g % P)) * 2 * D : nullptr;

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
    // A lone extra key (the learned null key: E == 1, every IQT call) is taken by the VALU instead of opening a 32-key MFMA tile
    // for one row: the temporal attentions (n = 32 frames: 33 keys) halve their matrix work, the 8x8 spatial ones go from 3 tiles to 2.
    const int kbase = (E == 1) ? 1 : 0;                    // kernel-uniform: first key handled by the tile loop
    if (kbase) {
        float s0 = 0.f;
#pragma unroll
        for (int gq = 0; gq < D / 8; ++gq) {
            const float4 kk = *reinterpret_cast<const float4*>(kvg + 8 * gq + 4 * hf);
            s0 = fmaf(qreg[4 * gq], kk.x, s0); s0 = fmaf(qreg[4 * gq + 1], kk.y, s0);
            s0 = fmaf(qreg[4 * gq + 2], kk.z, s0); s0 = fmaf(qreg[4 * gq + 3], kk.w, s0);
        }
        s0 += __shfl_xor(s0, 32, 64);
        mrun = s0 + nbv;                                   // its probability against the running maximum is exp(0) = 1
        lrun = 1.f;
#pragma unroll
        for (int c = 0; c < ND; ++c)
#pragma unroll
            for (int i = 0; i < 16; ++i) o[c][i] = kvg[D + 32 * c + (i & 3) + 8 * (i >> 2) + 4 * hf];
    }

    // K | V tile of keys [kbase + 32t, kbase + 32t + 32): kv row = [k(D) | v(D)].  The next tile is loaded global -> registers BEFORE this tile's
    // MFMAs and written to the other LDS buffer after them: one barrier per tile and no memory round trip between tiles.
    // Loads are unconditional (clamped key, zero-selected): a guarded load becomes an exec-masked branch with its own wait.
    float4 pre[NPF];
    auto load_tile = [&](int t) {
#pragma unroll
        for (int u = 0; u < NPF; ++u) {
            const int e = u * 256 + tid;
            const int key = e / (2 * D / 4), c4 = (e % (2 * D / 4)) * 4;
            const int j = kbase + t * AKT + key;
            const int jc = min(j, M - 1);
            const float* krow = P > 0 ? kvSelf + (size_t)(jc - 1) * tokStride * 2 * D : kvg + (size_t)jc * 2 * D;     // P > 0: kbase == 1, jc >= 1
            const float4 v = *reinterpret_cast<const float4*>(krow + c4);
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
    const int ntiles = (M - kbase + AKT - 1) / AKT;
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
        // operand reads run one group ahead of their MFMAs (left alone hipcc issues each read right in front of its first use and
        // waits: ~130 cycles of LDS latency per 4 MFMAs here, per 2 MFMAs in the P V loop below)
        f32x4a ka = *reinterpret_cast<const f32x4a*>(kp);
#pragma unroll
        for (int gq = 0; gq < D / 8; ++gq) {
            f32x4a kn = ka;
            if (gq + 1 < D / 8) kn = *reinterpret_cast<const f32x4a*>(kp + 8 * (gq + 1));
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[0], qreg[4 * gq], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[1], qreg[4 * gq + 1], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[2], qreg[4 * gq + 2], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[3], qreg[4 * gq + 3], s, 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            ka = kn;
        }
        // ---- bias, mask, online softmax (per lane = per query; rows of s are keys) ----
        float tmax = -INFINITY;
        const bool plain = !causal && !rel && kbase + t * AKT >= E && kbase + (t + 1) * AKT <= M;     // every key of the tile is an unbiased self key
        if (!plain) {
            float rv[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) rv[i] = 0.f;
            if (rel) {                                     // kernel-uniform: 16 gathers in flight together
#pragma unroll
                for (int i = 0; i < 16; ++i) rv[i] = rel[attn_rel_index(kbase + t * AKT + (i & 3) + 8 * (i >> 2) + 4 * hf, E, ns, h, qi, qh)];
            }
#pragma unroll
            for (int i = 0; i < 16; ++i)
                s[i] = attn_bias_apply(s[i], kbase + t * AKT + (i & 3) + 8 * (i >> 2) + 4 * hf, M, E, qi, causal, rel != nullptr, rv[i], nbv);
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
        {
            float vc[ND], vn[ND];
#pragma unroll
            for (int c = 0; c < ND; ++c) vc[c] = Vs[(4 * hf) * ROW + l31 + 32 * c];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (i + 1 < 16) {
                    const float* vp = Vs + (((i + 1) & 3) + 8 * ((i + 1) >> 2) + 4 * hf) * ROW + l31;
#pragma unroll
                    for (int c = 0; c < ND; ++c) vn[c] = vp[32 * c];
                }
#pragma unroll
                for (int c = 0; c < ND; ++c) o[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(vc[c], s[i], o[c], 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, ND, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, ND, 0);
#pragma unroll
                for (int c = 0; c < ND; ++c) vc[c] = vn[c];
            }
        }
        if (t + 1 < ntiles) store_tile((t + 1) & 1);       // that buffer was last read in tile t-1, retired by its barrier
        __syncthreads();
    }
    // ---- epilogue: O^T rows are head-dim indices, columns are queries: out[g, r, dd] = o / l ----
    if (rvalid) {
        const float inv = 1.f / lrun;
        if (lse && hf == 0) lse[(size_t)g * R + r] = mrun + __logf(lrun);      // log-sum-exp of the row: what the backward recomputes P from
        float* og = out + ((seqRow0 + (size_t)qi * tokStride) * h + qh) * D;
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

// HASREL: a relative-position bias table is given.  A template parameter, not a test of the pointer: hipcc hoists the 32 clamped table
// indices of a tile (min / max / 64-bit multiply-add each) above the `plain tile` branch, where they cost ~190 vector instructions per
// tile and wave even when there is no table (the joint space-time attention at the middle of the U-Net: 257 tiles per query block)
// NWV waves per workgroup: 8 (256 query rows; K|V streamed once per 256 rows) or 4 with TWO workgroups per CU -- the two waves of a SIMD then
// belong to different workgroups and are not held in the same phase by the tile barrier, so one's soft-max can run under the other's MFMAs
template <int ND, bool BF, bool HASREL, int NWV = 8>   // dim_head = 32 * ND
__global__ __launch_bounds__(64 * NWV, NWV == 4 ? 2 : 1) void mqa_flash_fwd_h_kernel(const float* __restrict__ q, const unsigned short* __restrict__ kv,
                                                                 const float* __restrict__ rel_, const float* __restrict__ null_bias,
                                                                 float* __restrict__ out, int n, int h, int E, int ns, int causal,
                                                                 float scale, int round_out) {
    constexpr int NTH = 64 * NWV, AQW = 32 * NWV;
    constexpr int D = 32 * ND, KROWB = 2 * D + 16, VROWB = 2 * HKT + 16, NPF = HKT * (2 * D / 8) / NTH;   // 16-byte pieces per thread
    const float* __restrict__ rel = HASREL ? rel_ : nullptr;
    static_assert(NPF >= 1, "tile too small for the workgroup");
    __shared__ __attribute__((aligned(16))) unsigned char Ksm[2][HKT * KROWB];    // [buffer][key][d] 16-bit, padded rows
    // V ROW-MAJOR like K ([buffer][key][d]); the V^T operand of the second product comes out of `ds_read_b64_tr_b16`, which hands lane i
    // column i of a 4-key x 16-channel block.  (The first version stored V transposed with eight 2-byte LDS writes per piece: rows 8
    // channels apart are 1152 B apart, i.e. TWO banks for a whole wave -- 1.9e9 bank-conflict cycles per launch of the 16k-token
    // attention, a third of its run time.)
    // rows 192 B apart: the 4 rows x 2 channel blocks a 32-lane half reads at once then fall on 8 disjoint 8-bank groups (48 q + 8 blk)
    constexpr int VRB = 192;
    __shared__ __attribute__((aligned(16))) unsigned char Vsm[2][HKT * VRB];
    const int g = blockIdx.y;
    const int M = E + ns, R = n * h;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hf = lane >> 5;
    const int r = blockIdx.x * AQW + wave * 32 + l31;
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
            const int e = u * NTH + tid;
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
            const int e = u * NTH + tid;
            const int key = e / (2 * D / 8), c8 = (e % (2 * D / 8)) * 8;
            if (c8 < D) *reinterpret_cast<u32x4a*>(&Ksm[buf][key * KROWB + c8 * 2]) = pre[u];
            else *reinterpret_cast<u32x4a*>(&Vsm[buf][key * VRB + (c8 - D) * 2]) = pre[u];
        }
    };
    // V^T fragment of k-step sx (16 keys) for channel tile c: element j of half hf is key 16 sx + (j < 4 ? 4 hf + j : 8 + 4 hf + j - 4),
    // the order the P^T registers present the keys in; a 16-lane group reads 4 keys x 16 channels per transposing read
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const int trq = (lane & 15) >> 2, trp = lane & 3, trd = (lane >> 4) & 1;
    auto v_frag = [&](const unsigned char* Vb, int c, int sx) {
        const unsigned char* a0 = Vb + (16 * sx + 4 * hf + trq) * VRB + (32 * c + 16 * trd + 4 * trp) * 2;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a0);
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0 + 8 * VRB));
        u32x4a r;
        const unsigned long long l = __builtin_bit_cast(unsigned long long, lo), h2 = __builtin_bit_cast(unsigned long long, hi);
        r.x = (unsigned)l; r.y = (unsigned)(l >> 32); r.z = (unsigned)h2; r.w = (unsigned)(h2 >> 32);
        return r;
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
            float rv[2][16];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int i = 0; i < 16; ++i) rv[kb][i] = 0.f;
            if (rel) {                                     // kernel-uniform: 32 gathers in flight together
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int i = 0; i < 16; ++i)
                        rv[kb][i] = rel[attn_rel_index(t * HKT + kb * 32 + (i & 3) + 8 * (i >> 2) + 4 * hf, E, ns, h, qi, qh)];
            }
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    sacc[kb][i] = attn_bias_apply(sacc[kb][i], t * HKT + kb * 32 + (i & 3) + 8 * (i >> 2) + 4 * hf, M, E, qi, causal,
                                                  rel != nullptr, rv[kb][i], nbv);
        }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 16; ++i) tmax = fmaxf(tmax, sacc[kb][i]);
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float mnew = fmaxf(mrun, tmax);
        constexpr float LOG2E = 1.4426950408889634f;
        float corr, psum = 0.f;
        if (plain) {
            // every key of the tile is live, so mnew is finite: exp(s - m) = 2^(s log2e - m log2e) is ONE fma + v_exp_f32 per score
            // and needs no guard (this loop is vector-ALU bound: 2 waves per SIMD spend more issue cycles here than on the 16 MFMAs)
            const float mneg = -mnew * LOG2E;
            corr = __builtin_amdgcn_exp2f(fmaf(mrun, LOG2E, mneg));            // mrun = -inf on the first tile: 2^-inf = 0
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float pv = __builtin_amdgcn_exp2f(fmaf(sacc[kb][i], LOG2E, mneg));
                    sacc[kb][i] = pv;
                    psum += pv;
                }
        } else {
            corr = (mnew == -INFINITY) ? 1.f : __expf(mrun - mnew);
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float pv = (mnew == -INFINITY) ? 0.f : __expf(sacc[kb][i] - mnew);
                    sacc[kb][i] = pv;
                    psum += pv;
                }
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
                o[c] = amfma16<BF>(v_frag(Vb, c, sx), pk, o[c]);
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
        hipLaunchKernelGGL((rel ? mqa_flash_fwd_kernel<2, true> : mqa_flash_fwd_kernel<2, false>), grid, dim3(256), 0, (hipStream_t)stream, q, kv,
                           rel, null_bias, out, (float*)nullptr, n, h, n_extra, n_self, causal, scale, 0, (const float*)nullptr);
    else
        hipLaunchKernelGGL((rel ? mqa_flash_fwd_kernel<1, true> : mqa_flash_fwd_kernel<1, false>), grid, dim3(256), 0, (hipStream_t)stream, q, kv,
                           rel, null_bias, out, (float*)nullptr, n, h, n_extra, n_self, causal, scale, 0, (const float*)nullptr);
    return check_launch("mqa_attention_fwd");
}

// The temporal attentions of the pseudo-3D U-Net (EinopsToAndFrom('b c f h w', '(b h w) f c', Attention), imagen_video.py:1351-1354)
// on the channels-last tensors as they stand: q[B][F][P][h d], kv[B][F][P][2 d] (k | v of frame f at pixel p), out like q; sequence =
// (b, p), tokens = the F frames, one extra key / value nullkv[2 d] in front (the learned null row, with null_bias[h] when rel is
// given).  Same kernel and arithmetic as diqt_mqa_attention_fwd on transposed copies + a concatenated kv (bit-identical results).
extern "C" int diqt_mqa_attention_fwd_frames(const float* q, const float* kv, const float* nullkv, const float* rel, const float* null_bias,
                                             float* out, int B, int F, int P, int h, int d, int causal, float scale, void* stream) {
    DIQT_REQUIRE(q && kv && nullkv && out, DIQT_E_ALIGN, "mqa_attention_fwd_frames: null pointer");
    DIQT_REQUIRE(B > 0 && F > 0 && P > 0 && h > 0, DIQT_E_SHAPE, "mqa_attention_fwd_frames: bad shape");
    DIQT_REQUIRE(d == 32 || d == 64, DIQT_E_UNSUPPORTED, "mqa_attention_fwd_frames: dim_head %d (32 or 64 are built)", d);
    DIQT_REQUIRE(aligned16(q) && aligned16(kv) && aligned16(out) && aligned16(nullkv), DIQT_E_ALIGN, "mqa_attention_fwd_frames: pointers must be 16-byte aligned");
    DIQT_REQUIRE((long long)B * P <= 65535, DIQT_E_SHAPE, "mqa_attention_fwd_frames: B * P > 65535");
    const dim3 grid((unsigned)(((long long)F * h + AQ - 1) / AQ), (unsigned)(B * P));
    if (d == 64)
        hipLaunchKernelGGL(mqa_flash_fwd_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, q, kv, rel, null_bias, out, (float*)nullptr, F, h,
                           1, F, causal, scale, P, nullkv);
    else
        hipLaunchKernelGGL(mqa_flash_fwd_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, q, kv, rel, null_bias, out, (float*)nullptr, F, h,
                           1, F, causal, scale, P, nullkv);
    return check_launch("mqa_attention_fwd_frames");
}

extern "C" int diqt_mqa_attention_fwd_lse(const float* q, const float* kv, const float* rel, const float* null_bias, float* out, float* lse,
                                          int G, int n, int h, int d, int n_extra, int n_self, int causal, float scale, void* stream) {
    DIQT_REQUIRE(q && kv && out && lse, DIQT_E_ALIGN, "mqa_attention_fwd_lse: null pointer");
    DIQT_REQUIRE(G > 0 && n > 0 && h > 0 && n_extra >= 0 && n_self >= 0 && n_extra + n_self > 0, DIQT_E_SHAPE, "mqa_attention_fwd_lse: bad shape");
    DIQT_REQUIRE(d == 32 || d == 64, DIQT_E_UNSUPPORTED, "mqa_attention_fwd_lse: dim_head %d (32 or 64 are built)", d);
    DIQT_REQUIRE(!(causal || rel) || n_self == n, DIQT_E_SHAPE, "mqa_attention_fwd_lse: causal / relative bias need n_self == n");
    DIQT_REQUIRE(!null_bias || n_extra >= 1, DIQT_E_SHAPE, "mqa_attention_fwd_lse: null bias without a null key");
    DIQT_REQUIRE(aligned16(q) && aligned16(kv) && aligned16(out), DIQT_E_ALIGN, "mqa_attention_fwd_lse: pointers must be 16-byte aligned");
    DIQT_REQUIRE(G <= 65535, DIQT_E_SHAPE, "mqa_attention_fwd_lse: G > 65535");
    const dim3 grid((unsigned)(((long long)n * h + AQ - 1) / AQ), G);
    if (d == 64)
        hipLaunchKernelGGL(mqa_flash_fwd_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, q, kv, rel, null_bias, out, lse, n, h, n_extra,
                           n_self, causal, scale, 0, (const float*)nullptr);
    else
        hipLaunchKernelGGL(mqa_flash_fwd_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, q, kv, rel, null_bias, out, lse, n, h, n_extra,
                           n_self, causal, scale, 0, (const float*)nullptr);
    return check_launch("mqa_attention_fwd_lse");
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
    // long un-biased sequences (the joint space-time attention): 4-wave workgroups, two per CU (see the kernel); DIQT_ATTN_H_W8=1: always 8 waves
    static const bool w8 = [] { const char* e = getenv("DIQT_ATTN_H_W8"); return e && e[0] == '1'; }();
    const bool four = !w8 && !rel && d == 64 && (long long)n * h >= 4096;
    const int rowsPerWg = four ? 128 : AQH;
    const dim3 grid((unsigned)(((long long)n * h + rowsPerWg - 1) / rowsPerWg), G);
    void (*k)(const float*, const unsigned short*, const float*, const float*, float*, int, int, int, int, int, float, int) =
        four ? (bf16 ? mqa_flash_fwd_h_kernel<2, true, false, 4> : mqa_flash_fwd_h_kernel<2, false, false, 4>)
        : rel ? (d == 64 ? (bf16 ? mqa_flash_fwd_h_kernel<2, true, true> : mqa_flash_fwd_h_kernel<2, false, true>)
                         : (bf16 ? mqa_flash_fwd_h_kernel<1, true, true> : mqa_flash_fwd_h_kernel<1, false, true>))
              : (d == 64 ? (bf16 ? mqa_flash_fwd_h_kernel<2, true, false> : mqa_flash_fwd_h_kernel<2, false, false>)
                         : (bf16 ? mqa_flash_fwd_h_kernel<1, true, false> : mqa_flash_fwd_h_kernel<1, false, false>));
    hipLaunchKernelGGL(k, grid, dim3(four ? 256 : 512), 0, (hipStream_t)stream, q, static_cast<const unsigned short*>(kv), rel, null_bias, out, n, h,
                       n_extra, n_self, causal, scale, round_out ? 1 : 0);
    return check_launch("mqa_attention_fwd_h");
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Fused multi-query attention BACKWARD (training path of Attention.forward, /root/reference/imagen_video.py:410-525): the
// [G, n*h, E+n] score / probability tensors are never written -- both kernels recompute S from q, k and the row log-sum-exp
// the forward saved (diqt_mqa_attention_fwd_lse).  With P = softmax(S), S = scale q k^T + bias:
//     dP = dO V^T,  delta = rowsum(dO . O),  dS = P . (dP - delta),  dQ = scale dS K,  dK = scale dS^T Q,  dV = P^T dO,
//     d rel[i - j + n - 1][head] += dS,  d null_bias[head] += dS[null key].
//  * mqa_flash_bwd_dq_kernel: the forward's transposed formulation, a wave owns 32 query rows as COLUMNS:
//       S^T = K Q^T, dP^T = V dO^T, dQ^T += K^T dS^T   (dS^T goes from the accumulator registers straight into the B operand).
//    It also writes delta for the second kernel and accumulates the bias-gradient tables: per WAVE in LDS (no two lanes of
//    one update instruction hit the same entry: the two lane halves -- keys 4 apart -- update one after the other), over all the
//    batch entries a workgroup walks (persistent in g), then one row per wave in the workspace, summed in a fixed order.
//  * mqa_flash_bwd_dkv_kernel: a wave owns 32 KEYS as columns and walks query tiles staged in its private LDS region
//       S = Q K^T, dP = dO V^T, dV^T += dO^T P, dK^T += Q^T dS   (P and dS again straight from the accumulators);
//    all heads' query rows feed the one shared K/V head.  Short sequences (<= 64 keys: the temporal and 8x8 spatial attentions)
//    split the QUERY tiles over the 4 waves instead and combine the partial sums through LDS in a fixed order.
// Deterministic: no atomics across waves, fixed summation orders.
// ---------------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int acc_row(int i, int hf) { return (i & 3) + 8 * (i >> 2) + 4 * hf; }

template <int ND>
__global__ __launch_bounds__(256, DIQT_DQ_OCC) void mqa_flash_bwd_dq_kernel(const float* __restrict__ q, const float* __restrict__ kv,
                                                                  const float* __restrict__ rel, const float* __restrict__ null_bias,
                                                                  const float* __restrict__ out, const float* __restrict__ dout,
                                                                  const float* __restrict__ lse, float* __restrict__ dq,
                                                                  float* __restrict__ delta, float* __restrict__ tbl_part,
                                                                  float* __restrict__ dnull_part, int G, int n, int h, int E, int ns,
                                                                  int causal, float scale) {
    constexpr int D = 32 * ND, ROW = D + 4, NPF = AKT * (2 * D / 4) / 256;
    __shared__ __attribute__((aligned(16))) float KVs[2][2][AKT * ROW];
    extern __shared__ float tbl_all[];                     // [4 waves][(2 ns - 1) * h] when rel
    const int M = E + ns, R = n * h;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hf = lane >> 5;
    const int r = blockIdx.x * AQ + wave * 32 + l31;
    const bool rvalid = r < R;
    const int rc = rvalid ? r : R - 1;
    const int qi = rc / h, qh = rc % h;
    const int TBL = rel ? (2 * ns - 1) * h : 0;
    float* tbl = tbl_all + wave * TBL;
    for (int e = lane; e < TBL; e += 64) tbl[e] = 0.f;
    const float nbv = null_bias ? null_bias[qh] : 0.f;
    const int kbase = (E == 1) ? 1 : 0;
    const int ntiles = (M - kbase + AKT - 1) / AKT;
    float dnb = 0.f;                                       // d null_bias contribution of this lane's query row (lane half 0 only)

    for (int g = blockIdx.y; g < G; g += gridDim.y) {
        const float* qg = q + ((size_t)g * R + rc) * D;
        const float* og = out + ((size_t)g * R + rc) * D;
        const float* dog = dout + ((size_t)g * R + rc) * D;
        const float* kvg = kv + (size_t)g * M * 2 * D;
        float qreg[D / 2], doreg[D / 2];
        float dl = 0.f;
#pragma unroll
        for (int gq = 0; gq < D / 8; ++gq) {
            const float4 v = *reinterpret_cast<const float4*>(qg + 8 * gq + 4 * hf);
            const float4 w = *reinterpret_cast<const float4*>(dog + 8 * gq + 4 * hf);
            const float4 ov = *reinterpret_cast<const float4*>(og + 8 * gq + 4 * hf);
            qreg[4 * gq] = v.x * scale; qreg[4 * gq + 1] = v.y * scale; qreg[4 * gq + 2] = v.z * scale; qreg[4 * gq + 3] = v.w * scale;
            doreg[4 * gq] = w.x; doreg[4 * gq + 1] = w.y; doreg[4 * gq + 2] = w.z; doreg[4 * gq + 3] = w.w;
            dl = fmaf(w.x, ov.x, dl); dl = fmaf(w.y, ov.y, dl); dl = fmaf(w.z, ov.z, dl); dl = fmaf(w.w, ov.w, dl);
        }
        dl += __shfl_xor(dl, 32, 64);
        const float L = lse[(size_t)g * R + rc];
        if (rvalid && hf == 0) delta[(size_t)g * R + r] = dl;
        f32x16 dqt[ND];
#pragma unroll
        for (int c = 0; c < ND; ++c)
#pragma unroll
            for (int i = 0; i < 16; ++i) dqt[c][i] = 0.f;

        if (kbase) {                                       // the lone null key by VALU (see the forward)
            float s0 = 0.f, dp0 = 0.f;
#pragma unroll
            for (int gq = 0; gq < D / 8; ++gq) {
                const float4 kk = *reinterpret_cast<const float4*>(kvg + 8 * gq + 4 * hf);
                const float4 vv = *reinterpret_cast<const float4*>(kvg + D + 8 * gq + 4 * hf);
                s0 = fmaf(qreg[4 * gq], kk.x, s0); s0 = fmaf(qreg[4 * gq + 1], kk.y, s0);
                s0 = fmaf(qreg[4 * gq + 2], kk.z, s0); s0 = fmaf(qreg[4 * gq + 3], kk.w, s0);
                dp0 = fmaf(doreg[4 * gq], vv.x, dp0); dp0 = fmaf(doreg[4 * gq + 1], vv.y, dp0);
                dp0 = fmaf(doreg[4 * gq + 2], vv.z, dp0); dp0 = fmaf(doreg[4 * gq + 3], vv.w, dp0);
            }
            s0 += __shfl_xor(s0, 32, 64);
            dp0 += __shfl_xor(dp0, 32, 64);
            const float p0 = __expf(s0 + nbv - L);
            const float ds0 = rvalid ? p0 * (dp0 - dl) : 0.f;
            if (hf == 0) dnb += ds0;
#pragma unroll
            for (int c = 0; c < ND; ++c)
#pragma unroll
                for (int i = 0; i < 16; ++i) dqt[c][i] = ds0 * kvg[32 * c + acc_row(i, hf)];
        }

        float4 pre[NPF];
        auto load_tile = [&](int t) {
#pragma unroll
            for (int u = 0; u < NPF; ++u) {
                const int e = u * 256 + tid;
                const int key = e / (2 * D / 4), c4 = (e % (2 * D / 4)) * 4;
                const int j = kbase + t * AKT + key;
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
        __syncthreads();                                   // the previous batch entry's last tile is retired
        if (ntiles > 0) { load_tile(0); store_tile(0); }
        __syncthreads();
        for (int t = 0; t < ntiles; ++t) {
            const float* Ks = KVs[t & 1][0];
            const float* Vs = KVs[t & 1][1];
            if (t + 1 < ntiles) load_tile(t + 1);
            // ---- S^T = K Q^T and dP^T = V dO^T (rows = keys of the tile, columns = queries) ----
            f32x16 s, dp;
#pragma unroll
            for (int i = 0; i < 16; ++i) { s[i] = 0.f; dp[i] = 0.f; }
            const float* kp = Ks + l31 * ROW + 4 * hf;
            const float* vp4 = Vs + l31 * ROW + 4 * hf;
            f32x4a ka = *reinterpret_cast<const f32x4a*>(kp), va = *reinterpret_cast<const f32x4a*>(vp4);      // reads one group ahead
#pragma unroll
            for (int gq = 0; gq < D / 8; ++gq) {
                f32x4a kn = ka, vn4 = va;
                if (gq + 1 < D / 8) {
                    kn = *reinterpret_cast<const f32x4a*>(kp + 8 * (gq + 1));
                    vn4 = *reinterpret_cast<const f32x4a*>(vp4 + 8 * (gq + 1));
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    s = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[e], qreg[4 * gq + e], s, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_32x32x2f32(va[e], doreg[4 * gq + e], dp, 0, 0, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
                ka = kn; va = vn4;
            }
            // ---- P^T = exp(S^T + bias - L), dS^T = P^T (dP^T - delta); bias-gradient tables ----
            float rv[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) rv[i] = 0.f;
            if (rel) {                                     // kernel-uniform: 16 gathers in flight together
#pragma unroll
                for (int i = 0; i < 16; ++i) rv[i] = rel[attn_rel_index(kbase + t * AKT + acc_row(i, hf), E, ns, h, qi, qh)];
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int j = kbase + t * AKT + acc_row(i, hf);
                const float sv = attn_bias_apply(s[i], j, M, E, qi, causal, rel != nullptr, rv[i], nbv);
                const float p = (sv == -INFINITY) ? 0.f : __expf(sv - L);
                s[i] = rvalid ? p * (dp[i] - dl) : 0.f;
            }
            if (rel || (null_bias && !kbase)) {            // kernel-uniform
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    if (hf == half && rvalid) {        // rows beyond R are clamped onto the last row: they must not take part in its update
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            const int j = kbase + t * AKT + acc_row(i, hf);
                            if (j >= E && j < M) {
                                if (rel) {
                                    const int idx = max(0, min(qi - (j - E) + ns - 1, 2 * ns - 2));
                                    tbl[idx * h + qh] += s[i];          // masked entries carry dS = 0 (an LDS atomic add here is slower: 1010 vs 855 us)
                                }
                            } else if (j == E - 1 && null_bias) dnb += s[i];
                        }
                    }
                }
            }
            // ---- dQ^T += K^T dS^T : step i uses the key pair held in register i of the two lane halves ----
            {
                float kc[ND], kn2[ND];
#pragma unroll
                for (int c = 0; c < ND; ++c) kc[c] = Ks[acc_row(0, hf) * ROW + l31 + 32 * c];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    if (i + 1 < 16) {
                        const float* kq = Ks + acc_row(i + 1, hf) * ROW + l31;
#pragma unroll
                        for (int c = 0; c < ND; ++c) kn2[c] = kq[32 * c];
                    }
#pragma unroll
                    for (int c = 0; c < ND; ++c) dqt[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(kc[c], s[i], dqt[c], 0, 0, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, ND, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, ND, 0);
#pragma unroll
                    for (int c = 0; c < ND; ++c) kc[c] = kn2[c];
                }
            }
            if (t + 1 < ntiles) store_tile((t + 1) & 1);
            __syncthreads();
        }
        if (rvalid) {
            float* dqg = dq + ((size_t)g * R + r) * D;
#pragma unroll
            for (int c = 0; c < ND; ++c)
#pragma unroll
                for (int i = 0; i < 16; i += 4) {
                    const int dd = 32 * c + 8 * (i >> 2) + 4 * hf;
                    *reinterpret_cast<float4*>(dqg + dd) = make_float4(dqt[c][i] * scale, dqt[c][i + 1] * scale, dqt[c][i + 2] * scale, dqt[c][i + 3] * scale);
                }
        }
    }
    // ---- bias-gradient partials of this wave: one row of the workspace each ----
    const size_t wrow = ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave;
    if (TBL) {
        __builtin_amdgcn_s_waitcnt(0xc07f);                // this wave's own LDS updates are done (in-order LDS queue)
        for (int e = lane; e < TBL; e += 64) tbl_part[wrow * TBL + e] = tbl[e];
    }
    // (a null key at index 4 .. 7 of its group of eight -- five or more extra keys -- sits in the accumulator rows of lane half 1)
    dnb += __shfl_xor(dnb, 32, 64);
    if (dnull_part && hf == 0) dnull_part[wrow * 32 + l31] = dnb;
}

// drel[idx][head] = sum over the per-wave rows; dnull[head] = sum over rows and over the lanes whose query row has that head.
// Fixed summation orders.  Blocks 0 .. nTblBlocks-1 own 64 table entries each (4 waves x a quarter of the rows, 8 loads in flight);
// the last block sums the null-bias partials (every thread its own entries into a private LDS column, then columns in thread order).
__global__ __launch_bounds__(256) void attn_bias_reduce_kernel(const float* __restrict__ tbl_part, const float* __restrict__ dnull_part,
                                                               float* __restrict__ drel, float* __restrict__ dnull, int rows, int TBL,
                                                               int gx, int h, int R, int nTblBlocks) {
    extern __shared__ float red_sm[];                      // max(4 * 64, 256 * h) floats
    if ((int)blockIdx.x < nTblBlocks) {
        const int lane = threadIdx.x & 63, sg = threadIdx.x >> 6;
        const int e = blockIdx.x * 64 + lane;
        const int k0 = sg * rows / 4, k1 = (sg + 1) * rows / 4;
        float s = 0.f;
        if (e < TBL) {
            int k = k0;
            for (; k + 8 <= k1; k += 8) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = tbl_part[(size_t)(k + u) * TBL + e];
#pragma unroll
                for (int u = 0; u < 8; ++u) s += v[u];
            }
            for (; k < k1; ++k) s += tbl_part[(size_t)k * TBL + e];
        }
        red_sm[sg * 64 + lane] = s;
        __syncthreads();
        if (sg == 0 && e < TBL) drel[e] = ((red_sm[lane] + red_sm[64 + lane]) + red_sm[128 + lane]) + red_sm[192 + lane];
        return;
    }
    if (dnull) {
        for (int hh = 0; hh < h; ++hh) red_sm[threadIdx.x * h + hh] = 0.f;
        const int total = rows * 32;
        for (int e2 = threadIdx.x; e2 < total; e2 += 256) {
            const int row = e2 / 32, l = e2 % 32;          // row = (by * gx + bx) * 4 + wave
            const int bx = (row / 4) % gx, wave = row % 4;
            const int r = bx * AQ + wave * 32 + l;
            if (r < R) red_sm[threadIdx.x * h + r % h] += dnull_part[e2];
        }
        __syncthreads();
        if ((int)threadIdx.x < h) {
            float sacc = 0.f;
            for (int t2 = 0; t2 < 256; ++t2) sacc += red_sm[t2 * h + threadIdx.x];
            dnull[threadIdx.x] = sacc;
        }
    }
}

template <int ND>
__global__ __launch_bounds__(256, 1) void mqa_flash_bwd_dkv_kernel(const float* __restrict__ q, const float* __restrict__ kv,
                                                                   const float* __restrict__ rel, const float* __restrict__ null_bias,
                                                                   const float* __restrict__ dout, const float* __restrict__ lse,
                                                                   const float* __restrict__ delta, float* __restrict__ dkv, int n, int h,
                                                                   int E, int ns, int causal, float scale, int KW, int G, int relLds, int EV) {
    constexpr int D = 32 * ND, ROW = D + 4, NPQ = 32 * D / 4 / 64;       // float4 pieces of a 32-row tile per lane
    extern __shared__ __attribute__((aligned(16))) float smem_dkv[];
    // XCD-aware block mapping: workgroups are dealt to the 8 XCDs round-robin by linear id.  All key-tile workgroups of a batch entry
    // walk the same Q / dO tiles at the same pace, so they belong behind ONE L2: batch entry g = xcd + 8 * (...) instead of every
    // XCD streaming every entry's Q and dO (the joint space-time attentions: 8 entries x 64 key tiles, 8.4 MB of Q + dO each).
    int g = blockIdx.y, bx = blockIdx.x;
    if ((gridDim.y & 7) == 0) {
        const unsigned id = blockIdx.y * gridDim.x + blockIdx.x, slot = id >> 3;
        g = (int)((id & 7u) + 8u * (slot / gridDim.x));
        bx = (int)(slot % gridDim.x);
    }
    const int M = E + ns, R = n * h;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hf = lane >> 5;
    // KW == 0: thousands of SHORT sequences (one key tile each: the temporal attentions, 8192 x 32 frames).  A workgroup per sequence is
    // a latency chain -- K/V load, two query tiles per wave, a 48-KB combine through LDS, store, and the next workgroup's dispatch --
    // that leaves the CU idle two thirds of the time at one workgroup per CU.  Instead every WAVE takes a sequence of its own and walks
    // all its query tiles (next tile prefetched in registers): no combine, no barrier, four independent chains per CU.
    const bool perWave = KW == 0;
    // The bias tables -- rel[(2 ns - 1) x h] and null_bias[h] -- are gathered 16 times per lane and tile.  Global loads inside the tile
    // loop queue BEHIND the next tile's prefetch and their `s_waitcnt vmcnt` waits for it too (loads retire in order), so short tables
    // (relLds floats reserved by the host behind the staging regions) are copied to LDS once and read with ds_read.
    float* const Rs = smem_dkv + (size_t)4 * (2 * 32 * ROW + 128 + 2 * D);
    float* const NBs = Rs + relLds;
    const bool tblLds = relLds > 0;
    if (tblLds) {
        for (int e = threadIdx.x; e < relLds; e += 256) Rs[e] = rel[e];
        if (null_bias && threadIdx.x < h) NBs[threadIdx.x] = null_bias[threadIdx.x];
        __syncthreads();
    }
    if (perWave) {
        g = g * 4 + wave;
        if (g >= G) return;                                // (no barriers behind this point on this path)
    }
    const int QW = perWave ? 1 : 4 / KW;                   // waves that share a key tile and split the query tiles
    const int kt = perWave ? bx : bx * KW + wave % KW, qw = perWave ? 0 : wave / KW;
    // EV (= E or 0) extra keys in front of the self keys -- the lone null key of the temporal attentions, context tokens + null key of the
    // joint attentions -- are taken by the VALU (below), one by each of the first EV key tiles, instead of costing a 32-key tile of their
    // own: 2048 + 5 keys were 65 tiles x 8 entries = 520 workgroups on 256 CUs, a third round for eight of them (2.84 vs 2.35 ms)
    const int kbase = EV;
    const int j = kbase + kt * 32 + l31;                   // this lane's key (column)
    const bool jvalid = j < M;
    const int jc = jvalid ? j : M - 1;
    const float* kvg = kv + ((size_t)g * M + jc) * 2 * D;
    float* Qs = smem_dkv + (size_t)wave * (2 * 32 * ROW + 128 + 2 * D);      // wave-private: Q tile, dO tile, lse, delta, p0, ds0 of 32 query rows, null k | v
    float* dOs = Qs + 32 * ROW;
    float* Ls = dOs + 32 * ROW;
    float* Dls = Ls + 32;
    float* P0s = Dls + 32;
    float* dS0s = P0s + 32;

    // K^T and V^T operands (k-index dd = 8 gq + 4 hf + e, like qreg in the forward); K carries the soft-max scale
    float kreg[D / 2], vreg[D / 2];
#pragma unroll
    for (int gq = 0; gq < D / 8; ++gq) {
        const float4 a = *reinterpret_cast<const float4*>(kvg + 8 * gq + 4 * hf);
        const float4 b = *reinterpret_cast<const float4*>(kvg + D + 8 * gq + 4 * hf);
        kreg[4 * gq] = a.x * scale; kreg[4 * gq + 1] = a.y * scale; kreg[4 * gq + 2] = a.z * scale; kreg[4 * gq + 3] = a.w * scale;
        vreg[4 * gq] = b.x; vreg[4 * gq + 1] = b.y; vreg[4 * gq + 2] = b.z; vreg[4 * gq + 3] = b.w;
    }
    f32x16 dkt[ND], dvt[ND];
#pragma unroll
    for (int c = 0; c < ND; ++c)
#pragma unroll
        for (int i = 0; i < 16; ++i) { dkt[c][i] = 0.f; dvt[c][i] = 0.f; }
    const bool doNull = kt < EV;                           // wave-uniform: the waves of key tile kt also own extra key row kt
    const float* kv0 = kv + ((size_t)g * M + (doNull ? kt : 0)) * 2 * D;      // that key / value row
    // ... kept in the wave's LDS region: read from global memory inside the tile loop its loads queue BEHIND the next tile's prefetch,
    // and the `s_waitcnt vmcnt(0)` in front of their first use waits for that prefetch too (the temporal attentions, where every wave
    // owns the null key: 714 -> 505 us of this kernel were that wait and the scalar column sums below)
    float* K0s = Qs + 2 * 32 * ROW + 128;
    if (doNull) {
        for (int e = lane; e < 2 * D; e += 64) K0s[e] = kv0[e];
    }
    float dkn = 0.f, dvn = 0.f;                            // d k_null[lane], d v_null[lane] (lane = head-dim index, lanes < D)

    const int nqt = (R + 31) / 32;
    const bool active = kbase + kt * 32 < M || doNull;     // wave-uniform
    const bool plain = !rel && !null_bias && !causal;      // kernel-uniform
    const int hs = (h & (h - 1)) == 0 ? __builtin_ctz(h) : -1;
    typedef float f32x4v __attribute__((ext_vector_type(4)));      // plain vector registers (HIP's float4 class kept these arrays in scratch)
    f32x4v pq[NPQ], pdo[NPQ];
    float pl = 0.f, pdl = 0.f;
    auto load_q = [&](int qt) __attribute__((always_inline)) {      // next query tile -> registers (unconditional, clamped rows)
        const int r0 = qt * 32;
#pragma unroll
        for (int u = 0; u < NPQ; ++u) {
            const int e = u * 64 + lane;
            const int row = e / (D / 4), c4 = (e % (D / 4)) * 4;
            const int rr = min(r0 + row, R - 1);
            pq[u] = *reinterpret_cast<const f32x4v*>(q + ((size_t)g * R + rr) * D + c4);
            pdo[u] = *reinterpret_cast<const f32x4v*>(dout + ((size_t)g * R + rr) * D + c4);
        }
        const int rr = min(r0 + l31, R - 1);
        pl = lse[(size_t)g * R + rr];
        pdl = delta[(size_t)g * R + rr];
    };
    // causal: a query tile whose largest token index is below the key tile's smallest self-key index sees only masked scores
    auto tile_live = [&](int qt) __attribute__((always_inline)) { return !(causal && !doNull && (kt * 32 + kbase - E) > (min(qt * 32 + 31, R - 1)) / h); };
    int qt = qw;
    if (active) {
        while (qt < nqt && !tile_live(qt)) qt += QW;
        if (qt < nqt) load_q(qt);
    }
    if (active)
    while (qt < nqt) {
        const int r0 = qt * 32;
        // ---- registers -> the wave-private LDS tile (no workgroup barrier: the LDS queue of a wave is in order) ----
#pragma unroll
        for (int u = 0; u < NPQ; ++u) {
            const int e = u * 64 + lane;
            const int row = e / (D / 4), c4 = (e % (D / 4)) * 4;
            *reinterpret_cast<f32x4v*>(Qs + row * ROW + c4) = pq[u];
            *reinterpret_cast<f32x4v*>(dOs + row * ROW + c4) = pdo[u];
        }
        if (lane < 32) { Ls[lane] = pl; Dls[lane] = pdl; }
        int qn = qt + QW;
        while (qn < nqt && !tile_live(qn)) qn += QW;
        if (qn < nqt) load_q(qn);                          // in flight during this tile's MFMAs
        const float* qp = Qs + l31 * ROW + 4 * hf;
        const float* dop = dOs + l31 * ROW + 4 * hf;
        if (doNull) {
            // scores of the null key for the 32 rows: lane (row l31, half hf) sums its half of the head dimension
            float s0 = 0.f, dp0 = 0.f;
#pragma unroll
            for (int gq = 0; gq < D / 8; ++gq) {
                const float4 a = *reinterpret_cast<const float4*>(qp + 8 * gq);
                const float4 b = *reinterpret_cast<const float4*>(dop + 8 * gq);
                const float4 kk = *reinterpret_cast<const float4*>(K0s + 8 * gq + 4 * hf);
                const float4 vv = *reinterpret_cast<const float4*>(K0s + D + 8 * gq + 4 * hf);
                s0 = fmaf(a.x, kk.x, s0); s0 = fmaf(a.y, kk.y, s0); s0 = fmaf(a.z, kk.z, s0); s0 = fmaf(a.w, kk.w, s0);
                dp0 = fmaf(b.x, vv.x, dp0); dp0 = fmaf(b.y, vv.y, dp0); dp0 = fmaf(b.z, vv.z, dp0); dp0 = fmaf(b.w, vv.w, dp0);
            }
            s0 += __shfl_xor(s0, 32, 64);
            dp0 += __shfl_xor(dp0, 32, 64);
            const int rr = r0 + l31;
            const float nbv0 = (null_bias && kt == E - 1) ? (tblLds ? NBs[min(rr, R - 1) % h] : null_bias[min(rr, R - 1) % h]) : 0.f;      // (the null key is the last extra key)
            const float p0 = rr < R ? __expf(s0 * scale + nbv0 - Ls[l31]) : 0.f;
            const float ds0 = p0 * (dp0 - Dls[l31]);
            // lane = head-dim index: column sums over the 32 rows.  p0 / ds0 of row rq live in lane rq: broadcast through a scalar
            // register (v_readlane) instead of an LDS round trip per row; the 64 operand reads of the unrolled loop pipeline freely.
            if (lane < D) {
#pragma unroll
                for (int rq = 0; rq < 32; ++rq) {
                    const float pb = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, p0), rq));
                    const float db = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ds0), rq));
                    dvn = fmaf(pb, dOs[rq * ROW + lane], dvn);
                    dkn = fmaf(db, Qs[rq * ROW + lane], dkn);
                }
            }
        }
        if (kbase + kt * 32 < M) {                         // wave-uniform: this wave has self keys (a lone-null-key launch may not)
        // ---- S = Q K^T, dP = dO V^T (rows = queries of the tile, columns = this wave's keys) ----
        f32x16 s, dp;
#pragma unroll
        for (int i = 0; i < 16; ++i) { s[i] = 0.f; dp[i] = 0.f; }
        f32x4a qa = *reinterpret_cast<const f32x4a*>(qp), da = *reinterpret_cast<const f32x4a*>(dop);      // reads one group ahead
#pragma unroll
        for (int gq = 0; gq < D / 8; ++gq) {
            f32x4a qn = qa, dn = da;
            if (gq + 1 < D / 8) {
                qn = *reinterpret_cast<const f32x4a*>(qp + 8 * (gq + 1));
                dn = *reinterpret_cast<const f32x4a*>(dop + 8 * (gq + 1));
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                s = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[e], kreg[4 * gq + e], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x2f32(da[e], vreg[4 * gq + e], dp, 0, 0, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
            qa = qn; da = dn;
        }
        // ---- P = exp(S + bias - L[row]), dS = P (dP - delta[row]) ----
        // row statistics of the 16 accumulator rows: 4 vector reads each instead of a read + wait per row inside the loop below
        float lrow[16], drow[16];
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            const float4 lv = *reinterpret_cast<const float4*>(Ls + 8 * q4 + 4 * hf);
            const float4 dv = *reinterpret_cast<const float4*>(Dls + 8 * q4 + 4 * hf);
            lrow[4 * q4] = lv.x; lrow[4 * q4 + 1] = lv.y; lrow[4 * q4 + 2] = lv.z; lrow[4 * q4 + 3] = lv.w;
            drow[4 * q4] = dv.x; drow[4 * q4 + 1] = dv.y; drow[4 * q4 + 2] = dv.z; drow[4 * q4 + 3] = dv.w;
        }
        f32x16 ds;
        if (plain) {
            // no bias, no mask (the joint space-time attentions: 2048 tokens x 8 heads per batch entry): only ragged keys / rows die
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int rr = r0 + acc_row(i, hf);
                const float e = __expf(s[i] - lrow[i]);
                const float p = (!jvalid || rr >= R) ? 0.f : e;
                s[i] = p;
                ds[i] = p * (dp[i] - drow[i]);
            }
        } else {
            int qis[16];
            float rv[16], nbvs[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {                 // token, head of the row: shifts when h is a power of two (an integer
                const int rcl = min(r0 + acc_row(i, hf), R - 1);                 // division costs ~2 MFMAs)
                int qh;
                if (hs >= 0) { qis[i] = rcl >> hs; qh = rcl & (h - 1); } else { qis[i] = rcl / h; qh = rcl - qis[i] * h; }
                rv[i] = __int_as_float(qh);                // parked until the gathers below
                nbvs[i] = 0.f;
            }
            if (tblLds) {                                  // kernel-uniform: tables in LDS
                if (null_bias) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) nbvs[i] = NBs[__float_as_int(rv[i])];
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) rv[i] = Rs[attn_rel_index(j, E, ns, h, qis[i], __float_as_int(rv[i]))];
            } else {
                if (null_bias) {                           // kernel-uniform
#pragma unroll
                    for (int i = 0; i < 16; ++i) nbvs[i] = null_bias[__float_as_int(rv[i])];
                }
                if (rel) {                                 // kernel-uniform: 16 gathers in flight together
#pragma unroll
                    for (int i = 0; i < 16; ++i) rv[i] = rel[attn_rel_index(j, E, ns, h, qis[i], __float_as_int(rv[i]))];
                }
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int rr = r0 + acc_row(i, hf);
                const float sv = attn_bias_apply(s[i], j, M, E, qis[i], causal, rel != nullptr, rv[i], nbvs[i]);
                const float e = __expf(sv - lrow[i]);      // exp(-inf) = 0 for masked scores
                const float p = (sv == -INFINITY || rr >= R) ? 0.f : e;
                s[i] = p;
                ds[i] = p * (dp[i] - drow[i]);
            }
        }
        // ---- dV^T += dO^T P, dK^T += Q^T dS : step i contracts the query pair held in register i of the two lane halves ----
        {
            float oc[ND], qc[ND], on[ND], qn2[ND];
#pragma unroll
            for (int c = 0; c < ND; ++c) { oc[c] = dOs[acc_row(0, hf) * ROW + l31 + 32 * c]; qc[c] = Qs[acc_row(0, hf) * ROW + l31 + 32 * c]; }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (i + 1 < 16) {
                    const int row = acc_row(i + 1, hf);
#pragma unroll
                    for (int c = 0; c < ND; ++c) { on[c] = dOs[row * ROW + l31 + 32 * c]; qn2[c] = Qs[row * ROW + l31 + 32 * c]; }
                }
#pragma unroll
                for (int c = 0; c < ND; ++c) {
                    dvt[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(oc[c], s[i], dvt[c], 0, 0, 0);
                    dkt[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(qc[c], ds[i], dkt[c], 0, 0, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x100, 2 * ND, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 2 * ND, 0);
#pragma unroll
                for (int c = 0; c < ND; ++c) { oc[c] = on[c]; qc[c] = qn2[c]; }
            }
        }
        }
        qt = qn;
    }
    // ---- combine the partial sums of the waves that share a key tile (fixed order: query split 1, 2, 3 onto 0) ----
    if (QW > 1) {
        __syncthreads();                                   // every wave is done with its staging region
        float* red = smem_dkv;                             // region of wave w (qw > 0) at (w - KW): [2 ND][16][64] + [2][64] floats
        const size_t per = (size_t)2 * ND * 16 * 64 + 128;
        if (qw > 0) {
            float* dst = red + (size_t)(wave - KW) * per;
#pragma unroll
            for (int c = 0; c < ND; ++c)
#pragma unroll
                for (int i = 0; i < 16; ++i) { dst[((c * 2) * 16 + i) * 64 + lane] = dkt[c][i]; dst[((c * 2 + 1) * 16 + i) * 64 + lane] = dvt[c][i]; }
            dst[2 * ND * 16 * 64 + lane] = dkn;
            dst[2 * ND * 16 * 64 + 64 + lane] = dvn;
        }
        __syncthreads();
        if (qw == 0) {
            for (int w2 = 1; w2 < QW; ++w2) {
                const float* src = red + (size_t)(w2 * KW + wave - KW) * per;
#pragma unroll
                for (int c = 0; c < ND; ++c)
#pragma unroll
                    for (int i = 0; i < 16; ++i) { dkt[c][i] += src[((c * 2) * 16 + i) * 64 + lane]; dvt[c][i] += src[((c * 2 + 1) * 16 + i) * 64 + lane]; }
                dkn += src[2 * ND * 16 * 64 + lane];
                dvn += src[2 * ND * 16 * 64 + 64 + lane];
            }
        }
    }
    if (qw == 0 && jvalid && kbase + kt * 32 < M) {
        float* og = dkv + ((size_t)g * M + j) * 2 * D;
#pragma unroll
        for (int c = 0; c < ND; ++c)
#pragma unroll
            for (int i = 0; i < 16; i += 4) {
                const int dd = 32 * c + 8 * (i >> 2) + 4 * hf;
                // S = scale q k^T: kreg carried the scale for S, the gradient wrt k takes it once more
                *reinterpret_cast<float4*>(og + dd) = make_float4(dkt[c][i] * scale, dkt[c][i + 1] * scale, dkt[c][i + 2] * scale, dkt[c][i + 3] * scale);
                *reinterpret_cast<float4*>(og + D + dd) = make_float4(dvt[c][i], dvt[c][i + 1], dvt[c][i + 2], dvt[c][i + 3]);
            }
    }
    if (qw == 0 && doNull && lane < D) {
        float* og = dkv + ((size_t)g * M + kt) * 2 * D;
        og[lane] = dkn * scale;
        og[D + lane] = dvn;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Backward of the SHORT-sequence attentions in ONE kernel (round 3): the temporal attentions of the pseudo-3D U-Net -- thousands of
// sequences of <= 32 frames, one learned null key in front (E == 1), 8 heads x 64 -- are bound by HBM bytes, not by their five
// products: q, dO and the forward's output are 8x wider than x (537 MB each at the 32 x 32 level).  The two-kernel backward above reads
// q and dO twice and `out` once (2.7 GB + the dQ write); here every WAVE takes whole sequences (persistent), keeps K / V of the
// sequence in registers (and K once more in LDS for the dQ product) and walks its query tiles ONCE:
//   rows = queries:   S = Q K^T, dP = dO V^T  ->  P, dS  ->  dV^T += dO^T P, dK^T += Q^T dS          (as mqa_flash_bwd_dkv_kernel)
//   cols = queries:   S^T = K Q^T, dP^T = V dO^T (the SAME operand registers, A and B swapped)  ->  dS^T  ->  dQ^T = K^T dS^T + null key
// (the score tile is recomputed in both orientations -- 7 MFMA products instead of 5 -- because the accumulator layout of dS feeds
// the dK^T product directly and that of dS^T the dQ^T product; a transpose through LDS would cost more than 64 MFMAs per tile).
// delta = rowsum(dO * out) is computed in the kernel from the tile of `out`, which arrives by LDS-DMA (no registers) while the
// previous tile computes.  Bias-gradient tables accumulate per wave in LDS over all its sequences; one workspace row per wave, summed
// in a fixed order by attn_bias_reduce_kernel.  Deterministic.
// ---------------------------------------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void lds_void_at;

template <int ND>
__global__ __launch_bounds__(256, 1) void mqa_seq_bwd_kernel(const float* __restrict__ q, const float* __restrict__ kv,
                                                             const float* __restrict__ rel, const float* __restrict__ null_bias,
                                                             const float* __restrict__ out, const float* __restrict__ dout,
                                                             const float* __restrict__ lse, float* __restrict__ dq, float* __restrict__ dkv,
                                                             float* __restrict__ tbl_part, float* __restrict__ dnull_part, int n, int h,
                                                             int causal, float scale, int G, int relLds, unsigned outBytes) {
    constexpr int D = 32 * ND, ROW = D + 4, NPQ = 32 * D / 4 / 64;
    constexpr int E = 1;
    extern __shared__ __attribute__((aligned(1024))) float smem_seq[];
    const int ns = n, M = E + ns, R = n * h;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hf = lane >> 5;
    const int TBL = rel ? (2 * ns - 1) * h : 0;
    // wave-private region: out tile (LDS-DMA target, unpadded, first: 1-KiB aligned) | Q | dO | K | lse | delta | null k|v | bias table
    constexpr int PERW = 32 * D + 3 * 32 * ROW + 64 + 2 * D;
    const int perw = (PERW + TBL + 255) / 256 * 256;                               // floats, keeps every wave's DMA target 1-KiB aligned
    float* const base = smem_seq + (size_t)wave * perw;
    float* const Os = base;
    float* const Qs = Os + 32 * D;
    float* const dOs = Qs + 32 * ROW;
    float* const Ks = dOs + 32 * ROW;
    float* const Ls = Ks + 32 * ROW;
    float* const Dls = Ls + 32;
    float* const K0s = Dls + 32;
    float* const tbl = K0s + 2 * D;
    float* const Rs = smem_seq + (size_t)4 * perw;                                  // shared: rel table, null bias
    float* const NBs = Rs + relLds;
    for (int e = threadIdx.x; e < relLds; e += 256) Rs[e] = rel[e];
    if (null_bias && threadIdx.x < h) NBs[threadIdx.x] = null_bias[threadIdx.x];
    for (int e = lane; e < TBL; e += 64) tbl[e] = 0.f;
    __syncthreads();
    const bool plain = !rel && !null_bias && !causal;      // kernel-uniform
    const int hs = (h & (h - 1)) == 0 ? __builtin_ctz(h) : -1;
    const auto rs_o = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(out), 0, (int)outBytes, 0x00020000);
    const unsigned osLds = (unsigned)(size_t)(lds_void_at*)Os;
    typedef float f32x4v __attribute__((ext_vector_type(4)));
    const int nqt = (R + 31) / 32;
    float dnb = 0.f;
    const int nw = gridDim.x * 4;

    for (int g = blockIdx.x * 4 + wave; g < G; g += nw) {
        // ---- K, V of the sequence: this lane's key j = 1 + l31 ----
        const int j = E + l31;
        const bool jvalid = j < M;
        const int jc = jvalid ? j : M - 1;
        const float* kvg = kv + ((size_t)g * M + jc) * 2 * D;
        const float* kv0 = kv + (size_t)g * M * 2 * D;
        float kreg[D / 2], vreg[D / 2];
#pragma unroll
        for (int gq = 0; gq < D / 8; ++gq) {
            float4 a = *reinterpret_cast<const float4*>(kvg + 8 * gq + 4 * hf);
            const float4 b = *reinterpret_cast<const float4*>(kvg + D + 8 * gq + 4 * hf);
            if (!jvalid) a = make_float4(0.f, 0.f, 0.f, 0.f);                       // K^T dS^T must not pick up a clamped row
            *reinterpret_cast<float4*>(Ks + l31 * ROW + 8 * gq + 4 * hf) = a;       // unscaled: dQ takes the scale at the store
            kreg[4 * gq] = a.x * scale; kreg[4 * gq + 1] = a.y * scale; kreg[4 * gq + 2] = a.z * scale; kreg[4 * gq + 3] = a.w * scale;
            vreg[4 * gq] = b.x; vreg[4 * gq + 1] = b.y; vreg[4 * gq + 2] = b.z; vreg[4 * gq + 3] = b.w;
        }
        for (int e = lane; e < 2 * D; e += 64) K0s[e] = kv0[e];
        f32x16 dkt[ND], dvt[ND];
#pragma unroll
        for (int c = 0; c < ND; ++c)
#pragma unroll
            for (int i = 0; i < 16; ++i) { dkt[c][i] = 0.f; dvt[c][i] = 0.f; }
        float dkn = 0.f, dvn = 0.f;

        f32x4v pq[NPQ], pdo[NPQ];
        float pl = 0.f;
        auto load_q = [&](int qt) __attribute__((always_inline)) {
            const int r0 = qt * 32;
#pragma unroll
            for (int u = 0; u < NPQ; ++u) {
                const int e = u * 64 + lane;
                const int row = e / (D / 4), c4 = (e % (D / 4)) * 4;
                const int rr = min(r0 + row, R - 1);
                pq[u] = *reinterpret_cast<const f32x4v*>(q + ((size_t)g * R + rr) * D + c4);
                pdo[u] = *reinterpret_cast<const f32x4v*>(dout + ((size_t)g * R + rr) * D + c4);
            }
            pl = lse[(size_t)g * R + min(r0 + l31, R - 1)];
            // the tile of `out` (32 rows x D floats = 32 D / 256 one-KiB pieces) straight into the LDS; rows past the end: zeros
#pragma unroll
            for (int p = 0; p < 32 * D / 256; ++p) {
                const int row = p * (256 / D) + lane / (D / 4);
                const unsigned off = (r0 + row < R) ? (unsigned)((((size_t)g * R + r0 + row) * D + (lane % (D / 4)) * 4) * 4) : 0x80000000u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_o, (lds_void_at*)(size_t)(osLds + (unsigned)p * 1024u), 16, off, 0, 0, 0);
            }
        };
        load_q(0);
        for (int qt = 0; qt < nqt; ++qt) {
            const int r0 = qt * 32;
            // ---- registers -> the wave-private LDS tiles; delta of the 32 rows from dO (registers) and `out` (LDS, landed: vmcnt) ----
            __builtin_amdgcn_s_waitcnt(0x0f70);            // vmcnt(0): q, dO, lse and the DMA of `out`
            float dpart[NPQ];
#pragma unroll
            for (int u = 0; u < NPQ; ++u) {
                const int e = u * 64 + lane;
                const int row = e / (D / 4), c4 = (e % (D / 4)) * 4;
                *reinterpret_cast<f32x4v*>(Qs + row * ROW + c4) = pq[u];
                *reinterpret_cast<f32x4v*>(dOs + row * ROW + c4) = pdo[u];
                const f32x4v ov = *reinterpret_cast<const f32x4v*>(Os + row * D + c4);
                dpart[u] = pdo[u][0] * ov[0] + pdo[u][1] * ov[1] + pdo[u][2] * ov[2] + pdo[u][3] * ov[3];
            }
#pragma unroll
            for (int u = 0; u < NPQ; ++u) {                // a row's D / 4 pieces sit in D / 4 consecutive lanes
#pragma unroll
                for (int o = 1; o < D / 4; o <<= 1) dpart[u] += __shfl_xor(dpart[u], o, 64);
                if ((lane & (D / 4 - 1)) == 0) Dls[(u * 64 + lane) / (D / 4)] = dpart[u];
            }
            if (lane < 32) Ls[lane] = pl;
            __builtin_amdgcn_s_waitcnt(0xc07f);            // lgkmcnt(0): the reads of Os are done before the next tile's DMA may overwrite it
            if (qt + 1 < nqt) load_q(qt + 1);              // in flight during this tile's MFMAs
            const float* qp = Qs + l31 * ROW + 4 * hf;
            const float* dop = dOs + l31 * ROW + 4 * hf;
            const float Lq = Ls[l31], dlq = Dls[l31];      // this lane's query row (the columns of the transposed products)
            const int rq = r0 + l31;
            const bool rvalid = rq < R;
            const int rcq = rvalid ? rq : R - 1;
            int qiq, qhq;
            if (hs >= 0) { qiq = rcq >> hs; qhq = rcq & (h - 1); } else { qiq = rcq / h; qhq = rcq - qiq * h; }
            const float nbq = null_bias ? NBs[qhq] : 0.f;
            // ---- the null key by VALU: p0, ds0 of the 32 rows; its own gradients; dQ starts from its contribution ----
            float ds0;
            {
                float s0 = 0.f, dp0 = 0.f;
#pragma unroll
                for (int gq = 0; gq < D / 8; ++gq) {
                    const float4 a = *reinterpret_cast<const float4*>(qp + 8 * gq);
                    const float4 b = *reinterpret_cast<const float4*>(dop + 8 * gq);
                    const float4 kk = *reinterpret_cast<const float4*>(K0s + 8 * gq + 4 * hf);
                    const float4 vv = *reinterpret_cast<const float4*>(K0s + D + 8 * gq + 4 * hf);
                    s0 = fmaf(a.x, kk.x, s0); s0 = fmaf(a.y, kk.y, s0); s0 = fmaf(a.z, kk.z, s0); s0 = fmaf(a.w, kk.w, s0);
                    dp0 = fmaf(b.x, vv.x, dp0); dp0 = fmaf(b.y, vv.y, dp0); dp0 = fmaf(b.z, vv.z, dp0); dp0 = fmaf(b.w, vv.w, dp0);
                }
                s0 += __shfl_xor(s0, 32, 64);
                dp0 += __shfl_xor(dp0, 32, 64);
                const float p0 = rvalid ? __expf(s0 * scale + nbq - Lq) : 0.f;
                ds0 = p0 * (dp0 - dlq);
                if (hf == 0) dnb += ds0;
                if (lane < D) {
#pragma unroll
                    for (int rr = 0; rr < 32; ++rr) {
                        const float pb = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, p0), rr));
                        const float db = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ds0), rr));
                        dvn = fmaf(pb, dOs[rr * ROW + lane], dvn);
                        dkn = fmaf(db, Qs[rr * ROW + lane], dkn);
                    }
                }
            }
            // ---- the four score products share their operand reads: S = Q K^T, dP = dO V^T (rows = queries), S^T, dP^T (cols = queries) ----
            f32x16 s, dp, st, dpt;
#pragma unroll
            for (int i = 0; i < 16; ++i) { s[i] = 0.f; dp[i] = 0.f; st[i] = 0.f; dpt[i] = 0.f; }
            {
                f32x4a qa = *reinterpret_cast<const f32x4a*>(qp), da = *reinterpret_cast<const f32x4a*>(dop);
#pragma unroll
                for (int gq = 0; gq < D / 8; ++gq) {
                    f32x4a qn = qa, dn = da;
                    if (gq + 1 < D / 8) {
                        qn = *reinterpret_cast<const f32x4a*>(qp + 8 * (gq + 1));
                        dn = *reinterpret_cast<const f32x4a*>(dop + 8 * (gq + 1));
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        s = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[e], kreg[4 * gq + e], s, 0, 0, 0);
                        dp = __builtin_amdgcn_mfma_f32_32x32x2f32(da[e], vreg[4 * gq + e], dp, 0, 0, 0);
                        st = __builtin_amdgcn_mfma_f32_32x32x2f32(kreg[4 * gq + e], qa[e], st, 0, 0, 0);
                        dpt = __builtin_amdgcn_mfma_f32_32x32x2f32(vreg[4 * gq + e], da[e], dpt, 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
                    qa = qn; da = dn;
                }
            }
            // ---- rows = queries: P = exp(S + bias - L[row]), dS = P (dP - delta[row]) (as mqa_flash_bwd_dkv_kernel) ----
            f32x16 ds;
            {
                float lrow[16], drow[16];
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const float4 lv = *reinterpret_cast<const float4*>(Ls + 8 * q4 + 4 * hf);
                    const float4 dv = *reinterpret_cast<const float4*>(Dls + 8 * q4 + 4 * hf);
                    lrow[4 * q4] = lv.x; lrow[4 * q4 + 1] = lv.y; lrow[4 * q4 + 2] = lv.z; lrow[4 * q4 + 3] = lv.w;
                    drow[4 * q4] = dv.x; drow[4 * q4 + 1] = dv.y; drow[4 * q4 + 2] = dv.z; drow[4 * q4 + 3] = dv.w;
                }
                if (plain) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int rr = r0 + acc_row(i, hf);
                        const float ex = __expf(s[i] - lrow[i]);
                        const float p = (!jvalid || rr >= R) ? 0.f : ex;
                        s[i] = p;
                        ds[i] = p * (dp[i] - drow[i]);
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int rr = r0 + acc_row(i, hf);
                        const int rcl = min(rr, R - 1);
                        int qi, qh;
                        if (hs >= 0) { qi = rcl >> hs; qh = rcl & (h - 1); } else { qi = rcl / h; qh = rcl - qi * h; }
                        const float rvv = rel ? Rs[attn_rel_index(j, E, ns, h, qi, qh)] : 0.f;
                        const float sv = attn_bias_apply(s[i], j, M, E, qi, causal, rel != nullptr, rvv, 0.f);
                        const float ex = __expf(sv - lrow[i]);
                        const float p = (sv == -INFINITY || rr >= R) ? 0.f : ex;
                        s[i] = p;
                        ds[i] = p * (dp[i] - drow[i]);
                    }
                }
            }
            // ---- dV^T += dO^T P, dK^T += Q^T dS ----
            {
                float oc[ND], qc[ND], on[ND], qn2[ND];
#pragma unroll
                for (int c = 0; c < ND; ++c) { oc[c] = dOs[acc_row(0, hf) * ROW + l31 + 32 * c]; qc[c] = Qs[acc_row(0, hf) * ROW + l31 + 32 * c]; }
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    if (i + 1 < 16) {
                        const int row = acc_row(i + 1, hf);
#pragma unroll
                        for (int c = 0; c < ND; ++c) { on[c] = dOs[row * ROW + l31 + 32 * c]; qn2[c] = Qs[row * ROW + l31 + 32 * c]; }
                    }
#pragma unroll
                    for (int c = 0; c < ND; ++c) {
                        dvt[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(oc[c], s[i], dvt[c], 0, 0, 0);
                        dkt[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(qc[c], ds[i], dkt[c], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x100, 2 * ND, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 2 * ND, 0);
#pragma unroll
                    for (int c = 0; c < ND; ++c) { oc[c] = on[c]; qc[c] = qn2[c]; }
                }
            }
            // ---- cols = queries: dS^T = exp(S^T + bias - L) (dP^T - delta) for this lane's query; bias-gradient table; dQ^T ----
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int jj = E + acc_row(i, hf);
                const float rvv = rel ? Rs[attn_rel_index(jj, E, ns, h, qiq, qhq)] : 0.f;
                const float sv = attn_bias_apply(st[i], jj, M, E, qiq, causal, rel != nullptr, rvv, 0.f);
                const float p = (sv == -INFINITY) ? 0.f : __expf(sv - Lq);
                st[i] = rvalid ? p * (dpt[i] - dlq) : 0.f;
            }
            if (rel) {                                     // kernel-uniform; the lane halves (keys 4 apart) update one after the other
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    if (hf == half && rvalid) {
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            const int jj = E + acc_row(i, hf);
                            if (jj < M) {
                                const int idx = max(0, min(qiq - (jj - E) + ns - 1, 2 * ns - 2));
                                tbl[idx * h + qhq] += st[i];
                            }
                        }
                    }
                }
            }
            f32x16 dqt[ND];
#pragma unroll
            for (int c = 0; c < ND; ++c)
#pragma unroll
                for (int i = 0; i < 16; ++i) dqt[c][i] = ds0 * K0s[32 * c + acc_row(i, hf)];
            {
                float kc[ND], kn2[ND];
#pragma unroll
                for (int c = 0; c < ND; ++c) kc[c] = Ks[acc_row(0, hf) * ROW + l31 + 32 * c];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    if (i + 1 < 16) {
                        const float* kq = Ks + acc_row(i + 1, hf) * ROW + l31;
#pragma unroll
                        for (int c = 0; c < ND; ++c) kn2[c] = kq[32 * c];
                    }
#pragma unroll
                    for (int c = 0; c < ND; ++c) dqt[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(kc[c], st[i], dqt[c], 0, 0, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, ND, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, ND, 0);
#pragma unroll
                    for (int c = 0; c < ND; ++c) kc[c] = kn2[c];
                }
            }
            if (rvalid) {
                float* dqg = dq + ((size_t)g * R + rq) * D;
#pragma unroll
                for (int c = 0; c < ND; ++c)
#pragma unroll
                    for (int i = 0; i < 16; i += 4) {
                        const int dd = 32 * c + 8 * (i >> 2) + 4 * hf;
                        *reinterpret_cast<float4*>(dqg + dd) = make_float4(dqt[c][i] * scale, dqt[c][i + 1] * scale, dqt[c][i + 2] * scale, dqt[c][i + 3] * scale);
                    }
            }
        }
        // ---- dK, dV of the sequence (K carried the scale for S: the gradient takes it once more), the null row ----
        if (jvalid) {
            float* og = dkv + ((size_t)g * M + j) * 2 * D;
#pragma unroll
            for (int c = 0; c < ND; ++c)
#pragma unroll
                for (int i = 0; i < 16; i += 4) {
                    const int dd = 32 * c + 8 * (i >> 2) + 4 * hf;
                    *reinterpret_cast<float4*>(og + dd) = make_float4(dkt[c][i] * scale, dkt[c][i + 1] * scale, dkt[c][i + 2] * scale, dkt[c][i + 3] * scale);
                    *reinterpret_cast<float4*>(og + D + dd) = make_float4(dvt[c][i], dvt[c][i + 1], dvt[c][i + 2], dvt[c][i + 3]);
                }
        }
        if (lane < D) {
            float* og = dkv + (size_t)g * M * 2 * D;
            og[lane] = dkn * scale;
            og[D + lane] = dvn;
        }
    }
    // ---- bias-gradient partials of this wave: one row of the workspace each ----
    const size_t wrow = (size_t)blockIdx.x * 4 + wave;
    if (TBL) {
        __builtin_amdgcn_s_waitcnt(0xc07f);
        for (int e = lane; e < TBL; e += 64) tbl_part[wrow * TBL + e] = tbl[e];
    }
    if (dnull_part && hf == 0) dnull_part[wrow * 32 + l31] = dnb;
}

static int attn_bwd_rows(int G, int n, int h) {
    const int gx = (int)(((long long)n * h + AQ - 1) / AQ), gy = G < 256 ? G : 256;
    return gy * gx * 4;
}

extern "C" size_t diqt_mqa_attention_bwd_workspace_bytes(int G, int n, int h, int d, int n_extra, int n_self, int has_rel) {
    if (G <= 0 || n <= 0 || h <= 0 || n_self < 0) return 0;
    const size_t rows = (size_t)attn_bwd_rows(G, n, h);
    const size_t tbl = has_rel ? (size_t)(2 * n_self - 1) * h : 0;
    return ((size_t)G * n * h + rows * tbl + rows * 32) * sizeof(float);
}

// Backward of diqt_mqa_attention_fwd_lse.  dq: [G, n*h, d]; dkv: [G, n_extra + n_self, 2d] (every row written); drel: [2 n_self - 1, h]
// or NULL; dnull: [h] or NULL (both only with rel / null_bias given).  workspace: diqt_mqa_attention_bwd_workspace_bytes.
extern "C" int diqt_mqa_attention_bwd(const float* q, const float* kv, const float* rel, const float* null_bias, const float* out,
                                      const float* dout, const float* lse, float* dq, float* dkv, float* drel, float* dnull,
                                      void* workspace, size_t workspace_bytes, int G, int n, int h, int d, int n_extra, int n_self,
                                      int causal, float scale, void* stream) {
    DIQT_REQUIRE(q && kv && out && dout && lse && dq && dkv && workspace, DIQT_E_ALIGN, "mqa_attention_bwd: null pointer");
    DIQT_REQUIRE(G > 0 && n > 0 && h > 0 && n_extra >= 0 && n_self >= 0 && n_extra + n_self > 0, DIQT_E_SHAPE, "mqa_attention_bwd: bad shape");
    DIQT_REQUIRE(d == 32 || d == 64, DIQT_E_UNSUPPORTED, "mqa_attention_bwd: dim_head %d (32 or 64 are built)", d);
    DIQT_REQUIRE(!(causal || rel) || n_self == n, DIQT_E_SHAPE, "mqa_attention_bwd: causal / relative bias need n_self == n");
    DIQT_REQUIRE(!null_bias || n_extra >= 1, DIQT_E_SHAPE, "mqa_attention_bwd: null bias without a null key");
    DIQT_REQUIRE(!rel || drel, DIQT_E_ALIGN, "mqa_attention_bwd: relative bias without a gradient buffer");
    DIQT_REQUIRE(!null_bias || dnull, DIQT_E_ALIGN, "mqa_attention_bwd: null bias without a gradient buffer");
    DIQT_REQUIRE(aligned16(q) && aligned16(kv) && aligned16(out) && aligned16(dout) && aligned16(dq) && aligned16(dkv) && aligned16(workspace),
                 DIQT_E_ALIGN, "mqa_attention_bwd: pointers must be 16-byte aligned");
    DIQT_REQUIRE(G <= 65535, DIQT_E_SHAPE, "mqa_attention_bwd: G > 65535");
    const size_t need = diqt_mqa_attention_bwd_workspace_bytes(G, n, h, d, n_extra, n_self, rel ? 1 : 0);
    DIQT_REQUIRE(workspace_bytes >= need, DIQT_E_WORKSPACE, "mqa_attention_bwd: workspace %zu < %zu", workspace_bytes, need);
    const int TBL = rel ? (2 * n_self - 1) * h : 0;
    DIQT_REQUIRE((size_t)4 * TBL * sizeof(float) <= 24 * 1024, DIQT_E_UNSUPPORTED,
                 "mqa_attention_bwd: relative-bias table of %d entries per wave does not fit the LDS budget", TBL);
    hipStream_t s = (hipStream_t)stream;
    const int R = n * h, M = n_extra + n_self;
    const int gx = (R + AQ - 1) / AQ, gy = G < 256 ? G : 256, rows = gy * gx * 4;
    float* delta = static_cast<float*>(workspace);
    float* tbl_part = delta + (size_t)G * R;
    float* dnull_part = tbl_part + (size_t)rows * TBL;
    {
        // thousands of short sequences with the lone null key (the temporal attentions): everything in one pass, a sequence per wave
        static const bool noSeq = [] { const char* e = getenv("DIQT_ATTN_NO_SEQ"); return e && e[0] == '1'; }();
        const unsigned long long ob = (unsigned long long)G * R * d * 4ull;
        const int relLds = rel ? TBL : 0;
        const int perw = (32 * d + 3 * 32 * (d + 4) + 64 + 2 * d + TBL + 255) / 256 * 256;
        const size_t lds = ((size_t)4 * perw + relLds + 64) * sizeof(float);
        if (!noSeq && n_extra == 1 && n_self == n && n <= 32 && 32 % h == 0 && G >= 512 && ob < (1ull << 31) && lds <= 160 * 1024) {
            const int nwg = (G + 3) / 4 < 256 ? (G + 3) / 4 : 256;
            auto kern = d == 64 ? mqa_seq_bwd_kernel<2> : mqa_seq_bwd_kernel<1>;
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            DIQT_REQUIRE(e == hipSuccess, DIQT_E_LAUNCH, "mqa_attention_bwd(seq): hipFuncSetAttribute: %s", hipGetErrorString(e));
            float* dnp = tbl_part + (size_t)nwg * 4 * TBL;                          // (nwg * 4 <= rows: fits the workspace of the two-kernel path)
            hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(256), lds, s, q, kv, rel, null_bias, out, dout, lse, dq, dkv, tbl_part,
                               null_bias ? dnp : (float*)nullptr, n, h, causal, scale, G, relLds, (unsigned)ob);
            int rc = check_launch("mqa_attention_bwd(seq)");
            if (rc) return rc;
            if (rel || null_bias) {
                const int ntb = rel ? (TBL + 63) / 64 : 0;
                const size_t lds_r = (size_t)(256 * h > 256 ? 256 * h : 256) * sizeof(float);
                // lane l of a wave's row always carries head l % h here (32 % h == 0): gx = 1 and an unbounded R give exactly that mapping
                hipLaunchKernelGGL(attn_bias_reduce_kernel, dim3((unsigned)(ntb + (null_bias ? 1 : 0))), dim3(256), lds_r, s, tbl_part, dnp,
                                   rel ? drel : (float*)nullptr, null_bias ? dnull : (float*)nullptr, nwg * 4, TBL, 1, h, 0x7fffffff, ntb);
                rc = check_launch("mqa_attention_bwd(seq, bias reduce)");
                if (rc) return rc;
            }
            return DIQT_OK;
        }
    }
    {
        const dim3 grid(gx, gy);
        const size_t lds = (size_t)4 * TBL * sizeof(float);
        if (d == 64)
            hipLaunchKernelGGL(mqa_flash_bwd_dq_kernel<2>, grid, dim3(256), lds, s, q, kv, rel, null_bias, out, dout, lse, dq, delta, tbl_part,
                               null_bias ? dnull_part : (float*)nullptr, G, n, h, n_extra, n_self, causal, scale);
        else
            hipLaunchKernelGGL(mqa_flash_bwd_dq_kernel<1>, grid, dim3(256), lds, s, q, kv, rel, null_bias, out, dout, lse, dq, delta, tbl_part,
                               null_bias ? dnull_part : (float*)nullptr, G, n, h, n_extra, n_self, causal, scale);
        int rc = check_launch("mqa_attention_bwd(dq)");
        if (rc) return rc;
    }
    if (rel || null_bias) {
        const int ntb = rel ? (TBL + 63) / 64 : 0;
        const size_t lds_r = (size_t)(256 * h > 256 ? 256 * h : 256) * sizeof(float);
        hipLaunchKernelGGL(attn_bias_reduce_kernel, dim3((unsigned)(ntb + (null_bias ? 1 : 0))), dim3(256), lds_r, s, tbl_part, dnull_part,
                           rel ? drel : (float*)nullptr, null_bias ? dnull : (float*)nullptr, rows, TBL, gx, h, R, ntb);
        int rc = check_launch("mqa_attention_bwd(bias reduce)");
        if (rc) return rc;
    }
    {
        // keys that go through the MFMA tiles: a lone null key, or a few extra keys in front of at least as many self-key tiles, are VALU work
        const int EV = (n_extra == 1 || (n_extra >= 1 && n_extra <= 8 && n_extra <= (n_self + 31) / 32)) ? n_extra : 0;
        const int Mt = M - EV;
        // key tiles per workgroup; the other 4 / KW waves split the query tiles.  Few batch entries (the joint 2048-token attentions:
        // G = 8) need the finer split to fill 256 CUs: every workgroup walks ALL query rows of its batch entry.
        int KW = Mt <= 32 ? 1 : (Mt <= 64 ? 2 : 4);
        while (KW > 1 && (long long)((Mt + 32 * KW - 1) / (32 * KW)) * G < 512) KW >>= 1;
        static const bool noPerWave = [] { const char* e = getenv("DIQT_ATTN_NO_PERWAVE"); return e && e[0] == '1'; }();
        const bool perWave = !noPerWave && Mt <= 32 && G >= 2048;      // one sequence per wave (see the kernel)
        if (perWave) KW = 0;
        const int nkt = perWave ? 1 : (Mt > 0 ? (Mt + 32 * KW - 1) / (32 * KW) : 1);
        const dim3 grid((unsigned)nkt, perWave ? (unsigned)((G + 3) / 4) : (unsigned)G);
        const int ROW = d + 4;
        size_t lds = (size_t)4 * (2 * 32 * ROW + 128 + 2 * d) * sizeof(float);
        const int relLds = (rel && h <= 64 && (2 * n_self - 1) * h <= 4096) ? (2 * n_self - 1) * h : 0;      // bias tables in LDS (<= 16 KB)
        const size_t red = (size_t)(perWave ? 0 : 4 - KW) * (2 * (d / 32) * 16 * 64 + 128) * sizeof(float);
        DIQT_REQUIRE(red <= lds, DIQT_E_UNSUPPORTED, "mqa_attention_bwd: combine region larger than the staging regions");
        lds += (size_t)(relLds + 64) * sizeof(float);          // bias tables behind the staging regions
        auto kern = d == 64 ? mqa_flash_bwd_dkv_kernel<2> : mqa_flash_bwd_dkv_kernel<1>;
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            DIQT_REQUIRE(e == hipSuccess, DIQT_E_LAUNCH, "mqa_attention_bwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
        }
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, q, kv, rel, null_bias, dout, lse, delta, dkv, n, h, n_extra, n_self, causal, scale, KW, G, relLds, EV);
        return check_launch("mqa_attention_bwd(dkv)");
    }
}
