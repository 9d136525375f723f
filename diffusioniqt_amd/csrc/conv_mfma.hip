// Implicit-GEMM 3-D convolution for gfx950 on v_mfma_f32_32x32x2_f32 (exact fp32).
//
// Forward / backward-data kernel (conv_fwd_kernel):
//   GEMM view  M = B*Do*Ho*Wo voxels, N = Cout, K = taps*Cin.  One 256-thread workgroup (4 waves)
//   owns a 128-voxel spatial tile (TD x TH x TW) x 64 output channels.  The input HALO tile
//   (TD+kd-1)(TH+kh-1)(TW+kw-1) x 32 channels is staged ONCE per 32-channel chunk into LDS and reused
//   by all taps (no im2col, each input voxel is read from HBM/L2 once per chunk and tile);  the
//   64x32 weight panel of the current tap is double-buffered in LDS and prefetched through registers
//   one tap ahead.  Each wave computes 32 voxels x 64 channels = two 32x32 accumulators.
//   K order inside an 8-wide group is permuted (lane half h takes k = 8q+4h..+3) so that both
//   operands are read with ds_read_b128 along the channel axis; A and B use the same permutation.
//   LDS rows are padded to 36 floats: 36*i mod 64 hits all sixteen 16-B slots -> conflict-free
//   weight reads, <=3-way on the strided halo reads (1 read per 8 MFMAs: irrelevant at the f32 rate).
//   ~76 KB LDS for a 3x3x3 filter -> 2 workgroups per CU, so one workgroup's halo staging overlaps
//   the other's MFMA phase.  Block ids are remapped so each XCD gets a contiguous run of tiles.
//
// Backward-weight kernel (conv_bwd_weight_kernel):
//   dW[co][ci][tap] = sum_v dY[v][co] * X[v+tap][ci]:  M = co (64/WG), N = ci (32/WG), K = voxels.
//   A workgroup owns one kd-plane x one tap group (<=10 (kh,kw) taps) x 32 ci x 64 co and walks a
//   contiguous range of voxel tiles (split-K); per tile it stages the X halo plane and the dY tile in
//   LDS and issues one MFMA per (tap, 2 voxels).  Partial slabs go to the workspace in the packed
//   layout and are reduced in a fixed order (deterministic) by conv_reduce_dw_kernel.
#include "common.h"
#include "conv_wgrad.h"
#include "conv_wgrad_h.h"
#include "conv_pw.h"
#include "conv_fwd9.h"
#include <type_traits>
#include <stdlib.h>
#include <atomic>

namespace diqt {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int CK = 32;          // channels per K-chunk
constexpr int LDSROW = CK + 4;  // padded LDS row (floats)
constexpr int NT = 64;          // output channels per workgroup
constexpr int MTILE = 128;      // output voxels per workgroup

struct ConvGeom {
    int B, D, H, W, Cin, Cout;
    int Do, Ho, Wo;
    int kd, kh, kw, pd, ph, pw;
    int TD, TH, TW;
    int tilesD, tilesH, tilesW;
    int nNt, nChunks, CoutPad;
    int HD, HH, HWd;            // halo extents
    unsigned xBytes, yBytes;    // tensor extents for the buffer descriptors (BUF kernels; 0 when >= 1 GiB)
    float* stats;               // optional per-tile column sums of the output: [B][tiles per batch][2 (sum, sum of squares)][Cout]
    int tilesPerWg;             // forward kernel: consecutive tiles per workgroup
    int stagger;                // experiment: first-round workgroups sleep (slot % stagger) x ~6.4k cycles before starting
    int chunksPerSplit;         // forward split-K over input-channel chunks (grid.y slices; == nChunks when unsplit)
    unsigned long long slabStride;   // floats between the split-K output slabs
    unsigned long long* dbg;    // diagnostic cycle stamps (NULL in production)
    int subF;                   // > 0: the batch is a cube of subF^3 sub-volumes and halo voxels are read from the NEIGHBOUR sub-volume
};

__host__ __device__ inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// Source voxel (index into x[B][D][H][W]) of input coordinate (iz, iy, ix) of batch entry b, or -1 for zero padding.
// Neighbour mode (g.subF = f > 0; `boundary=True` of the reference, imagen_pytorch3D.py:37-46 boundary_pad): the batch holds the
// f^3 sub-volumes of one merged (fA)^3 volume, entry n = b2 + f b3 + f^2 b4 with axis D <-> b2, H <-> b3, W <-> b4
// (utils_mine.py:25-67); a coordinate one voxel outside a sub-volume is the edge voxel of its neighbour, and only the faces of the
// merged volume are zero padded -- the conv reads what merge -> zero-pad -> overlapping split would have copied, without the copies.
__device__ __forceinline__ int conv_src_voxel(const ConvGeom& g, int b, int iz, int iy, int ix) {
    if (g.subF > 0) {
        const int f = g.subF, A = g.D, S = f * A;
        const int gz = (b % f) * A + iz, gy = ((b / f) % f) * A + iy, gx = (b / (f * f)) * A + ix;
        if ((unsigned)gz >= (unsigned)S || (unsigned)gy >= (unsigned)S || (unsigned)gx >= (unsigned)S) return -1;
        const int nb = gz / A + f * (gy / A) + f * f * (gx / A);
        return ((nb * A + gz % A) * A + gy % A) * A + gx % A;
    }
    if (iz < 0 || iz >= g.D || iy < 0 || iy >= g.H || ix < 0 || ix >= g.W) return -1;
    return ((b * g.D + iz) * g.H + iy) * g.W + ix;
}

// ---------------------------------------------------------------------------------------------
// weight packing: packed[((chunk*T + tap)*CoutPad + co)*32 + k]
// ---------------------------------------------------------------------------------------------
__global__ void conv_pack_weight_kernel(const float* __restrict__ w, float* __restrict__ packed,
                                        int Cout, int Cin, int T, int mode, int CoutPadEff,
                                        int nChunksEff, size_t total) {
    // mode 0: effective (out,in) = (Cout,Cin), value w[co][ci][tap]
    // mode 1: effective (out,in) = (Cin,Cout), value w[in_eff][out_eff][T-1-tap]
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int k = (int)(i % CK);
        size_t r = i / CK;
        const int o = (int)(r % CoutPadEff);
        r /= CoutPadEff;
        const int tap = (int)(r % T);
        const int chunk = (int)(r / T);
        const int in = chunk * CK + k;
        float v = 0.f;
        if (mode == 0) {
            if (o < Cout && in < Cin) v = w[((size_t)o * Cin + in) * T + tap];
        } else {
            if (o < Cin && in < Cout) v = w[((size_t)in * Cin + o) * T + (T - 1 - tap)];
        }
        packed[i] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// forward / backward-data
// ---------------------------------------------------------------------------------------------
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// buffer offsets past every descriptor's range (tensors are < 1 GiB on this path): loads return 0, stores are dropped.
// voxel/row offsets use BUF_OOB, channel offsets BUF_OOB_C, so that any sum of the two stays out of range without wrapping
constexpr unsigned BUF_OOB = 0x80000000u, BUF_OOB_C = 0x40000000u;

// BUF = true (VEC4 and tensors < 1 GiB): halo staging and the epilogue go through buffer descriptors -- the per-tile
// tables hold byte offsets (or BUF_OOB for padding voxels / rows outside the output), the hardware range check supplies
// the zeros and drops the masked stores, and a piece costs ~6 instructions instead of ~25.  These phases share a SIMD
// with the co-resident workgroup's MFMA stream, which stretches every non-MFMA instruction ~3x (profiles/r01_conv_ablation.md).
// CKT = channels per staged K-chunk: 32 (78 KB of LDS for a 3x3x3 filter, 2 workgroups per CU) or 16 (44 KB, 3 per CU: a third
// resident workgroup keeps two waves per SIMD on the MFMA pipe while one stages or stores).  The packed weight layout is the
// 32-wide one either way; a 16-wide sub-chunk reads half rows of it.
template <bool VEC4, bool BUF, int CKT>
__global__ __launch_bounds__(256, CKT == 16 ? 3 : 2) void conv_fwd_kernel(const float* __restrict__ x,
                                                          const float* __restrict__ wp,
                                                          const float* __restrict__ bias,
                                                          const float* __restrict__ residual,
                                                          float* __restrict__ y, ConvGeom g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int LROW = CKT + 4, PCS = CKT / 4, PSH = CKT == 32 ? 3 : 2, SUB = CK / CKT;   // padded row, 16-B pieces per row, log2(PCS), sub-chunks per packed chunk
    const int HV = g.HD * g.HH * g.HWd;
    float* halo = smem;                                  // [HV][LROW]
    float* wbuf = smem + (size_t)HV * LROW;            // [2][64][LROW]
    int* out_off = reinterpret_cast<int*>(wbuf + 2 * NT * LROW);   // [128] voxel -> output row or -1
    int* halo_src = out_off + MTILE;                                 // [HV] halo voxel -> input voxel index or -1

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;

    const unsigned nwg = gridDim.x;
    if (g.stagger > 1 && blockIdx.x < 512) {
        const int nsl = __builtin_amdgcn_readfirstlane((int)((blockIdx.x / 8) % g.stagger));
        for (int i = 0; i < nsl; ++i) __builtin_amdgcn_s_sleep(100);
    }
    // a workgroup walks g.tilesPerWg consecutive tiles (1 by default; DIQT_CONV_TPW for the turn-over experiment)
    const unsigned totalTiles = (unsigned)g.B * g.tilesD * g.tilesH * g.tilesW * g.nNt;
    unsigned long long stamps[8];
    int nst = 0;
#define DIQT_STAMP() do { if (g.dbg && nst < 8) stamps[nst++] = __builtin_readcyclecounter(); } while (0)
    const unsigned long long rt0 = g.dbg ? __builtin_amdgcn_s_memrealtime() : 0ull;
    y += (size_t)blockIdx.y * g.slabStride;   // split-K slab (0 when unsplit)
    for (int rep = 0; rep < g.tilesPerWg; ++rep) {
    const unsigned L = xcd_remap(blockIdx.x, nwg) * g.tilesPerWg + rep;
    if (L >= totalTiles) break;
    if (rep) __syncthreads();       // the previous tile's epilogue is done with out_off
    const int nt = L % g.nNt;
    int mt = L / g.nNt;
    const int tx = mt % g.tilesW; mt /= g.tilesW;
    const int ty = mt % g.tilesH; mt /= g.tilesH;
    const int tz = mt % g.tilesD;
    const int b = mt / g.tilesD;
    const int d0 = tz * g.TD, h0 = ty * g.TH, w0 = tx * g.TW;
    const int n0 = nt * NT;
    const int T = g.kd * g.kh * g.kw;

    if (tid < MTILE) {
        const int tw = tid % g.TW, th = (tid / g.TW) % g.TH, td = tid / (g.TW * g.TH);
        const int od = d0 + td, oh = h0 + th, ow = w0 + tw;
        int off = BUF ? (int)BUF_OOB : -1;
        if (od < g.Do && oh < g.Ho && ow < g.Wo) {
            off = ((b * g.Do + od) * g.Ho + oh) * g.Wo + ow;
            if (BUF) off *= g.Cout * 4;          // byte offset of the output row
        }
        out_off[tid] = off;
    }

    // halo voxel -> source voxel, computed once per tile (keeps the integer divisions out of the staging loops)
    for (int hv = tid; hv < HV; hv += 256) {
        const int hx = hv % g.HWd, hy = (hv / g.HWd) % g.HH, hz = hv / (g.HWd * g.HH);
        const int iz = d0 + hz - g.pd, iy = h0 + hy - g.ph, ix = w0 + hx - g.pw;
        int src = conv_src_voxel(g, b, iz, iy, ix);
        if (BUF) src = src < 0 ? (int)BUF_OOB : src * g.Cin * 4;           // byte offset of the input voxel's channel row
        halo_src[hv] = src;
    }
    // buffer descriptors (wave-uniform inputs only); the slab base of a split-K launch is folded into the y descriptor below
    const auto rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, BUF ? (int)g.xBytes : 0, 0x00020000);

    // this lane's voxel (row of the A operand) -> halo index at tap (0,0,0)
    int hidx_lane;
    {
        const int v = wave * 32 + l31;
        const int tw = v % g.TW, th = (v / g.TW) % g.TH, td = v / (g.TW * g.TH);
        hidx_lane = (td * g.HH + th) * g.HWd + tw;
    }

    f32x16 acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
    DIQT_STAMP();

    const int wrow = tid >> PSH, wc4 = (tid & (PCS - 1)) * 4;      // weight staging: rows wrow (and wrow+32 when CKT == 32)

    // split-K (small spatial extents): grid.y slices the input-channel chunks; each slice writes its own output slab
    const int chunkBeg = blockIdx.y * g.chunksPerSplit * SUB, chunkEnd = min(g.nChunks, (int)(blockIdx.y + 1) * g.chunksPerSplit) * SUB;
    for (int chunk = chunkBeg; chunk < chunkEnd; ++chunk) {
        const int ci0 = chunk * CKT;
        if (ci0 >= g.Cin) break;      // second half of a ragged last chunk (block-uniform)
        __syncthreads();   // all reads of the previous chunk's halo and of both weight buffers are done
                if (chunk == chunkBeg + 1) DIQT_STAMP();
        // ---- stage halo chunk: loads are UNCONDITIONAL (clamped address, zero-selected afterwards) and issued in batches
        //      of 8 before any LDS store, so a batch costs one memory round trip instead of eight serialized ones ----
        if (BUF) {
            const unsigned coff = (ci0 + wc4 < g.Cin) ? (unsigned)(ci0 + wc4) * 4u : BUF_OOB_C;   // this thread's channel quad
            for (int base = 0; base < HV * PCS; base += 256 * 8) {
                u32x4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int idx = base + u * 256 + tid;
                    // unconditional table read (clamped index) + select: a guarded read becomes an exec-masked branch
                    // with its own ds_read -> s_waitcnt round trip per piece
                    const unsigned t = (unsigned)halo_src[min(idx >> PSH, HV - 1)] + coff;
                    const unsigned voff = (idx < HV * PCS) ? t : BUF_OOB;
                    v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, voff, 0, 0);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int idx = base + u * 256 + tid;
                    if (idx < HV * PCS) *reinterpret_cast<u32x4*>(halo + (idx >> PSH) * LROW + wc4) = v[u];
                }
            }
        } else
        for (int base = 0; base < HV * PCS; base += 256 * 8) {
            float4 v[8];
            bool ok[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = base + u * 256 + tid;
                const int hv = (idx < HV * PCS) ? (idx >> PSH) : 0, c4 = (idx & (PCS - 1)) * 4;
                const int src = halo_src[hv];
                ok[u] = idx < HV * PCS && src >= 0 && ci0 + c4 < g.Cin;
                const size_t off = ok[u] ? (size_t)src * g.Cin + ci0 + c4 : 0;
                if (VEC4) {
                    v[u] = *reinterpret_cast<const float4*>(x + off);
                } else {
                    const int rem = ok[u] ? g.Cin - (ci0 + c4) : 0;
                    v[u].x = x[off];
                    v[u].y = rem > 1 ? x[off + 1] : 0.f;
                    v[u].z = rem > 2 ? x[off + 2] : 0.f;
                    v[u].w = rem > 3 ? x[off + 3] : 0.f;
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = base + u * 256 + tid;
                if (idx < HV * PCS)
                    *reinterpret_cast<float4*>(halo + (idx >> PSH) * LROW + (idx & (PCS - 1)) * 4) =
                        ok[u] ? v[u] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        if (chunk == chunkBeg + 1) DIQT_STAMP();
        // ---- weights: panel of tap 0 -> LDS now; panel t+1 is written at the START of tap t (its buffer was last read in
        //      tap t-1, retired by the barrier) from registers loaded during tap t-1, so neither the global latency nor the
        //      LDS write sits between the last MFMA of a tap and its barrier ----
        const float* wchunk = wp + ((size_t)(chunk / SUB) * T * g.CoutPad + n0) * CK + (chunk % SUB) * CKT;
        float4 r0, r1;
        {
            const float4 p0 = *reinterpret_cast<const float4*>(wchunk + (size_t)wrow * CK + wc4);
            *reinterpret_cast<float4*>(wbuf + wrow * LROW + wc4) = p0;
            if (CKT == 32) *reinterpret_cast<float4*>(wbuf + (wrow + 32) * LROW + wc4) = *reinterpret_cast<const float4*>(wchunk + (size_t)(wrow + 32) * CK + wc4);
            if (T > 1) {
                const float* wt = wchunk + (size_t)g.CoutPad * CK;
                r0 = *reinterpret_cast<const float4*>(wt + (size_t)wrow * CK + wc4);
                if (CKT == 32) r1 = *reinterpret_cast<const float4*>(wt + (size_t)(wrow + 32) * CK + wc4);
            }
        }
        __syncthreads();
        DIQT_STAMP();

        int tap = 0;
        for (int kz = 0; kz < g.kd; ++kz)
            for (int ky = 0; ky < g.kh; ++ky)
                for (int kx = 0; kx < g.kw; ++kx, ++tap) {
                    if (tap + 1 < T) {
                        float* wnext = wbuf + ((tap + 1) & 1) * (NT * LROW);
                        *reinterpret_cast<float4*>(wnext + wrow * LROW + wc4) = r0;
                        if (CKT == 32) *reinterpret_cast<float4*>(wnext + (wrow + 32) * LROW + wc4) = r1;
                    }
                    if (tap + 2 < T) {
                        const float* wt = wchunk + (size_t)(tap + 2) * g.CoutPad * CK;
                        r0 = *reinterpret_cast<const float4*>(wt + (size_t)wrow * CK + wc4);
                        if (CKT == 32) r1 = *reinterpret_cast<const float4*>(wt + (size_t)(wrow + 32) * CK + wc4);
                    }
                    const float* wcur = wbuf + (tap & 1) * (NT * LROW);
                    const float* ap = halo + (hidx_lane + (kz * g.HH + ky) * g.HWd + kx) * LROW + 4 * h;
                    const float* bp = wcur + l31 * LROW + 4 * h;
                    // software-pipelined over q: the fragments of q+1 are in flight while the 8 MFMAs of q issue
                    float4 a = *reinterpret_cast<const float4*>(ap);
                    float4 b0 = *reinterpret_cast<const float4*>(bp);
                    float4 b1 = *reinterpret_cast<const float4*>(bp + 32 * LROW);
#pragma unroll
                    for (int q = 0; q < CKT / 8; ++q) {
                        float4 an, b0n, b1n;
                        if (q < CKT / 8 - 1) {
                            an = *reinterpret_cast<const float4*>(ap + 8 * (q + 1));
                            b0n = *reinterpret_cast<const float4*>(bp + 8 * (q + 1));
                            b1n = *reinterpret_cast<const float4*>(bp + 32 * LROW + 8 * (q + 1));
                        }
                        __builtin_amdgcn_sched_barrier(0);   // keep the q+1 reads ahead of q's MFMAs (hipcc sinks them otherwise)
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b0.x, acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b1.x, acc1, 0, 0, 0);
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b0.y, acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b1.y, acc1, 0, 0, 0);
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b0.z, acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b1.z, acc1, 0, 0, 0);
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b0.w, acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b1.w, acc1, 0, 0, 0);
                        if (q < CKT / 8 - 1) { a = an; b0 = b0n; b1 = b1n; }
                    }
                    __syncthreads();
                }
        DIQT_STAMP();
    }

    // ---- epilogue: D[row=voxel][col=co]; row = (r&3) + 8*(r>>2) + 4*h ----
    const int co0 = n0 + l31, co1 = n0 + 32 + l31;
    const float bias0 = (bias && co0 < g.Cout) ? bias[co0] : 0.f;
    const float bias1 = (bias && co1 < g.Cout) ? bias[co1] : 0.f;
    if (BUF) {
        const auto rs_y = __builtin_amdgcn_make_buffer_rsrc(y, 0, (int)g.yBytes, 0x00020000);
        const auto rs_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(residual), 0, residual ? (int)g.yBytes : 0, 0x00020000);
        const unsigned c0 = co0 < g.Cout ? (unsigned)co0 * 4u : BUF_OOB_C, c1 = co1 < g.Cout ? (unsigned)co1 * 4u : BUF_OOB_C;
        float cs0 = 0.f, cq0 = 0.f, cs1 = 0.f, cq1 = 0.f;      // column sums for the consumer's GroupNorm / SE pooling
        float rr0[16], rr1[16];
        if (residual) {            // wave-uniform: all 32 loads in flight before the first add
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const unsigned off = (unsigned)out_off[wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h];
                rr0[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_r, off + c0, 0, 0));
                rr1[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_r, off + c1, 0, 0));
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const unsigned off = (unsigned)out_off[wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h];
            float v0 = acc0[r] + bias0, v1 = acc1[r] + bias1;
            if (residual) { v0 += rr0[r]; v1 += rr1[r]; }
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v0), rs_y, off + c0, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v1), rs_y, off + c1, 0, 0);
            if (g.stats && off != BUF_OOB) { cs0 += v0; cq0 = fmaf(v0, v0, cq0); cs1 += v1; cq1 = fmaf(v1, v1, cq1); }
        }
        if (g.stats) {             // block-uniform: fixed-order combine of lane halves, then of the 4 waves through LDS
            cs0 += __shfl_xor(cs0, 32, 64); cq0 += __shfl_xor(cq0, 32, 64);
            cs1 += __shfl_xor(cs1, 32, 64); cq1 += __shfl_xor(cq1, 32, 64);
            float* red = halo;     // every wave passed the last tap barrier: the halo image is dead
            __syncthreads();
            if (h == 0) {
                red[(wave * 4 + 0) * 32 + l31] = cs0; red[(wave * 4 + 1) * 32 + l31] = cq0;
                red[(wave * 4 + 2) * 32 + l31] = cs1; red[(wave * 4 + 3) * 32 + l31] = cq1;
            }
            __syncthreads();
            if (tid < 128) {       // q = tid >> 5: 0 sum(co0) 1 sumsq(co0) 2 sum(co1) 3 sumsq(co1)
                const int q = tid >> 5, l = tid & 31;
                const float v = ((red[(0 * 4 + q) * 32 + l] + red[(1 * 4 + q) * 32 + l]) + red[(2 * 4 + q) * 32 + l]) + red[(3 * 4 + q) * 32 + l];
                const int co = n0 + (q >> 1) * 32 + l;
                const int tpb = g.tilesD * g.tilesH * g.tilesW, tIn = (tz * g.tilesH + ty) * g.tilesW + tx;
                if (co < g.Cout) g.stats[(((size_t)b * tpb + tIn) * 2 + (q & 1)) * g.Cout + co] = v;
            }
        }
    } else
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        const int off = out_off[wave * 32 + row];
        if (off < 0) continue;
        const size_t o = (size_t)off * g.Cout;
        if (co0 < g.Cout) {
            float v = acc0[r] + bias0;
            if (residual) v += residual[o + co0];
            y[o + co0] = v;
        }
        if (co1 < g.Cout) {
            float v = acc1[r] + bias1;
            if (residual) v += residual[o + co1];
            y[o + co1] = v;
        }
    }
    DIQT_STAMP();
    }   // tiles of this workgroup
    if (g.dbg && tid == 0) {
        for (int q = 0; q < 8; ++q) g.dbg[(size_t)blockIdx.x * 8 + q] = q < nst ? stamps[q] : 0ull;
        if (g.stagger == -1) {      // clock probe: slots 1 and 6 carry the constant-rate (100 MHz) counter at start / end
            g.dbg[(size_t)blockIdx.x * 8 + 1] = rt0;
            g.dbg[(size_t)blockIdx.x * 8 + 6] = __builtin_amdgcn_s_memrealtime();
            g.dbg[(size_t)blockIdx.x * 8 + 7] = __builtin_readcyclecounter();
        }
    }
#undef DIQT_STAMP
}

// ---------------------------------------------------------------------------------------------
// 8-wave variant for the large launches (>= two resident rounds): ONE 512-thread workgroup per CU owns 256 output voxels
// (e.g. 4x8x8: halo 6x10x10 = 2.34 input voxels per output voxel instead of 3.13 for the 128-voxel tile) x 64 output channels.
// With a single workgroup on the CU nothing else would run beside its staging, so the staging is taken off the critical path
// inside the workgroup instead of across workgroups:
//   * the NEXT chunk's halo is loaded global -> registers at the start of a chunk's last tap group and written to LDS at the
//     chunk boundary (the loads fly during ~6k MFMA cycles per wave);
//   * weight panels come in groups of F8_TG taps, double-buffered: group s+1 global -> registers before the taps of group s, ->
//     the other LDS buffer after them: ONE barrier per F8_TG taps (per 96 MFMAs of a wave) instead of one per tap.
// Two waves per SIMD as before (wave w: voxels 32w..32w+31, both 32-channel halves), same fragment order, LDS rows, packed weights
// and epilogue (incl. the per-tile column sums) as conv_fwd_kernel, hence the same bits per output element.
// ---------------------------------------------------------------------------------------------
constexpr int F8_MT = 256;       // output voxels per workgroup
constexpr int F8_TG = 2;         // taps per step
constexpr int F8_NWB = 3;        // weight-group buffers (see the step protocol in the kernel)
constexpr int F8_HREG = 10;      // 16-byte halo pieces per thread held in registers (HV <= 640)
constexpr int F8_WREG = F8_TG;   // 16-byte weight pieces per thread and step: a tap's 64 x 32-float panel = 512 pieces

__global__ __launch_bounds__(512, 2) void conv_fwd8_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                           const float* __restrict__ bias, const float* __restrict__ residual,
                                                           float* __restrict__ y, ConvGeom g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int HV = g.HD * g.HH * g.HWd;
    float* halo = smem;                                              // [HV][36]
    float* wbuf = smem + (size_t)HV * LDSROW;                        // [F8_NWB][F8_TG][64][36]
    constexpr int WBUF = F8_TG * NT * LDSROW;
    int* out_off = reinterpret_cast<int*>(wbuf + F8_NWB * WBUF);     // [2][256]   (slot = parity of the workgroup's tile count)
    int* halo_src = out_off + 2 * F8_MT;                             // [2][HV]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int T = g.kd * g.kh * g.kw;
    const int nGroups = (T + F8_TG - 1) / F8_TG;
    const int nSteps = g.nChunks * nGroups;

    // Persistent tile walk: workgroup w (XCD-contiguous numbering) runs the tiles w, w + G, w + 2G, ... (G = gridDim.x; the host makes
    // G a multiple of nNt, so the 64-channel block n0 of a workgroup never changes).  A tile boundary is handled like a chunk boundary:
    // the next tile's tables are filled two steps before the end (published by that step's barrier), its first halo chunk is
    // prefetched into registers during the last step and the weight stream simply continues, so table fill, the first chunk's
    // memory latency and the workgroup launch are off the MFMA path; what remains between two tiles is the epilogue and one full stop.
    const unsigned G = gridDim.x;
    const unsigned totalTiles = (unsigned)g.B * g.tilesD * g.tilesH * g.tilesW * g.nNt;
    unsigned L = xcd_remap(blockIdx.x, G);
    if (L >= totalTiles) return;
    const int nMine = (int)((totalTiles - 1 - L) / G) + 1;
    const int n0 = (int)(L % g.nNt) * NT;

    auto fill_tables = [&](unsigned Lt, int slot) {
        int mt = (int)(Lt / g.nNt);
        const int tx = mt % g.tilesW; mt /= g.tilesW;
        const int ty = mt % g.tilesH; mt /= g.tilesH;
        const int tz = mt % g.tilesD;
        const int b = mt / g.tilesD;
        const int d0 = tz * g.TD, h0 = ty * g.TH, w0 = tx * g.TW;
        if (tid < F8_MT) {
            const int tw = tid % g.TW, th = (tid / g.TW) % g.TH, td = tid / (g.TW * g.TH);
            const int od = d0 + td, oh = h0 + th, ow = w0 + tw;
            int off = (int)BUF_OOB;
            if (od < g.Do && oh < g.Ho && ow < g.Wo) off = (((b * g.Do + od) * g.Ho + oh) * g.Wo + ow) * g.Cout * 4;
            out_off[slot * F8_MT + tid] = off;
        }
        for (int hv = tid; hv < HV; hv += 512) {
            const int hx = hv % g.HWd, hy = (hv / g.HWd) % g.HH, hz = hv / (g.HWd * g.HH);
            const int iz = d0 + hz - g.pd, iy = h0 + hy - g.ph, ix = w0 + hx - g.pw;
            const int sv = conv_src_voxel(g, b, iz, iy, ix);
            halo_src[slot * HV + hv] = sv < 0 ? (int)BUF_OOB : sv * g.Cin * 4;
        }
    };
    const auto rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, (int)g.xBytes, 0x00020000);
    const auto rs_y = __builtin_amdgcn_make_buffer_rsrc(y, 0, (int)g.yBytes, 0x00020000);
    const auto rs_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(residual), 0, residual ? (int)g.yBytes : 0, 0x00020000);

    int hidx_lane;
    {
        const int v = wave * 32 + l31;
        const int tw = v % g.TW, th = (v / g.TW) % g.TH, td = v / (g.TW * g.TH);
        hidx_lane = (td * g.HH + th) * g.HWd + tw;
    }
    f32x16 acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }

    const int hc4 = (tid & 7) * 4;                  // this thread's channel quad inside a chunk (512 % 8 == 0)
    const int nHalo = HV * 8;
    fill_tables(L, 0);
    __syncthreads();                                // tables visible

    u32x4 hr[F8_HREG];
    auto load_halo = [&](int ci0, int slot) {       // all pieces of a chunk in flight at once; unconditional loads (range-checked descriptor)
        const unsigned coff = (ci0 + hc4 < g.Cin) ? (unsigned)(ci0 + hc4) * 4u : BUF_OOB_C;
        const int* tbl = halo_src + slot * HV;
#pragma unroll
        for (int u = 0; u < F8_HREG; ++u) {
            const int idx = u * 512 + tid;
            const unsigned t = (unsigned)tbl[min(idx >> 3, HV - 1)] + coff;
            hr[u] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, idx < nHalo ? t : BUF_OOB, 0, 0);
        }
    };
    auto store_halo = [&]() {
#pragma unroll
        for (int u = 0; u < F8_HREG; ++u) {
            const int idx = u * 512 + tid;
            if (idx < nHalo) *reinterpret_cast<u32x4*>(halo + (idx >> 3) * LDSROW + hc4) = hr[u];
        }
    };
    u32x4 wr[F8_WREG];
    const int wrow = tid >> 3, wc4 = (tid & 7) * 4;            // piece tid of a tap panel: row tid / 8, channel quad tid % 8
    // weight panels of (chunk, tap group) -> registers (unconditional, clamped tap); all index arithmetic of the step loop is kept
    // incremental and wave-uniform: a division per step costs as much as one of its MFMAs
    const float* wthread = wp + ((size_t)n0 + wrow) * CK + wc4;
    const size_t tapStride = (size_t)g.CoutPad * CK;
    auto load_wstep = [&](int chunk, int grp) {
        const int t0 = grp * F8_TG, n = min(F8_TG, T - t0);
        const float* src = wthread + ((size_t)chunk * T + t0) * tapStride;
#pragma unroll
        for (int u = 0; u < F8_WREG; ++u) wr[u] = *reinterpret_cast<const u32x4*>(src + (size_t)min(u, n - 1) * tapStride);
    };
    float* wthread_lds = wbuf + wrow * LDSROW + wc4;
    auto store_wstep = [&](int buf) {                          // a clamped duplicate panel is harmless
#pragma unroll
        for (int u = 0; u < F8_WREG; ++u) *reinterpret_cast<u32x4*>(wthread_lds + buf * WBUF + u * (NT * LDSROW)) = wr[u];
    };
    const int co0 = n0 + l31, co1 = n0 + 32 + l31;
    const float bias0 = (bias && co0 < g.Cout) ? bias[co0] : 0.f;
    const float bias1 = (bias && co1 < g.Cout) ? bias[co1] : 0.f;
    const unsigned c0 = co0 < g.Cout ? (unsigned)co0 * 4u : BUF_OOB_C, c1 = co1 < g.Cout ? (unsigned)co1 * 4u : BUF_OOB_C;
    // epilogue of the finished tile: D[row=voxel][col=co]; row = (r&3) + 8*(r>>2) + 4*h
    auto epilogue = [&](unsigned Lt, int slot) {
        float cs0 = 0.f, cq0 = 0.f, cs1 = 0.f, cq1 = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const unsigned off = (unsigned)out_off[slot * F8_MT + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h];
            float v0 = acc0[r] + bias0, v1 = acc1[r] + bias1;
            if (residual) {        // wave-uniform (kept per element: batching the 32 loads costs this kernel's main loop 4 %)
                v0 += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_r, off + c0, 0, 0));
                v1 += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_r, off + c1, 0, 0));
            }
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v0), rs_y, off + c0, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v1), rs_y, off + c1, 0, 0);
            if (g.stats && off != BUF_OOB) { cs0 += v0; cq0 = fmaf(v0, v0, cq0); cs1 += v1; cq1 = fmaf(v1, v1, cq1); }
            acc0[r] = 0.f; acc1[r] = 0.f;
        }
        if (g.stats) {             // block-uniform: fixed-order combine of lane halves, then of the 8 waves through LDS
            cs0 += __shfl_xor(cs0, 32, 64); cq0 += __shfl_xor(cq0, 32, 64);
            cs1 += __shfl_xor(cs1, 32, 64); cq1 += __shfl_xor(cq1, 32, 64);
            float* red = halo;     // the tile's last MFMA operands are in registers and its trailing prefetch reads are never used
            __syncthreads();       // ... but every wave has to be past them before the image is reused as scratch
            if (h == 0) {
                red[(wave * 4 + 0) * 32 + l31] = cs0; red[(wave * 4 + 1) * 32 + l31] = cq0;
                red[(wave * 4 + 2) * 32 + l31] = cs1; red[(wave * 4 + 3) * 32 + l31] = cq1;
            }
            __syncthreads();
            if (tid < 128) {       // q = tid >> 5: 0 sum(co0) 1 sumsq(co0) 2 sum(co1) 3 sumsq(co1)
                const int q = tid >> 5, l = tid & 31;
                float v = red[q * 32 + l];
#pragma unroll
                for (int w = 1; w < 8; ++w) v += red[(w * 4 + q) * 32 + l];
                const int co = n0 + (q >> 1) * 32 + l;
                const int tpb = g.tilesD * g.tilesH * g.tilesW, mtile = (int)(Lt / g.nNt);
                if (co < g.Cout) g.stats[(((size_t)(mtile / tpb) * tpb + mtile % tpb) * 2 + (q & 1)) * g.Cout + co] = v;
            }
        }
    };

    // Step protocol (a step = F8_TG taps of one chunk; step st reads weight buffer st % 3; the stream of steps runs on across tiles):
    //   start of step st : store W(st+1) (registers, loaded during step st-1) into buffer (st+1) % 3 -- its previous content W(st-2)
    //                      was last read in step st-2, and every wave that passed the barrier of step st-1 has finished step st-2;
    //                      then request W(st+2) from global memory into the registers;
    //   before the step's last 8-MFMA group: ONE barrier (the group's operands are already in registers, so nobody restarts cold);
    //                      behind it W(st+1) is visible and the first fragments of step st+1 are prefetched during that last group.
    // So a wave never waits for LDS data at a step boundary, and a wave whose SIMD partner sits in the barrier runs at full rate.
    // Only a chunk (or tile) boundary -- a new halo image -- is a full stop: barrier, halo registers -> LDS, barrier, cold reads.
    load_halo(0, 0);
    load_wstep(0, 0);
    store_halo();
    store_wstep(0);
    int remaining = nMine * nSteps;                               // steps left in this workgroup's stream, the current one included
    int chunk = 0, grp = 0, sc1 = 0, sg1 = 0, sc2, sg2;           // (chunk, group) of the steps st, st+1, st+2 (wrapping from tile to tile)
    auto advance = [&](int& c, int& gq) { if (++gq == nGroups) { gq = 0; if (++c == g.nChunks) c = 0; } };
    advance(sc1, sg1);
    sc2 = sc1; sg2 = sg1;
    advance(sc2, sg2);
    int bufNext = 1, slot = 0, stepInTile = 0;
    if (remaining > 1) load_wstep(sc1, sg1);
    __syncthreads();

    int kx = 0, ky = 0, trow = 0;                                 // tap (kz, ky, kx) of the NEXT tap to run and its halo row offset
    const float* ap = halo + hidx_lane * LDSROW + 4 * h;
    const float* bp = wbuf + l31 * LDSROW + 4 * h;
    float4 a = *reinterpret_cast<const float4*>(ap);
    float4 b0 = *reinterpret_cast<const float4*>(bp);
    float4 b1 = *reinterpret_cast<const float4*>(bp + 32 * LDSROW);
    __builtin_amdgcn_s_waitcnt(0xc07f);                           // lgkmcnt(0)
    for (; remaining > 0; --remaining) {
        const int t0 = grp * F8_TG, nTap = min(F8_TG, T - t0);
        const bool lastOfChunk = grp + 1 == nGroups, lastOfTile = lastOfChunk && chunk + 1 == g.nChunks, more = remaining > 1;
        if (more) store_wstep(bufNext);
        if (remaining > 2) load_wstep(sc2, sg2);
        if (stepInTile == nSteps - 2 && remaining > 2) fill_tables(L + G, slot ^ 1);      // published by this step's barrier
        const bool nextHalo = lastOfChunk && more;
        if (nextHalo) load_halo(lastOfTile ? 0 : (chunk + 1) * CK, lastOfTile ? slot ^ 1 : slot);
        for (int t = 0; t < nTap; ++t) {
            // halo row offset of the tap after this one (wave-uniform counters; wraps to tap 0 at the end of a chunk)
            int ntrow = trow + 1, nkx = kx + 1, nky = ky;
            if (nkx == g.kw) {
                nkx = 0; ntrow += g.HWd - g.kw;
                if (++nky == g.kh) { nky = 0; ntrow += (g.HH - g.kh) * g.HWd; }
            }
            const bool lastTap = t + 1 == nTap;
            if (lastTap && lastOfChunk) { ntrow = 0; nkx = 0; nky = 0; }
            const float* apn = halo + (hidx_lane + ntrow) * LDSROW + 4 * h;
            const float* bpn = lastTap ? wbuf + bufNext * WBUF + l31 * LDSROW + 4 * h : bp + NT * LDSROW;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4 an, b0n, b1n;
                if (q < 3) {
                    an = *reinterpret_cast<const float4*>(ap + 8 * (q + 1));
                    b0n = *reinterpret_cast<const float4*>(bp + 8 * (q + 1));
                    b1n = *reinterpret_cast<const float4*>(bp + 32 * LDSROW + 8 * (q + 1));
                } else {
                    if (lastTap) __syncthreads();        // wave-uniform: the step's barrier, operands of this group already in registers
                    // unconditional prefetch of the next tap's / next step's first fragments (stale but harmless at a chunk boundary
                    // and after the last step: the values are reloaded or never used)
                    an = *reinterpret_cast<const float4*>(apn);
                    b0n = *reinterpret_cast<const float4*>(bpn);
                    b1n = *reinterpret_cast<const float4*>(bpn + 32 * LDSROW);
                }
                __builtin_amdgcn_sched_barrier(0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b0.x, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b1.x, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b0.y, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b1.y, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b0.z, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b1.z, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b0.w, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b1.w, acc1, 0, 0, 0);
                a = an; b0 = b0n; b1 = b1n;
            }
            trow = ntrow; kx = nkx; ky = nky;
            ap = apn; bp = bpn;
            __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0): the prefetch was requested eight MFMAs ago
        }
        if (lastOfTile) epilogue(L, slot);
        if (nextHalo) {
            __syncthreads();                             // every wave is done with this chunk's halo image (and the epilogue's scratch)
            store_halo();
            __syncthreads();
            a = *reinterpret_cast<const float4*>(ap);    // cold reads from the new image (ap / bp already point at the next step)
            b0 = *reinterpret_cast<const float4*>(bp);
            b1 = *reinterpret_cast<const float4*>(bp + 32 * LDSROW);
            __builtin_amdgcn_s_waitcnt(0xc07f);
        }
        if (lastOfTile) { L += G; slot ^= 1; stepInTile = 0; } else ++stepInTile;
        chunk = sc1; grp = sg1; sc1 = sc2; sg1 = sg2;
        advance(sc2, sg2);
        bufNext = bufNext + 1 == F8_NWB ? 0 : bufNext + 1;
    }
}

// ---------------------------------------------------------------------------------------------
// 1x1x1 filters / Linear rows (pixel-shuffle and down-sample projections, attention and feed-forward projections, res_conv):
// a plain GEMM y[rows][Cout] = x[rows][Cin] W^T.  With ONE tap per 32-channel chunk the generic kernel above stages a chunk,
// issues 32 MFMAs per wave and stages again -- every chunk pays a global round trip (61-71 TFLOP/s).  Here the next chunk's
// A rows and weight panel are loaded global -> registers BEFORE the chunk's MFMAs and written to the other LDS buffer after
// them (one barrier per chunk, no memory latency between chunks).  Same tile (128 rows x 64 co, 4 waves), packed weights,
// fragment order and buffer-descriptor epilogue as conv_fwd_kernel.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void conv1x1_fwd_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                             const float* __restrict__ bias, const float* __restrict__ residual,
                                                             float* __restrict__ y, ConvGeom g) {
    __shared__ __attribute__((aligned(16))) float As[2][MTILE * LDSROW];
    __shared__ __attribute__((aligned(16))) float Ws[2][NT * LDSROW];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const unsigned L = xcd_remap(blockIdx.x, gridDim.x);
    const int nt = L % g.nNt, mt = L / g.nNt;
    const int n0 = nt * NT;
    const long long rows = (long long)g.B * g.D * g.H * g.W;          // make_geom flattened every voxel into W
    const long long r0 = (long long)mt * MTILE;
    const auto rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, (int)g.xBytes, 0x00020000);

    // this thread's pieces: A rows (tid >> 3) + 32 u, channel quad (tid & 7); weight rows (tid >> 3), +32
    const int prow = tid >> 3, pc4 = (tid & 7) * 4;
    unsigned aoff[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const long long r = r0 + prow + 32 * u;
        aoff[u] = r < rows ? (unsigned)(r * g.Cin * 4) : BUF_OOB;
    }
    u32x4 ra[4];
    float4 rw0, rw1;
    auto load_chunk = [&](int chunk) {
        const int ci = chunk * CK + pc4;
        const unsigned coff = ci < g.Cin ? (unsigned)ci * 4u : BUF_OOB_C;
#pragma unroll
        for (int u = 0; u < 4; ++u) ra[u] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, aoff[u] + coff, 0, 0);
        const float* wchunk = wp + ((size_t)chunk * g.CoutPad + n0) * CK;
        rw0 = *reinterpret_cast<const float4*>(wchunk + (size_t)prow * CK + pc4);
        rw1 = *reinterpret_cast<const float4*>(wchunk + (size_t)(prow + 32) * CK + pc4);
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int u = 0; u < 4; ++u) *reinterpret_cast<u32x4*>(&As[buf][(prow + 32 * u) * LDSROW + pc4]) = ra[u];
        *reinterpret_cast<float4*>(&Ws[buf][prow * LDSROW + pc4]) = rw0;
        *reinterpret_cast<float4*>(&Ws[buf][(prow + 32) * LDSROW + pc4]) = rw1;
    };

    f32x16 acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
    load_chunk(0);
    store_chunk(0);
    __syncthreads();
    for (int chunk = 0; chunk < g.nChunks; ++chunk) {
        if (chunk + 1 < g.nChunks) load_chunk(chunk + 1);
        const float* ap = &As[chunk & 1][(wave * 32 + l31) * LDSROW + 4 * h];
        const float* bp = &Ws[chunk & 1][l31 * LDSROW + 4 * h];
        float4 a = *reinterpret_cast<const float4*>(ap);
        float4 b0 = *reinterpret_cast<const float4*>(bp);
        float4 b1 = *reinterpret_cast<const float4*>(bp + 32 * LDSROW);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 an, b0n, b1n;
            if (q < 3) {
                an = *reinterpret_cast<const float4*>(ap + 8 * (q + 1));
                b0n = *reinterpret_cast<const float4*>(bp + 8 * (q + 1));
                b1n = *reinterpret_cast<const float4*>(bp + 32 * LDSROW + 8 * (q + 1));
            }
            __builtin_amdgcn_sched_barrier(0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b0.x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b1.x, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b0.y, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b1.y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b0.z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b1.z, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b0.w, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b1.w, acc1, 0, 0, 0);
            if (q < 3) { a = an; b0 = b0n; b1 = b1n; }
        }
        if (chunk + 1 < g.nChunks) store_chunk((chunk + 1) & 1);       // last read in chunk-1, retired by its barrier
        __syncthreads();
    }

    // ---- epilogue: D[row][col = co]; row = (r&3) + 8*(r>>2) + 4*h ----
    const int co0 = n0 + l31, co1 = n0 + 32 + l31;
    const float bias0 = (bias && co0 < g.Cout) ? bias[co0] : 0.f;
    const float bias1 = (bias && co1 < g.Cout) ? bias[co1] : 0.f;
    const auto rs_y = __builtin_amdgcn_make_buffer_rsrc(y, 0, (int)g.yBytes, 0x00020000);
    const auto rs_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(residual), 0, residual ? (int)g.yBytes : 0, 0x00020000);
    const unsigned c0 = co0 < g.Cout ? (unsigned)co0 * 4u : BUF_OOB_C, c1 = co1 < g.Cout ? (unsigned)co1 * 4u : BUF_OOB_C;
    // residual: all 32 loads in flight before the first add (a load + wait + add per element serialises 16 L2 round trips)
    float rr0[16], rr1[16];
    if (residual) {            // wave-uniform
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long long row = r0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            const unsigned off = row < rows ? (unsigned)(row * g.Cout * 4) : BUF_OOB;
            rr0[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_r, off + c0, 0, 0));
            rr1[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_r, off + c1, 0, 0));
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const long long row = r0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const unsigned off = row < rows ? (unsigned)(row * g.Cout * 4) : BUF_OOB;
        float v0 = acc0[r] + bias0, v1 = acc1[r] + bias1;
        if (residual) { v0 += rr0[r]; v1 += rr1[r]; }
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v0), rs_y, off + c0, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v1), rs_y, off + c1, 0, 0);
    }
}

// ---------------------------------------------------------------------------------------------
// forward for Cin <= 4 (the init convs: 2 -> 64 3x3x3 of Family A, the (1,k,k) cross-embed convs of Family B): the GEMM K
// axis is (tap, ci) packed densely -- K = taps * CINP in chunks of 32 -- instead of one 32-wide (mostly zero) channel chunk
// per tap: 13.5x fewer MFMAs for 3x3x3 x 2 channels.  Per chunk an im2col tile As[128 voxels][32 k] is gathered from the
// compact halo image in LDS through a 32-entry offset table, then the usual b128-fragment MFMA loop runs on it.
// packed weights (conv_pack_smallcin_kernel): [chunk][co padded to 64][32 k],  k = tap * CINP + ci.
// ---------------------------------------------------------------------------------------------
// mode 0: effective (out, in) = (Cout, Cin), value w[o][i][tap];  mode 1 (backward-data of a conv with <= 4 OUTPUT channels):
// effective (out, in) = (Cin, Cout), value w[i][o][T-1-tap]
__global__ void conv_pack_smallcin_kernel(const float* __restrict__ w, float* __restrict__ packed, int Cout, int Cin, int T,
                                          int CINP, int CoutPadEff, int mode, size_t total) {
    const int outEff = mode == 0 ? Cout : Cin, inEff = mode == 0 ? Cin : Cout;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int kk = (int)(i % CK);
        size_t r = i / CK;
        const int o = (int)(r % CoutPadEff);
        const int chunk = (int)(r / CoutPadEff);
        const int k = chunk * CK + kk, tap = k / CINP, in = k % CINP;
        float v = 0.f;
        if (o < outEff && tap < T && in < inEff)
            v = mode == 0 ? w[((size_t)o * Cin + in) * T + tap] : w[((size_t)in * Cin + o) * T + (T - 1 - tap)];
        packed[i] = v;
    }
}

template <int CINP>
__global__ __launch_bounds__(256, 2) void conv_fwd_smallcin_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                                   const float* __restrict__ bias,
                                                                   const float* __restrict__ residual, float* __restrict__ y,
                                                                   ConvGeom g, int nCh) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int HV = g.HD * g.HH * g.HWd;
    const int haloElems = (HV * CINP + 3) & ~3;
    float* halo = smem;                                  // [HV][CINP]
    float* As = smem + haloElems;                        // [128][36]
    float* wbuf = As + MTILE * LDSROW;                   // [64][36]
    int* out_off = reinterpret_cast<int*>(wbuf + NT * LDSROW);      // [128]
    int* koff = out_off + MTILE;                                     // [32]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const unsigned L = xcd_remap(blockIdx.x, gridDim.x);
    const int nt = L % g.nNt;
    int mt = L / g.nNt;
    const int tx = mt % g.tilesW; mt /= g.tilesW;
    const int ty = mt % g.tilesH; mt /= g.tilesH;
    const int tz = mt % g.tilesD;
    const int b = mt / g.tilesD;
    const int d0 = tz * g.TD, h0 = ty * g.TH, w0 = tx * g.TW;
    const int n0 = nt * NT;
    const int T = g.kd * g.kh * g.kw;

    if (tid < MTILE) {
        const int tw = tid % g.TW, th = (tid / g.TW) % g.TH, td = tid / (g.TW * g.TH);
        const int od = d0 + td, oh = h0 + th, ow = w0 + tw;
        out_off[tid] = (od < g.Do && oh < g.Ho && ow < g.Wo) ? ((b * g.Do + od) * g.Ho + oh) * g.Wo + ow : -1;
    }
    for (int e = tid; e < HV * CINP; e += 256) {         // compact halo image, zero padded
        const int hv = e / CINP, ci = e % CINP;
        const int hx = hv % g.HWd, hy = (hv / g.HWd) % g.HH, hz = hv / (g.HWd * g.HH);
        const int iz = d0 + hz - g.pd, iy = h0 + hy - g.ph, ix = w0 + hx - g.pw;
        const int sv = ci < g.Cin ? conv_src_voxel(g, b, iz, iy, ix) : -1;
        halo[e] = sv >= 0 ? x[(size_t)sv * g.Cin + ci] : 0.f;
    }
    // the voxel this thread gathers for the im2col tile (rows 0..127, two 16-wide k halves)
    const int gv = tid & (MTILE - 1), ghalf = tid >> 7;
    const int gbase = (((gv / (g.TW * g.TH)) * g.HH + (gv / g.TW) % g.TH) * g.HWd + gv % g.TW) * CINP;

    f32x16 acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
    const bool two = n0 + 32 < g.Cout;                   // kernel-uniform per workgroup: the block's second 32 output channels exist
    const int wrow = tid >> 3, wc4 = (tid & 7) * 4;
    for (int chunk = 0; chunk < nCh; ++chunk) {
        __syncthreads();                                 // previous chunk's fragment reads (and the halo stores) are done
        if (tid < CK) {
            const int k = chunk * CK + tid, tap = k / CINP, ci = k % CINP;
            int o = -1;
            if (tap < T) {
                const int kx = tap % g.kw, ky = (tap / g.kw) % g.kh, kz = tap / (g.kw * g.kh);
                o = ((kz * g.HH + ky) * g.HWd + kx) * CINP + ci;
            }
            koff[tid] = o;
        }
        {
            const float* wc = wp + ((size_t)chunk * g.CoutPad + n0) * CK;
            *reinterpret_cast<float4*>(wbuf + wrow * LDSROW + wc4) = *reinterpret_cast<const float4*>(wc + (size_t)wrow * CK + wc4);
            *reinterpret_cast<float4*>(wbuf + (wrow + 32) * LDSROW + wc4) =
                *reinterpret_cast<const float4*>(wc + (size_t)(wrow + 32) * CK + wc4);
        }
        __syncthreads();
        {
            float v[16];
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                const int o = koff[ghalf * 16 + kk];
                v[kk] = halo[gbase + (o < 0 ? 0 : o)];                   // unconditional read, select afterwards
                v[kk] = o < 0 ? 0.f : v[kk];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                *reinterpret_cast<float4*>(As + gv * LDSROW + ghalf * 16 + 4 * j) = make_float4(v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]);
        }
        __syncthreads();
        const float* ap = As + (wave * 32 + l31) * LDSROW + 4 * h;
        const float* bp = wbuf + l31 * LDSROW + 4 * h;
        float4 a = *reinterpret_cast<const float4*>(ap);
        float4 b0 = *reinterpret_cast<const float4*>(bp);
        if (two) {
            float4 b1 = *reinterpret_cast<const float4*>(bp + 32 * LDSROW);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4 an, b0n, b1n;
                if (q < 3) {
                    an = *reinterpret_cast<const float4*>(ap + 8 * (q + 1));
                    b0n = *reinterpret_cast<const float4*>(bp + 8 * (q + 1));
                    b1n = *reinterpret_cast<const float4*>(bp + 32 * LDSROW + 8 * (q + 1));
                }
                __builtin_amdgcn_sched_barrier(0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b0.x, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b1.x, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b0.y, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b1.y, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b0.z, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b1.z, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b0.w, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b1.w, acc1, 0, 0, 0);
                if (q < 3) { a = an; b0 = b0n; b1 = b1n; }
            }
        } else {
            // at most 32 output channels in this block (the cross-embed convs of the pseudo-3D U-Net: 16 / 32 channels from 2): the
            // second 32-channel accumulator would multiply padding -- half the MFMAs of the kernel
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4 an, b0n;
                if (q < 3) {
                    an = *reinterpret_cast<const float4*>(ap + 8 * (q + 1));
                    b0n = *reinterpret_cast<const float4*>(bp + 8 * (q + 1));
                }
                __builtin_amdgcn_sched_barrier(0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b0.x, acc0, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b0.y, acc0, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b0.z, acc0, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b0.w, acc0, 0, 0, 0);
                if (q < 3) { a = an; b0 = b0n; }
            }
        }
    }
    const int co0 = n0 + l31, co1 = n0 + 32 + l31;
    const float bias0 = (bias && co0 < g.Cout) ? bias[co0] : 0.f;
    const float bias1 = (bias && co1 < g.Cout) ? bias[co1] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int off = out_off[wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h];
        if (off < 0) continue;
        const size_t o = (size_t)off * g.Cout;
        if (co0 < g.Cout) { float v = acc0[r] + bias0; if (residual) v += residual[o + co0]; y[o + co0] = v; }
        if (co1 < g.Cout) { float v = acc1[r] + bias1; if (residual) v += residual[o + co1]; y[o + co1] = v; }
    }
}

// ---------------------------------------------------------------------------------------------
// backward-weight (split-K partial slabs)
// ---------------------------------------------------------------------------------------------
constexpr int BW_MAXT = 5;       // taps per wave pair
struct BwGeom {
    ConvGeom g;
    int tapGroups;               // groups of <=10 (kh,kw) taps per kd plane
    int lTW, lTH;                // log2 of the (power-of-two) tile extents
    int tilesPerSplit, MT;       // voxel tiles per grid.y block, total voxel tiles
};

template <bool VEC4>
__global__ __launch_bounds__(256, 2) void conv_bwd_weight_kernel(const float* __restrict__ x,
                                                                 const float* __restrict__ dy,
                                                                 float* __restrict__ slabs, BwGeom bg) {
    const ConvGeom& g = bg.g;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int HHp = g.HH, HWp = g.HWd;                    // halo plane extents (TD x HH x HWd)
    const int HVp = g.TD * HHp * HWp;
    float* xh = smem;                                     // [HVp][32]
    float* dyt = smem + (size_t)HVp * CK;                 // [128][64]
    int* xsrc = reinterpret_cast<int*>(dyt + MTILE * NT);  // [HVp] halo voxel -> input voxel or -1
    int* ysrc = xsrc + HVp;                                // [128] tile voxel -> output voxel or -1

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform
    const int l31 = lane & 31, h = lane >> 5;
    const int coHalf = wave & 1, tapHalf = wave >> 1;

    // blockIdx.x -> (chunk, coTile, kz, tapGroup)
    int bx = blockIdx.x;
    const int tg = bx % bg.tapGroups; bx /= bg.tapGroups;
    const int kz = bx % g.kd; bx /= g.kd;
    const int ct = bx % g.nNt;
    const int chunk = bx / g.nNt;
    const int ci0 = chunk * CK, n0 = ct * NT;
    const int planeT = g.kh * g.kw;
    const int tapBase = tg * (2 * BW_MAXT) + tapHalf * BW_MAXT;       // first (kh,kw) tap of this wave
    int ntap = planeT - tapBase;
    ntap = ntap < 0 ? 0 : (ntap > BW_MAXT ? BW_MAXT : ntap);

    int tapoff[BW_MAXT];
#pragma unroll
    for (int t = 0; t < BW_MAXT; ++t) {
        const int tp = tapBase + t;
        const int ky = tp / g.kw, kx = tp % g.kw;
        tapoff[t] = (t < ntap) ? (ky * HWp + kx) * CK : 0;
    }

    f32x16 acc[BW_MAXT];
#pragma unroll
    for (int t = 0; t < BW_MAXT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

    const int mtBegin = blockIdx.y * bg.tilesPerSplit;
    int mtEnd = mtBegin + bg.tilesPerSplit;
    if (mtEnd > bg.MT) mtEnd = bg.MT;

    for (int mt0 = mtBegin; mt0 < mtEnd; ++mt0) {
        int mt = mt0;
        const int tx = mt % g.tilesW; mt /= g.tilesW;
        const int ty = mt % g.tilesH; mt /= g.tilesH;
        const int tz = mt % g.tilesD;
        const int b = mt / g.tilesD;
        const int d0 = tz * g.TD, h0 = ty * g.TH, w0 = tx * g.TW;
        // index tables (one division set per voxel, not per 16-byte piece); xsrc/ysrc are only read during staging
        for (int hv = tid; hv < HVp + MTILE; hv += 256) {
            if (hv < HVp) {
                const int hx = hv % HWp, hy = (hv / HWp) % HHp, hz = hv / (HWp * HHp);
                const int iz = d0 + hz + kz - g.pd, iy = h0 + hy - g.ph, ix = w0 + hx - g.pw;
                xsrc[hv] = (iz >= 0 && iz < g.D && iy >= 0 && iy < g.H && ix >= 0 && ix < g.W)
                               ? ((b * g.D + iz) * g.H + iy) * g.W + ix : -1;
            } else {
                const int v = hv - HVp;
                const int tw = v % g.TW, th = (v / g.TW) % g.TH, td = v / (g.TW * g.TH);
                const int od = d0 + td, oh = h0 + th, ow = w0 + tw;
                ysrc[v] = (od < g.Do && oh < g.Ho && ow < g.Wo) ? ((b * g.Do + od) * g.Ho + oh) * g.Wo + ow : -1;
            }
        }
        __syncthreads();   // tables ready; all MFMA reads of the previous tile's xh/dyt are done (they precede this barrier)
        // x halo plane for this kz: input depth rows d0 + td + kz - pd
        for (int idx = tid; idx < HVp * 8; idx += 256) {
            const int hv = idx >> 3, c4 = (idx & 7) * 4;
            const int src = xsrc[hv];
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (src >= 0) {
                const size_t base = (size_t)src * g.Cin + ci0 + c4;
                if (VEC4) {
                    if (ci0 + c4 < g.Cin) v = *reinterpret_cast<const float4*>(x + base);
                } else {
                    const int rem = g.Cin - (ci0 + c4);
                    if (rem > 0) v.x = x[base];
                    if (rem > 1) v.y = x[base + 1];
                    if (rem > 2) v.z = x[base + 2];
                    if (rem > 3) v.w = x[base + 3];
                }
            }
            *reinterpret_cast<float4*>(xh + hv * CK + c4) = v;
        }
        // dY tile [128][64]
        for (int idx = tid; idx < MTILE * 16; idx += 256) {
            const int v = idx >> 4, c4 = (idx & 15) * 4;
            const int src = ysrc[v];
            float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
            if (src >= 0) {
                const size_t base = (size_t)src * g.Cout + n0 + c4;
                const int rem = g.Cout - (n0 + c4);
                if ((g.Cout & 3) == 0) {
                    if (rem > 0) r = *reinterpret_cast<const float4*>(dy + base);
                } else {
                    if (rem > 0) r.x = dy[base];
                    if (rem > 1) r.y = dy[base + 1];
                    if (rem > 2) r.z = dy[base + 2];
                    if (rem > 3) r.w = dy[base + 3];
                }
            }
            *reinterpret_cast<float4*>(dyt + v * NT + c4) = r;
        }
        __syncthreads();
        if (ntap > 0) {
            // K loop over voxel pairs; lane half h takes voxel 2s+h.  Tile extents are powers of two, so the
            // voxel -> halo index map is shifts/masks.  All BW_MAXT MFMAs are issued unconditionally (taps beyond
            // ntap alias tap 0 and are discarded at the end): no branch sits between the LDS reads and the MFMAs.
            const float* ap0 = dyt + coHalf * 32 + l31;
            const float* bp0 = xh + l31;
            const int mTW = g.TW - 1, mTH = g.TH - 1, sTH = bg.lTW, sTD = bg.lTW + bg.lTH;
#pragma unroll 4
            for (int s2 = 0; s2 < MTILE / 2; ++s2) {
                const int v = 2 * s2 + h;
                const int tw = v & mTW, th = (v >> sTH) & mTH, td = v >> sTD;
                const float a = ap0[v * NT];
                const float* bp = bp0 + ((td * HHp + th) * HWp + tw) * CK;
                float bv[BW_MAXT];
#pragma unroll
                for (int t = 0; t < BW_MAXT; ++t) bv[t] = bp[tapoff[t]];
#pragma unroll
                for (int t = 0; t < BW_MAXT; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv[t], acc[t], 0, 0, 0);
            }
        }
    }

    // slab layout == packed layout: [ksplit][chunk][tap][CoutPad][32]
    const int T = g.kd * planeT;
    float* slab = slabs + (size_t)blockIdx.y * g.nChunks * T * g.CoutPad * CK;
#pragma unroll
    for (int t = 0; t < BW_MAXT; ++t) {
        if (t < ntap) {
            const int tap = kz * planeT + tapBase + t;
            float* dst = slab + (((size_t)chunk * T + tap) * g.CoutPad + n0 + coHalf * 32) * CK + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * h;      // co within the half
                dst[(size_t)row * CK] = acc[t][r];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// backward-weight, version 2 (default): a workgroup owns  COB co x 32 ci x TG taps  and walks its split-K range of
// voxel tiles.  SPLIT_CO = false (filters with >= 4 taps): COB = 32 and the 4 waves split the taps (7/7/7/6 for 3x3x3:
// balanced to 96 %), all sharing the full halo tile;  SPLIT_CO = true (1x1x1 and other tiny filters): COB = 128 and
// the waves split the output channels.  The NEXT tile's X halo and dY tile are fetched global -> registers while the
// current tile's MFMAs run and are written to LDS at the tile boundary, so HBM/Infinity-Cache latency is off the
// critical path and each element of a tile is staged once per workgroup.
// ---------------------------------------------------------------------------------------------
struct BwGeom2 {
    ConvGeom g;
    int tapGroups, coBlocks;     // grid.x = nChunks * coBlocks * tapGroups
    int tilesPerSplit, MT;
    int lTW, lTH;
    int maxtA;                   // mode A: taps per SIMD (7 for 3x3x3; 3 for <= 12-tap filters such as (1,3,3))
};

template <bool VEC4, int NRX, int NRY, int MAXT, bool SPLIT_CO, int NTHR>
__global__ __launch_bounds__(NTHR, NTHR / 256) void conv_bwd_weight2_kernel(const float* __restrict__ x,
                                                                  const float* __restrict__ dy,
                                                                  float* __restrict__ slabs,
                                                                  float* __restrict__ bias_part, BwGeom2 bg) {
    const ConvGeom& g = bg.g;
    constexpr int COB = SPLIT_CO ? 128 : 32;
    constexpr int NWAVE = NTHR / 64;
    // SPLIT_CO = false: SIMD q (waves q and q+4) owns taps [q*MAXT, (q+1)*MAXT); its two waves take alternate k-steps
    // of every tile and write separate split-K slabs, so all 8 waves run the same code with MAXT accumulators
    constexpr int TG = SPLIT_CO ? MAXT : 4 * MAXT;
    constexpr int KPAR = SPLIT_CO ? 1 : NWAVE / 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int HV = g.HD * g.HH * g.HWd;
    float* xh = smem;                                        // [HV][32]
    float* dyt = smem + (size_t)HV * CK;                     // [128][COB]
    int* xsrc = reinterpret_cast<int*>(dyt + MTILE * COB);   // [2][HV]
    int* ysrc = xsrc + 2 * HV;                               // [2][128]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int T = g.kd * g.kh * g.kw;

    int bx = blockIdx.x;
    const int tg = bx % bg.tapGroups; bx /= bg.tapGroups;
    const int cb = bx % bg.coBlocks;
    const int chunk = bx / bg.coBlocks;
    const int ci0 = chunk * CK, n0 = cb * COB;
    const int co_off = SPLIT_CO ? 32 * wave : 0;
    const int kpar = SPLIT_CO ? 0 : wave >> 2;
    const int tapBase = tg * TG + (SPLIT_CO ? 0 : (wave & 3) * MAXT);
    int ntap = min(T, (tg + 1) * TG) - tapBase;
    ntap = ntap < 0 ? 0 : (ntap > MAXT ? MAXT : ntap);
    if (SPLIT_CO && n0 + co_off >= g.CoutPad) ntap = 0;

    int tapoff[MAXT];
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
        const int tp = tapBase + t;
        const int kx = tp % g.kw, ky = (tp / g.kw) % g.kh, kz = tp / (g.kw * g.kh);
        tapoff[t] = (t < ntap) ? ((kz * g.HH + ky) * g.HWd + kx) * CK : 0;
    }
    f32x16 acc[MAXT];
#pragma unroll
    for (int t = 0; t < MAXT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

    const int mtBegin = blockIdx.y * bg.tilesPerSplit;
    const int mtEnd = min(mtBegin + bg.tilesPerSplit, bg.MT);

    const float rHWd = 1.f / (float)g.HWd, rHH = 1.f / (float)g.HH;
    auto small_div = [](int a, int d, float rd) {        // a / d for 0 <= a < 2^16, d >= 1
        int q = (int)((float)a * rd);
        const int r = a - q * d;
        q += (r >= d) - (r < 0);
        return q;
    };
    auto fill_tables = [&](int mt0, int slot) {
        int mt = mt0;
        const int tx = mt % g.tilesW; mt /= g.tilesW;
        const int ty = mt % g.tilesH; mt /= g.tilesH;
        const int tz = mt % g.tilesD;
        const int b = mt / g.tilesD;
        const int d0 = tz * g.TD, h0 = ty * g.TH, w0 = tx * g.TW;
        for (int hv = tid; hv < HV + MTILE; hv += NTHR) {
            if (hv < HV) {
                // hv = (hz * HH + hy) * HWd + hx with hv < 2^16: quotients by float reciprocal + one correction step (exact), instead
                // of four ~40-instruction integer divisions per entry
                const int row = small_div(hv, g.HWd, rHWd), hx = hv - row * g.HWd;
                const int hz = small_div(row, g.HH, rHH), hy = row - hz * g.HH;
                const int iz = d0 + hz - g.pd, iy = h0 + hy - g.ph, ix = w0 + hx - g.pw;
                xsrc[slot * HV + hv] = (iz >= 0 && iz < g.D && iy >= 0 && iy < g.H && ix >= 0 && ix < g.W)
                                           ? ((b * g.D + iz) * g.H + iy) * g.W + ix : -1;
            } else {
                const int v = hv - HV;
                const int tw = v & (g.TW - 1), th = (v >> bg.lTW) & (g.TH - 1), td = v >> (bg.lTW + bg.lTH);
                const int od = d0 + td, oh = h0 + th, ow = w0 + tw;
                ysrc[slot * MTILE + v] = (od < g.Do && oh < g.Ho && ow < g.Wo) ? ((b * g.Do + od) * g.Ho + oh) * g.Wo + ow : -1;
            }
        }
    };
    float4 RX[NRX], RY[NRY];
    unsigned okx = 0, oky = 0;
    // bias gradient rides along: the workgroups of (chunk 0, tap group 0) see every dY element of their co-block once;
    // a thread always stages the same channel quad (NTHR % (COB/4) == 0), so it keeps one float4 of column sums
    const bool doBias = bias_part != nullptr && chunk == 0 && tg == 0;
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
    // all loads are unconditional (clamped address) so hipcc emits them back to back; invalid pieces are zero-selected
    auto issue_loads = [&](int slot) {
        // ALL table entries first, then the address arithmetic and the loads: written in one loop, hipcc serialises
        // ds_read -> s_waitcnt lgkmcnt(0) -> address -> global_load per piece (nine LDS round trips: ~3.5k cycles per tile)
        int sx[NRX], sy[NRY];
#pragma unroll
        for (int r = 0; r < NRX; ++r) {
            const int idx = tid + NTHR * r;
            sx[r] = xsrc[slot * HV + ((idx < HV * 8) ? (idx >> 3) : 0)];
        }
#pragma unroll
        for (int r = 0; r < NRY; ++r) sy[r] = ysrc[slot * MTILE + (tid + NTHR * r) / (COB / 4)];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < NRX; ++r) {
            const int idx = tid + NTHR * r;
            const int c4 = (idx & 7) * 4;
            const int src = sx[r];
            const bool ok = idx < HV * 8 && src >= 0 && ci0 + c4 < g.Cin;
            const size_t off = ok ? (size_t)src * g.Cin + ci0 + c4 : 0;
            float4 v;
            if (VEC4) {
                v = *reinterpret_cast<const float4*>(x + off);
            } else {
                const int rem = ok ? g.Cin - (ci0 + c4) : 0;
                v.x = x[off];
                v.y = rem > 1 ? x[off + 1] : 0.f;
                v.z = rem > 2 ? x[off + 2] : 0.f;
                v.w = rem > 3 ? x[off + 3] : 0.f;
            }
            RX[r] = v;                                   // zero-select deferred to store_tiles (see okx)
            if (ok) okx |= (1u << r); else okx &= ~(1u << r);
        }
#pragma unroll
        for (int r = 0; r < NRY; ++r) {
            const int idx = tid + NTHR * r;                   // idx < 128 * COB / 4 always (NRY = 128*COB/1024)
            const int c4 = (idx % (COB / 4)) * 4;
            const int src = sy[r];
            const int rem0 = g.Cout - (n0 + c4);
            const bool ok = src >= 0 && rem0 > 0;
            const size_t off = ok ? (size_t)src * g.Cout + n0 + c4 : 0;
            float4 q;
            if ((g.Cout & 3) == 0) {
                q = *reinterpret_cast<const float4*>(dy + off);
            } else {
                const int rem = ok ? rem0 : 0;
                q.x = dy[off];
                q.y = rem > 1 ? dy[off + 1] : 0.f;
                q.z = rem > 2 ? dy[off + 2] : 0.f;
                q.w = rem > 3 ? dy[off + 3] : 0.f;
            }
            RY[r] = q;
            if (ok) oky |= (1u << r); else oky &= ~(1u << r);
        }
    };
    auto store_tiles = [&]() {
#pragma unroll
        for (int r = 0; r < NRX; ++r) {
            const int idx = tid + NTHR * r;
            if (idx < HV * 8)
                *reinterpret_cast<float4*>(xh + (idx >> 3) * CK + (idx & 7) * 4) = (okx >> r) & 1u ? RX[r] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int r = 0; r < NRY; ++r) {
            const int idx = tid + NTHR * r;
            const float4 q = (oky >> r) & 1u ? RY[r] : make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4*>(dyt + (idx / (COB / 4)) * COB + (idx % (COB / 4)) * 4) = q;
            if (doBias) { bsum.x += q.x; bsum.y += q.y; bsum.z += q.z; bsum.w += q.w; }
        }
    };

    long long tsum[5] = {0, 0, 0, 0, 0};
    long long tprev = g.dbg ? (long long)__builtin_readcyclecounter() : 0;
#define DIQT_ACC(i) do { if (g.dbg) { const long long tn = (long long)__builtin_readcyclecounter(); tsum[i] += tn - tprev; tprev = tn; } } while (0)
    const int mTW = g.TW - 1, mTH = g.TH - 1, sTH = bg.lTW, sTD = bg.lTW + bg.lTH;
    if (mtBegin < mtEnd) {
        fill_tables(mtBegin, 0);
        __syncthreads();
        issue_loads(0);
    }
    int slot = 0;
    for (int mt0 = mtBegin; mt0 < mtEnd; ++mt0) {
        const bool haveNext = mt0 + 1 < mtEnd;
        __syncthreads();                 // all MFMA-phase reads of the previous tile are done
        DIQT_ACC(0);
        store_tiles();                   // waits for the registers prefetched one tile ago
        DIQT_ACC(1);
        if (haveNext) fill_tables(mt0 + 1, slot ^ 1);
        __syncthreads();
        DIQT_ACC(2);
        if (haveNext) issue_loads(slot ^ 1);
        DIQT_ACC(3);
        if (ntap > 0) {
            // voxel v = 2*s2 + h: bit 0 of v lies in exactly one of the (td, th, tw) bit-fields, so the halo offset of v is
            // offset(2*s2) [wave-uniform, scalar ALU] + offset(h) [per lane, hoisted]
            const int hoff = ((((h >> sTD) * g.HH + ((h >> sTH) & mTH)) * g.HWd) + (h & mTW)) * CK;
            const float* ap0 = dyt + co_off + l31 + h * COB;
            const float* bp0 = xh + l31 + hoff;
            auto soff = [&](int s2) {
                const int v = 2 * s2;
                return ((((v >> sTD) * g.HH + ((v >> sTH) & mTH)) * g.HWd) + (v & mTW)) * CK;
            };
            // software pipeline: the LDS reads of step s+1 are issued behind the first MFMA of step s, so the other
            // MAXT-1 MFMAs cover their latency (hipcc otherwise serialises read -> wait -> MFMA through one register)
            auto rd = [&](int s2, float& a, float (&b)[MAXT]) {
                a = ap0[2 * s2 * COB];
                const float* bp = bp0 + soff(s2);
#pragma unroll
                for (int t = 0; t < MAXT; ++t) b[t] = bp[tapoff[t]];
            };
            auto mm = [&](float a, const float (&b)[MAXT]) {
#pragma unroll
                for (int t = 0; t < MAXT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[t], acc[t], 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, MAXT + 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, MAXT - 1, 0);
            };
            float a0, b0[MAXT], a1, b1[MAXT];          // ping-pong operand registers (MTILE/2/KPAR steps: even)
            rd(kpar, a0, b0);
            for (int s2 = kpar; s2 < MTILE / 2; s2 += 2 * KPAR) {
                rd(s2 + KPAR, a1, b1);
                mm(a0, b0);
                rd(s2 + 2 * KPAR < MTILE / 2 ? s2 + 2 * KPAR : kpar, a0, b0);      // the last step re-reads the first (unused)
                mm(a1, b1);
            }
        }
        DIQT_ACC(4);
        slot ^= 1;
    }
    if (g.dbg && lane == 0)       // one record per wave: 5 phase sums + SIMD id (HW_REG_HW_ID bits 5:4)
        for (int q = 0; q < 6; ++q)
            g.dbg[(((size_t)blockIdx.y * gridDim.x + blockIdx.x) * NWAVE + wave) * 8 + q] =
                q < 5 ? (unsigned long long)tsum[q] : (unsigned long long)((__builtin_amdgcn_s_getreg((6 - 1) << 11 | 0 << 6 | 4) >> 4) & 3);
#undef DIQT_ACC

    if (doBias) {                        // block-uniform
        __syncthreads();
        float4* sb = reinterpret_cast<float4*>(smem);
        sb[tid] = bsum;
        __syncthreads();
        if (tid < COB && n0 + tid < g.CoutPad) {
            float sacc = 0.f;
            for (int t = tid >> 2; t < NTHR; t += COB / 4) sacc += reinterpret_cast<const float*>(&sb[t])[tid & 3];
            bias_part[(size_t)blockIdx.y * g.CoutPad + n0 + tid] = sacc;
        }
    }

    // the two waves of a SIMD hold partial sums of the same taps (alternate k-steps): combine them through LDS so a workgroup
    // writes ONE slab (halves the slab traffic and the split-K reduce)
    if (KPAR == 2) {
        __syncthreads();
        float* red = smem;                       // [4][MAXT][16][64] floats (the launch reserves this much LDS)
        if (wave >= 4) {
#pragma unroll
            for (int t = 0; t < MAXT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[(((wave & 3) * MAXT + t) * 16 + r) * 64 + lane] = acc[t][r];
        }
        __syncthreads();
        if (wave < 4) {
#pragma unroll
            for (int t = 0; t < MAXT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] += red[((wave * MAXT + t) * 16 + r) * 64 + lane];
        }
        if (wave >= 4) return;                   // no barrier follows on this path
    }
    float* slab = slabs + (size_t)blockIdx.y * g.nChunks * T * g.CoutPad * CK;
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
        if (t < ntap) {
            const int tap = tapBase + t;
            float* dst = slab + (((size_t)chunk * T + tap) * g.CoutPad + n0 + co_off) * CK + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) dst[(size_t)((r & 3) + 8 * (r >> 2) + 4 * h) * CK] = acc[t][r];
        }
    }
}

// dW[co][ci][tap] = sum_ks slab[ks][ci/32][tap][co][ci%32];  optionally dbias[c] = sum_y bias_part[y][c]
__global__ void conv_reduce_dw_kernel(const float* __restrict__ slabs, float* __restrict__ dw,
                                      int Cout, int Cin, int T, int CoutPad, int nChunks, int ksplit,
                                      const float* __restrict__ bias_part, float* __restrict__ dbias, int biasParts) {
    const size_t total = (size_t)Cout * Cin * T;
    const size_t slabElems = (size_t)nChunks * T * CoutPad * CK;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        // iterate in slab-friendly order: i -> (tap, co, ci) with ci fastest
        const int ci = (int)(i % Cin);
        size_t r = i / Cin;
        const int co = (int)(r % Cout);
        const int tap = (int)(r / Cout);
        const float* src = slabs + (((size_t)(ci / CK) * T + tap) * CoutPad + co) * CK + (ci % CK);
        float s = 0.f;
        int k = 0;
        for (; k + 4 <= ksplit; k += 4) {          // 4 independent loads in flight, summation order unchanged
            const float a0 = src[(size_t)k * slabElems], a1 = src[(size_t)(k + 1) * slabElems];
            const float a2 = src[(size_t)(k + 2) * slabElems], a3 = src[(size_t)(k + 3) * slabElems];
            s += a0; s += a1; s += a2; s += a3;
        }
        for (; k < ksplit; ++k) s += src[(size_t)k * slabElems];
        dw[((size_t)co * Cin + ci) * T + tap] = s;
    }
    if (dbias)
        for (size_t c = blockIdx.x * (size_t)blockDim.x + threadIdx.x; c < (size_t)Cout; c += (size_t)gridDim.x * blockDim.x) {
            float s = 0.f;
            for (int y = 0; y < biasParts; ++y) s += bias_part[(size_t)y * CoutPad + c];
            dbias[c] = s;
        }
}

// Split-K sum of the version-3 weight gradient (conv_wgrad.hip): its slabs are already in dw[co][ci][tap] order, one per
// workgroup (128 for a 64 -> 64 layer), so this is a streaming sum.  A 512-thread block owns 64 float4 of dW; its 8 waves each sum a
// contiguous eighth of the slabs with eight 1-KiB loads in flight, the eighths are combined in a fixed order through LDS.
// Deterministic: the order of every addition is fixed by (ksplit, position) alone.
__global__ __launch_bounds__(512) void conv_reduce_dw3_kernel(const float* __restrict__ slabs, float* __restrict__ dw, size_t n4,
                                                              int ksplit, int Cout, int CoutPad, const float* __restrict__ bias_part,
                                                              float* __restrict__ dbias, int biasParts) {
    __shared__ float4 part[8][64];
    const int lane = threadIdx.x & 63, sg = threadIdx.x >> 6;
    const size_t e = (size_t)blockIdx.x * 64 + lane;
    const int k0 = sg * ksplit / 8, k1 = (sg + 1) * ksplit / 8;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e < n4) {
        const float4* p = reinterpret_cast<const float4*>(slabs) + e;
        int k = k0;
        for (; k + 8 <= k1; k += 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(k + u) * n4];
#pragma unroll
            for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
        }
        for (; k < k1; ++k) {
            const float4 v = p[(size_t)k * n4];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    part[sg][lane] = s;
    __syncthreads();
    if (sg == 0 && e < n4) {
        float4 v = part[0][lane];
#pragma unroll
        for (int q = 1; q < 8; ++q) { const float4 u = part[q][lane]; v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w; }
        reinterpret_cast<float4*>(dw)[e] = v;
    }
    // bias gradient: sum of the per-slice partial rows.  One block, but every row is a dependent-latency load if a thread walks
    // them one by one (128 rows: 28 us, three times the slab sum above) -- the 8 waves take an eighth of the rows each, 8 loads in flight.
    if (dbias && blockIdx.x == gridDim.x - 1) {
        __shared__ float bpart[8][64];
        for (int c0 = 0; c0 < Cout; c0 += 64) {
            const int c = c0 + lane;
            const int y0 = sg * biasParts / 8, y1 = (sg + 1) * biasParts / 8;
            float sb = 0.f;
            if (c < Cout) {
                int y = y0;
                for (; y + 8 <= y1; y += 8) {
                    float v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = bias_part[(size_t)(y + u) * CoutPad + c];
#pragma unroll
                    for (int u = 0; u < 8; ++u) sb += v[u];
                }
                for (; y < y1; ++y) sb += bias_part[(size_t)y * CoutPad + c];
            }
            __syncthreads();
            bpart[sg][lane] = sb;
            __syncthreads();
            if (sg == 0 && c < Cout) {
                float v = bpart[0][lane];
#pragma unroll
                for (int q = 1; q < 8; ++q) v += bpart[q][lane];
                dbias[c] = v;
            }
        }
    }
}

// column sums: out[c] = sum_rows x[row][c]; stage 1 -> partial[block][C], stage 2 -> out
__global__ __launch_bounds__(256) void colsum_stage1_kernel(const float* __restrict__ x,
                                                            float* __restrict__ partial, size_t rows, int C) {
    const size_t rowsPer = (rows + gridDim.x - 1) / gridDim.x;
    const size_t r0 = blockIdx.x * rowsPer;
    size_t r1 = r0 + rowsPer;
    if (r1 > rows) r1 = rows;
    // thread t owns channels t, t+256, ... when C > 256; for C <= 256 several rows go in parallel
    extern __shared__ float sh[];
    if (C <= 256) {
        const int rpar = 256 / C;                 // rows in flight
        const int c = threadIdx.x % C, rr = threadIdx.x / C;
        float s = 0.f;
        if (rr < rpar)
            for (size_t r = r0 + rr; r < r1; r += rpar) s += x[r * C + c];
        sh[threadIdx.x] = (rr < rpar) ? s : 0.f;
        __syncthreads();
        if (threadIdx.x < C) {
            float t = 0.f;
            for (int k = 0; k < rpar; ++k) t += sh[k * C + threadIdx.x];
            partial[(size_t)blockIdx.x * C + threadIdx.x] = t;
        }
    } else {
        for (int c = threadIdx.x; c < C; c += 256) {
            float s = 0.f;
            for (size_t r = r0; r < r1; ++r) s += x[r * C + c];
            partial[(size_t)blockIdx.x * C + c] = s;
        }
    }
}
__global__ __launch_bounds__(256) void colsum_stage2_kernel(const float* __restrict__ partial, float* __restrict__ out,
                                                            int nblk, int C) {
    // one wave per channel, lanes over the partial blocks
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= C) return;
    float s = 0.f;
    for (int k = lane; k < nblk; k += 64) s += partial[(size_t)k * C + c];
    s = wave_sum(s);
    if (lane == 0) out[c] = s;
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static void choose_tile(int Do, int Ho, int Wo, int kd, int kh, int kw, int& TD, int& TH, int& TW) {
    if (kd == 1 && kh == 1 && kw == 1) { TD = 1; TH = 1; TW = 128; return; }   // caller flattens voxels
    // candidates with TD*TH*TW == 128; minimise halo volume x tile count (wasted lanes on ragged edges)
    static const int cand[][3] = {{2, 8, 8}, {8, 4, 4}, {4, 4, 8}, {4, 8, 4}, {1, 8, 16}, {1, 16, 8}, {16, 4, 2},
                                  {32, 2, 2}, {8, 8, 2}, {2, 4, 16}, {128, 1, 1}, {1, 1, 128}, {16, 8, 1}, {1, 2, 64}};
    double best = 1e300;
    for (auto& c : cand) {
        if (c[2] & 1) continue;   // bwd-weight pairs voxels along W
        const double tiles = (double)cdiv(Do, c[0]) * cdiv(Ho, c[1]) * cdiv(Wo, c[2]);
        const double halo = (double)(c[0] + kd - 1) * (c[1] + kh - 1) * (c[2] + kw - 1);
        if (halo * (LDSROW + 1) * 4 + 2 * NT * LDSROW * 4 + 512 > 150 * 1024) continue;
        const double cost = tiles * (halo * 0.15 + 128.0 * kd * kh * kw);   // staging + MFMA work
        if (cost < best) { best = cost; TD = c[0]; TH = c[1]; TW = c[2]; }
    }
}

static int make_geom(ConvGeom& g, int B, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw,
                     int pd, int ph, int pw, int epd = 0, int eph = 0, int epw = 0) {
    DIQT_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, DIQT_E_SHAPE, "conv3d: non-positive extent");
    DIQT_REQUIRE(kd > 0 && kh > 0 && kw > 0 && pd >= 0 && ph >= 0 && pw >= 0, DIQT_E_SHAPE, "conv3d: bad filter/pad");
    DIQT_REQUIRE(pd + epd >= 0 && ph + eph >= 0 && pw + epw >= 0, DIQT_E_SHAPE, "conv3d: negative high-side pad");
    if (kd == 1 && kh == 1 && kw == 1 && pd == 0 && ph == 0 && pw == 0 && epd == 0 && eph == 0 && epw == 0) {   // 1x1x1: flatten all voxels into W
        const long long rows = (long long)B * D * H * W;
        DIQT_REQUIRE(rows < (1ll << 31), DIQT_E_SHAPE, "conv3d: too many rows");
        B = 1; D = 1; H = 1; W = (int)rows;
    }
    g.dbg = nullptr;
    g.subF = 0;
    g.B = B; g.D = D; g.H = H; g.W = W; g.Cin = Cin; g.Cout = Cout;
    g.kd = kd; g.kh = kh; g.kw = kw; g.pd = pd; g.ph = ph; g.pw = pw;
    g.Do = D + 2 * pd + epd - kd + 1; g.Ho = H + 2 * ph + eph - kh + 1; g.Wo = W + 2 * pw + epw - kw + 1;
    DIQT_REQUIRE(g.Do > 0 && g.Ho > 0 && g.Wo > 0, DIQT_E_SHAPE, "conv3d: empty output");
    g.TD = 2; g.TH = 8; g.TW = 8;
    choose_tile(g.Do, g.Ho, g.Wo, kd, kh, kw, g.TD, g.TH, g.TW);
    g.tilesD = cdiv(g.Do, g.TD); g.tilesH = cdiv(g.Ho, g.TH); g.tilesW = cdiv(g.Wo, g.TW);
    g.nNt = cdiv(Cout, NT); g.CoutPad = g.nNt * NT; g.nChunks = cdiv(Cin, CK);
    g.chunksPerSplit = g.nChunks; g.slabStride = 0; g.xBytes = 0; g.yBytes = 0; g.tilesPerWg = 1; g.stats = nullptr;
    { static const int stg = [] { const char* e = getenv("DIQT_CONV_STAGGER"); return e ? atoi(e) : 0; }(); g.stagger = stg; }
    g.HD = g.TD + kd - 1; g.HH = g.TH + kh - 1; g.HWd = g.TW + kw - 1;
    const long long nwg = (long long)g.B * g.tilesD * g.tilesH * g.tilesW * g.nNt;
    DIQT_REQUIRE(nwg < (1ll << 31), DIQT_E_SHAPE, "conv3d: grid too large");
    DIQT_REQUIRE((long long)g.B * g.Do * g.Ho * g.Wo < (1ll << 31), DIQT_E_SHAPE, "conv3d: too many output voxels");
    return DIQT_OK;
}

}  // namespace diqt

using namespace diqt;

static unsigned long long* g_dbg_ptr = nullptr;   // diagnostic cycle-stamp buffer of the last DIQT_CONV_DBG=1 launch
static unsigned g_dbg_n = 0;

// diagnostic only (not part of include/diqt.h): copies the cycle stamps of the last DIQT_CONV_DBG=1 launch to the host
extern "C" int diqt_debug_wgrad3_stamps(unsigned long long* host_out, unsigned max_waves) {
    if (!wgrad3_dbg_ptr || !wgrad3_dbg_n) return 0;
    const unsigned n = wgrad3_dbg_n < max_waves ? wgrad3_dbg_n : max_waves;
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(host_out, wgrad3_dbg_ptr, (size_t)n * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    return (int)n;
}

extern "C" int diqt_debug_conv_stamps(unsigned long long* host_out, unsigned max_wg) {
    if (!g_dbg_ptr || !g_dbg_n) return 0;
    const unsigned n = g_dbg_n < max_wg ? g_dbg_n : max_wg;
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(host_out, g_dbg_ptr, (size_t)n * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    return (int)n;
}

extern "C" size_t diqt_conv_packed_elems(int Cout, int Cin, int kd, int kh, int kw) {
    if (Cout <= 0 || Cin <= 0 || kd <= 0 || kh <= 0 || kw <= 0) return 0;
    return (size_t)cdiv(Cin, CK) * kd * kh * kw * (cdiv(Cout, NT) * NT) * CK;
}

extern "C" long long diqt_conv3d_lds_bytes(int D, int H, int W, int kd, int kh, int kw, int pd, int ph, int pw,
                                           int epd, int eph, int epw) {
    ConvGeom g;
    if (make_geom(g, 1, D, H, W, 4, 4, kd, kh, kw, pd, ph, pw, epd, eph, epw)) return -1;
    return (long long)(((size_t)g.HD * g.HH * g.HWd * (LDSROW + 1) + 2 * NT * LDSROW) * sizeof(float) + MTILE * sizeof(int));
}

// forward convs with <= 4 input channels and more than one tap use the tap-packed kernel (and its weight packing)
static inline int smallcin_pad(int Cin, int T) {
    static const bool off = [] { const char* e = getenv("DIQT_CONV_NOSMALLCIN"); return e && e[0] == '1'; }();
    if (off || Cin > 4 || T < 2) return 0;
    return Cin == 3 ? 4 : Cin;
}

extern "C" int diqt_conv_pack_weight(const float* w, float* packed, int Cout, int Cin, int kd, int kh, int kw,
                                     int mode, void* stream) {
    DIQT_REQUIRE(w && packed, DIQT_E_ALIGN, "conv_pack_weight: null pointer");
    DIQT_REQUIRE(Cout > 0 && Cin > 0 && kd > 0 && kh > 0 && kw > 0 && (mode == 0 || mode == 1), DIQT_E_SHAPE,
                 "conv_pack_weight: bad shape/mode");
    const int T = kd * kh * kw;
    if (const int CINP = smallcin_pad(mode == 0 ? Cin : Cout, T)) {
        const int CoutPadE = cdiv(mode == 0 ? Cout : Cin, NT) * NT, nCh = cdiv(T * CINP, CK);
        const size_t total = (size_t)nCh * CoutPadE * CK;
        hipLaunchKernelGGL(conv_pack_smallcin_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, w, packed,
                           Cout, Cin, T, CINP, CoutPadE, mode, total);
        return check_launch("conv_pack_weight(small Cin)");
    }
    const int outEff = mode == 0 ? Cout : Cin, inEff = mode == 0 ? Cin : Cout;
    const int CoutPadEff = cdiv(outEff, NT) * NT, nChunksEff = cdiv(inEff, CK);
    const size_t total = (size_t)nChunksEff * T * CoutPadEff * CK;
    hipLaunchKernelGGL(conv_pack_weight_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, w,
                       packed, Cout, Cin, T, mode, CoutPadEff, nChunksEff, total);
    return check_launch("conv_pack_weight");
}

// forward split-K plan: launches with fewer than ~1.5 resident rounds' worth of workgroups slice the Cin chunks
static int fwd_ksplit(const ConvGeom& g) {
    static const bool off = [] { const char* e = getenv("DIQT_CONV_NOSPLIT"); return e && e[0] == '1'; }();
    const long long nwg = (long long)g.B * g.tilesD * g.tilesH * g.tilesW * g.nNt;
    if (off || nwg >= 384 || g.nChunks < 2) return 1;
    int ks = (int)(512 / nwg);
    if (ks > g.nChunks) ks = g.nChunks;
    if (ks > 16) ks = 16;
    if (ks < 1) ks = 1;
    const int per = cdiv(g.nChunks, ks);
    return cdiv(g.nChunks, per);
}

// y[i] = sum_s slab[s][i] + bias[i % Cout] + residual[i]
__global__ __launch_bounds__(256) void conv_fwd_reduce_kernel(const float* __restrict__ slabs, const float* __restrict__ bias,
                                                              const float* __restrict__ residual, float* __restrict__ y,
                                                              size_t n, int Cout, int ks) {
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float s = slabs[i];
        for (int k = 1; k < ks; ++k) s += slabs[(size_t)k * n + i];
        if (bias) s += bias[i % Cout];
        if (residual) s += residual[i];
        y[i] = s;
    }
}

extern "C" size_t diqt_conv3d_fwd_workspace_bytes(int B, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw,
                                                  int pd, int ph, int pw, int epd, int eph, int epw) {
    ConvGeom g;
    if (make_geom(g, B, D, H, W, Cin, Cout, kd, kh, kw, pd, ph, pw, epd, eph, epw)) return 0;
    const int ks = fwd_ksplit(g);
    size_t need = ks > 1 ? (size_t)ks * g.B * g.Do * g.Ho * g.Wo * g.Cout * sizeof(float) : 0;
    {
        F9Geom g9;
        size_t l9;
        unsigned gr9;
        if (Cin % 4 == 0 && !smallcin_pad(Cin, kd * kh * kw) &&
            fwd9_plan(g9, l9, gr9, B, D, H, W, Cin, Cout, kd, kh, kw, pd, ph, pw, epd, eph, epw, diqt_conv_packed_elems(Cout, Cin, kd, kh, kw), true) &&
            g9.ksplit > 1) {
            const size_t n9 = (size_t)g9.ksplit * g.B * g.Do * g.Ho * g.Wo * g.Cout * sizeof(float);
            if (n9 > need) need = n9;
        }
    }
    return need;
}

static int conv3d_fwd_impl(const float* x, const float* packed, const float* bias, const float* residual,
                           float* y, void* workspace, size_t workspace_bytes, int B, int D, int H, int W, int Cin, int Cout,
                           int kd, int kh, int kw, int pd, int ph, int pw, int epd, int eph, int epw, void* stream, float* stats);

extern "C" int diqt_conv3d_fwd(const float* x, const float* packed, const float* bias, const float* residual,
                               float* y, int B, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw,
                               int pd, int ph, int pw, int epd, int eph, int epw, void* stream) {
    return conv3d_fwd_impl(x, packed, bias, residual, y, nullptr, 0, B, D, H, W, Cin, Cout, kd, kh, kw, pd, ph, pw, epd, eph,
                           epw, stream, nullptr);
}

// 8-wave forward plan: the geometry re-tiled for 256-voxel workgroups, or false when conv_fwd8_kernel does not take the launch
// (fewer than two resident rounds of 256-voxel tiles, halo beyond the prefetch registers or the LDS, tensors >= 1 GiB, Cin % 4)
static bool fwd8_plan(const ConvGeom& g, ConvGeom& g8, size_t& lds) {
    static const int mode = [] { const char* e = getenv("DIQT_CONV_W8"); return e ? atoi(e) : 1; }();      // 0: never
    const int T = g.kd * g.kh * g.kw;
    // measured: +1.2 % on 27-tap filters, -3.5 % on the 9-tap (1,3,3) filters of the pseudo-3D blocks (a chunk there is 3 steps long)
    if (!mode || (T < 12 && mode != 2) || T < 2 || g.Cin % 4 != 0 || smallcin_pad(g.Cin, T)) return false;
    const unsigned long long xb = (unsigned long long)g.B * g.D * g.H * g.W * g.Cin * 4ull;
    const unsigned long long yb = (unsigned long long)g.B * g.Do * g.Ho * g.Wo * g.Cout * 4ull;
    if (xb >= (1ull << 30) || yb >= (1ull << 30)) return false;
    static const int cand[][3] = {{4, 8, 8}, {8, 8, 4}, {8, 4, 8}, {2, 8, 16}, {2, 16, 8}, {1, 16, 16}, {16, 4, 4}, {4, 4, 16}, {4, 16, 4},
                                  {1, 8, 32}, {1, 32, 8}, {32, 4, 2}, {64, 2, 2}, {256, 1, 1}, {1, 1, 256}};
    double best = 1e300;
    bool found = false;
    g8 = g;
    for (auto& c : cand) {
        const long long hv = (long long)(c[0] + g.kd - 1) * (c[1] + g.kh - 1) * (c[2] + g.kw - 1);
        if (hv * 8 > 512 * F8_HREG) continue;
        const size_t l = ((size_t)hv * LDSROW + (size_t)F8_NWB * F8_TG * NT * LDSROW) * sizeof(float) + 2 * (F8_MT + (size_t)hv) * sizeof(int);
        if (l > 160 * 1024) continue;
        const double tiles = (double)cdiv(g.Do, c[0]) * cdiv(g.Ho, c[1]) * cdiv(g.Wo, c[2]);
        const double cost = tiles * ((double)hv * 0.15 + 256.0 * T);
        if (cost < best) { best = cost; g8.TD = c[0]; g8.TH = c[1]; g8.TW = c[2]; lds = l; found = true; }
    }
    if (!found) return false;
    g8.tilesD = cdiv(g.Do, g8.TD); g8.tilesH = cdiv(g.Ho, g8.TH); g8.tilesW = cdiv(g.Wo, g8.TW);
    g8.HD = g8.TD + g.kd - 1; g8.HH = g8.TH + g.kh - 1; g8.HWd = g8.TW + g.kw - 1;
    const long long nwg8 = (long long)g.B * g8.tilesD * g8.tilesH * g8.tilesW * g.nNt;
    // one workgroup per CU: the launch has to fill whole rounds of 256 (128->192 @ 8x16^3 = 384 workgroups = 1.5 rounds ran 26 % slower
    // than on the 4-wave kernel, whose two workgroups per CU halve the granularity); mode 2: always
    if (mode != 2 && (nwg8 < 256 || (double)nwg8 / (double)((nwg8 + 255) / 256 * 256) < 0.94)) return false;
    g8.xBytes = (unsigned)xb; g8.yBytes = (unsigned)yb;
    return true;
}

// per-tile output statistics are produced by the buffer-path kernel of an unsplit launch; returns the number of tiles per batch
// entry (the `nblk` of the [B][nblk][2][Cout] partial layout) or 0 when this shape would take another path
static int fwd_stats_blocks_impl(int B, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw, int pd, int ph, int pw, int epd,
                                 int eph, int epw, bool neighbours);
extern "C" int diqt_conv3d_fwd_stats_blocks(int B, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw, int pd, int ph,
                                            int pw, int epd, int eph, int epw) {
    return fwd_stats_blocks_impl(B, D, H, W, Cin, Cout, kd, kh, kw, pd, ph, pw, epd, eph, epw, false);
}
// the same for diqt_conv3d_fwd_neighbours (which never takes conv_fwd9_kernel: that kernel has no neighbour addressing)
extern "C" int diqt_conv3d_fwd_neighbours_stats_blocks(int f, int A, int Cin, int Cout, int k) {
    if (f < 1 || A < 1 || k < 1 || !(k & 1)) return 0;
    return fwd_stats_blocks_impl(f * f * f, A, A, A, Cin, Cout, k, k, k, k / 2, k / 2, k / 2, 0, 0, 0, true);
}
static int fwd_stats_blocks_impl(int B, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw, int pd, int ph, int pw, int epd,
                                 int eph, int epw, bool neighbours) {
    static const bool off = [] { const char* e = getenv("DIQT_CONV_NOSTATS"); return e && e[0] == '1'; }();
    static const bool nobuf = [] { const char* e = getenv("DIQT_CONV_NOBUF"); return e && e[0] == '1'; }();
    ConvGeom g;
    if (off || nobuf || make_geom(g, B, D, H, W, Cin, Cout, kd, kh, kw, pd, ph, pw, epd, eph, epw)) return 0;
    if (kd * kh * kw == 1) return 0;                                   // flattened 1x1x1 tiles cross batch entries
    if (Cin % 4 != 0 || smallcin_pad(Cin, kd * kh * kw) || fwd_ksplit(g) > 1) return 0;
    const unsigned long long xb = (unsigned long long)g.B * g.D * g.H * g.W * g.Cin * 4ull;
    const unsigned long long yb = (unsigned long long)g.B * g.Do * g.Ho * g.Wo * g.Cout * 4ull;
    if (xb >= (1ull << 30) || yb >= (1ull << 30)) return 0;
    const size_t lds = ((size_t)g.HD * g.HH * g.HWd * (LDSROW + 1) + 2 * NT * LDSROW) * sizeof(float) + MTILE * sizeof(int);
    if (lds > 160 * 1024) return 0;
    ConvGeom g8;
    size_t lds8;
    {
        F9Geom g9;
        size_t l9;
        unsigned gr9;
        if (!neighbours && Cin % 4 == 0 && fwd_ksplit(g) <= 1 && !smallcin_pad(Cin, kd * kh * kw) &&
            fwd9_plan(g9, l9, gr9, B, D, H, W, Cin, Cout, kd, kh, kw, pd, ph, pw, epd, eph, epw, diqt_conv_packed_elems(Cout, Cin, kd, kh, kw), false))
            return g9.tilesD * g9.tilesH * g9.tilesW;                          // conv_fwd9_kernel: one row per 512- / 256-voxel tile
    }
    if (fwd8_plan(g, g8, lds8)) return g8.tilesD * g8.tilesH * g8.tilesW;      // the 8-wave kernel writes one row per 256-voxel tile
    return g.tilesD * g.tilesH * g.tilesW;
}

// which kernel diqt_conv3d_fwd* dispatches this shape to (for profilers / bench.py, so that per-kernel numbers carry the names
// rocprofv3 reports): 0 conv_fwd_kernel, 1 conv_fwd_smallcin_kernel, 2 conv1x1_fwd_kernel, 3 conv_fwd8_kernel, 4 conv_fwd9_kernel, -1 bad shape
extern "C" int diqt_conv3d_fwd_kernel_id(int B, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw, int pd, int ph, int pw,
                                         int epd, int eph, int epw) {
    ConvGeom g;
    if (make_geom(g, B, D, H, W, Cin, Cout, kd, kh, kw, pd, ph, pw, epd, eph, epw)) return -1;
    const int T = kd * kh * kw;
    if (smallcin_pad(Cin, T)) return 1;
    const unsigned long long xb = (unsigned long long)g.B * g.D * g.H * g.W * g.Cin * 4ull;
    const unsigned long long yb = (unsigned long long)g.B * g.Do * g.Ho * g.Wo * g.Cout * 4ull;
    const bool buf = Cin % 4 == 0 && xb < (1ull << 30) && yb < (1ull << 30);
    if (!buf) return 0;
    {
        F9Geom g9;
        size_t l9;
        unsigned gr9;
        if (fwd9_plan(g9, l9, gr9, B, D, H, W, Cin, Cout, kd, kh, kw, pd, ph, pw, epd, eph, epw, diqt_conv_packed_elems(Cout, Cin, kd, kh, kw), true))
            return 4;                // incl. its split-K form (callers that pass the workspace diqt_conv3d_fwd_workspace_bytes asks for)
    }
    if (fwd_ksplit(g) > 1) return 0;
    ConvGeom g8;
    size_t lds8;
    if (fwd8_plan(g, g8, lds8)) return 3;
    if (T == 1 && g.B == 1 && g.D == 1 && g.H == 1 && g.Wo == g.W && g.TW == MTILE) return 2;
    return 0;
}

extern "C" int diqt_conv3d_fwd_ex(const float* x, const float* packed, const float* bias, const float* residual, float* y,
                                  float* stats, void* workspace, size_t workspace_bytes, int B, int D, int H, int W, int Cin,
                                  int Cout, int kd, int kh, int kw, int pd, int ph, int pw, int epd, int eph, int epw, void* stream) {
    DIQT_REQUIRE(!stats || diqt_conv3d_fwd_stats_blocks(B, D, H, W, Cin, Cout, kd, kh, kw, pd, ph, pw, epd, eph, epw) > 0,
                 DIQT_E_UNSUPPORTED, "conv3d_fwd_ex: this shape does not produce output statistics (diqt_conv3d_fwd_stats_blocks == 0)");
    return conv3d_fwd_impl(x, packed, bias, residual, y, workspace, workspace_bytes, B, D, H, W, Cin, Cout, kd, kh, kw, pd, ph, pw,
                           epd, eph, epw, stream, stats);
}

// The backward-data pass of a conv that sits behind a fused GroupNorm + scale/shift + Mish/SiLU (Block.forward: GN -> act -> conv,
// imagen_pytorch3D.py:535-566 / imagen_video.py:671-697): y = conv(x = dY of the conv, flipped packed weights) is the gradient w.r.t.
// the activated tensor, and the epilogue of conv_fwd9_kernel -- which holds that gradient in registers -- also reads the GroupNorm
// input gn_x at the same voxels and writes the per-tile partial sums of the GroupNorm backward (sum dz, sum dz xhat per channel):
// partials[B][nblk][2][Cout], nblk = diqt_conv3d_fwd_gnbwd_blocks(...) (0: this shape does not run on conv_fwd9_kernel un-split; use
// diqt_conv3d_fwd + diqt_gn_act_bwd).  diqt_gn_act_bwd_from_partials finishes the GroupNorm backward without its reduction pass.
// Process-wide switch of that fusion: -1 = not set yet (the first query reads DIQT_GNBWD_FUSE, default off), 0 / 1 = set by the caller.
// Both modes are product paths (the parity suite runs the whole-network gradient tests in each, tests/test_gpu_fullsize.py).
static std::atomic<int> g_gnbwd_fuse{-1};
extern "C" int diqt_get_gnbwd_fuse(void) {
    int v = g_gnbwd_fuse.load(std::memory_order_relaxed);
    if (v < 0) {
        const char* e = getenv("DIQT_GNBWD_FUSE");
        v = (e && e[0] == '1') ? 1 : 0;
        g_gnbwd_fuse.store(v, std::memory_order_relaxed);
    }
    return v;
}
extern "C" int diqt_set_gnbwd_fuse(int on) {
    const int prev = diqt_get_gnbwd_fuse();
    g_gnbwd_fuse.store(on ? 1 : 0, std::memory_order_relaxed);
    return prev;
}
extern "C" int diqt_conv3d_fwd_gnbwd_blocks(int B, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw, int pd, int ph, int pw,
                                            int epd, int eph, int epw) {
    // Opt-in (DIQT_GNBWD_FUSE=1 or diqt_set_gnbwd_fuse(1)): the epilogue saves the GroupNorm backward's reduction pass (35 us and 134 MB per
    // GroupNorm at the 32^3 level) but costs the launch as much: the activation derivative per output element (exp + two reciprocals at
    // quarter rate) is ~8.6k instructions per tile on a kernel with one wave per SIMD and nothing to overlap them with.  42.1-42.3 ms
    // per training micro-step either way, A/B on one box.
    const bool off = !diqt_get_gnbwd_fuse();
    F9Geom g9;
    size_t l9;
    unsigned gr9;
    if (off || Cin % 4 != 0 || smallcin_pad(Cin, kd * kh * kw) || (kd == 3 && kh == 1 && kw == 1)) return 0;     // (3,1,1): no such instantiation
    if (!fwd9_plan(g9, l9, gr9, B, D, H, W, Cin, Cout, kd, kh, kw, pd, ph, pw, epd, eph, epw, diqt_conv_packed_elems(Cout, Cin, kd, kh, kw), false))
        return 0;
    return g9.tilesD * g9.tilesH * g9.tilesW;
}
extern "C" int diqt_conv3d_fwd_gnbwd(const float* x, const float* packed, float* y, float* partials, const float* gn_x, const float* mean,
                                     const float* rstd, const float* gamma, const float* beta, const float* scale, const float* shift,
                                     int cond_stride, int G, int act, int B, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw,
                                     int pd, int ph, int pw, int epd, int eph, int epw, void* stream) {
    DIQT_REQUIRE(x && packed && y && partials && gn_x && mean && rstd, DIQT_E_ALIGN, "conv3d_fwd_gnbwd: null pointer");
    DIQT_REQUIRE(aligned16(x) && aligned16(packed), DIQT_E_ALIGN, "conv3d_fwd_gnbwd: x and packed weights must be 16-byte aligned");
    DIQT_REQUIRE(G > 0 && Cout % G == 0 && (act == DIQT_ACT_MISH || act == DIQT_ACT_SILU), DIQT_E_UNSUPPORTED,
                 "conv3d_fwd_gnbwd: Mish / SiLU, groups dividing the channels");
    DIQT_REQUIRE((scale == nullptr) == (shift == nullptr) && (!scale || cond_stride >= Cout), DIQT_E_SHAPE, "conv3d_fwd_gnbwd: scale / shift");
    F9Geom g9;
    size_t l9;
    unsigned gr9;
    DIQT_REQUIRE(diqt_conv3d_fwd_gnbwd_blocks(B, D, H, W, Cin, Cout, kd, kh, kw, pd, ph, pw, epd, eph, epw) > 0 &&
                     fwd9_plan(g9, l9, gr9, B, D, H, W, Cin, Cout, kd, kh, kw, pd, ph, pw, epd, eph, epw,
                               diqt_conv_packed_elems(Cout, Cin, kd, kh, kw), false),
                 DIQT_E_UNSUPPORTED, "conv3d_fwd_gnbwd: shape not taken by conv_fwd9_kernel (diqt_conv3d_fwd_gnbwd_blocks == 0)");
    g9.stats = partials;
    g9.gx = gn_x; g9.gmean = mean; g9.grstd = rstd; g9.ggamma = gamma; g9.gbeta = beta; g9.gscale = scale; g9.gshift = shift;
    g9.gG = G; g9.gcs = cond_stride; g9.gact = act;
    // the epilogue's parameters go through a small device ring (stream-ordered copy: every launch gets its own slot, 256 launches
    // deep), so the kernel carries ONE pointer for them through its main loop
    static F9GnParams* ring = nullptr;
    static unsigned slot = 0;
    if (!ring) {
        hipError_t e = hipMalloc(&ring, 256 * sizeof(F9GnParams));
        DIQT_REQUIRE(e == hipSuccess, DIQT_E_LAUNCH, "conv3d_fwd_gnbwd: hipMalloc: %s", hipGetErrorString(e));
    }
    const F9GnParams hp{mean, rstd, gamma, beta, scale, shift, G, cond_stride, act, 0};
    F9GnParams* dp = ring + (slot++ & 255u);
    hipError_t e = hipMemcpyAsync(dp, &hp, sizeof(hp), hipMemcpyHostToDevice, (hipStream_t)stream);
    DIQT_REQUIRE(e == hipSuccess, DIQT_E_LAUNCH, "conv3d_fwd_gnbwd: hipMemcpyAsync: %s", hipGetErrorString(e));
    g9.gnp = dp;
    return fwd9_launch(x, packed, nullptr, nullptr, y, g9, l9, gr9, stream);
}

static int conv3d_fwd_one(const float* x, const float* packed, const float* bias, const float* residual,
                          float* y, void* workspace, size_t workspace_bytes, int B, int D, int H, int W, int Cin, int Cout,
                          int kd, int kh, int kw, int pd, int ph, int pw, int epd, int eph, int epw, void* stream, float* stats,
                          int subF = 0);

// Block.forward on the sampling path (GroupNorm -> (scale + 1) x + shift -> Mish / SiLU -> conv; imagen_pytorch3D.py:546-566,
// imagen_video.py:680-697) as ONE launch: x is the RAW GroupNorm input and conv_fwd9_kernel's GroupNorm-apply instantiation rewrites
// every halo piece in the LDS as act(A x + Bc) right after its DMA landed -- the elementwise pass over the activation (a read and a
// write of the whole tensor per conv) is gone.  coef[2][B][Cin] = (A, Bc) from diqt_gn_coef_from_partials / diqt_gn_coef.
// diqt_conv3d_fwd_gn_supported: 1 when conv_fwd9_kernel takes the launch (given the workspace diqt_conv3d_fwd_workspace_bytes asks
// for) and has the instantiation for this filter and activation; otherwise run diqt_gn_act_fwd + diqt_conv3d_fwd_ex.
extern "C" int diqt_conv3d_fwd_gn_supported(int B, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw, int pd, int ph, int pw,
                                            int epd, int eph, int epw, int act) {
    static const bool off = [] { const char* e = getenv("DIQT_CONV_NOGNA"); return e && e[0] == '1'; }();       // A/B switch
    F9Geom g9;
    size_t l9;
    unsigned gr9;
    if (off || Cin % 4 != 0 || smallcin_pad(Cin, kd * kh * kw)) return 0;
    if (!fwd9_plan(g9, l9, gr9, B, D, H, W, Cin, Cout, kd, kh, kw, pd, ph, pw, epd, eph, epw, diqt_conv_packed_elems(Cout, Cin, kd, kh, kw), true))
        return 0;
    if (!fwd9_gna_available(g9.variant, act)) return 0;
    // Every 64-channel output block of a tile rewrites its own copy of the halo, so the rewrite grows with Cout / 64 while the pass it
    // replaces does not.  Measured on MI355X (tools/conv_bench.py gn, us saved per launch): 3x3x3 64->64 @ 8x32^3 +7, 128->64 +12,
    // 128->128 @ 8x16^3 +2, 192->128 +3, split-K 256->256 @ 8x8^3 +4 (the saved launch is latency-bound there), but 256->128 @ 8x16^3
    // -6, 256->256 @ 32^3 -23, 512->512 @ 16^3 -6; (1,3,3): 64->64 @ 32x32 frames +19, 128->128 +10, 256->256 @ 8x8 +5.
    if (kd == 1) return g9.nNt <= 4 ? 1 : 0;
    return (g9.nNt == 1 || (g9.nNt == 2 && Cin <= 192) || g9.ksplit > 1) ? 1 : 0;
}
extern "C" int diqt_conv3d_fwd_gn(const float* x, const float* packed, const float* bias, const float* residual, float* y, float* stats,
                                  void* workspace, size_t workspace_bytes, const float* coef, int act, int B, int D, int H, int W, int Cin,
                                  int Cout, int kd, int kh, int kw, int pd, int ph, int pw, int epd, int eph, int epw, void* stream) {
    DIQT_REQUIRE(x && packed && y && coef, DIQT_E_ALIGN, "conv3d_fwd_gn: null pointer");
    DIQT_REQUIRE(aligned16(x) && aligned16(packed) && aligned16(coef), DIQT_E_ALIGN, "conv3d_fwd_gn: x, packed weights and coef must be 16-byte aligned");
    DIQT_REQUIRE(diqt_conv3d_fwd_gn_supported(B, D, H, W, Cin, Cout, kd, kh, kw, pd, ph, pw, epd, eph, epw, act), DIQT_E_UNSUPPORTED,
                 "conv3d_fwd_gn: shape / activation not taken (diqt_conv3d_fwd_gn_supported == 0)");
    F9Geom g9;
    size_t l9;
    unsigned gr9;
    const bool maySplit = workspace && aligned16(workspace);
    const bool ok = fwd9_plan(g9, l9, gr9, B, D, H, W, Cin, Cout, kd, kh, kw, pd, ph, pw, epd, eph, epw,
                              diqt_conv_packed_elems(Cout, Cin, kd, kh, kw), maySplit);
    DIQT_REQUIRE(ok && fwd9_gna_available(g9.variant, act), DIQT_E_WORKSPACE,
                 "conv3d_fwd_gn: this launch needs the split-K workspace of diqt_conv3d_fwd_workspace_bytes");
    g9.gcoef = coef;
    g9.gnaAct = act;
    if (g9.ksplit == 1) {
        DIQT_REQUIRE(!stats || diqt_conv3d_fwd_stats_blocks(B, D, H, W, Cin, Cout, kd, kh, kw, pd, ph, pw, epd, eph, epw) > 0, DIQT_E_UNSUPPORTED,
                     "conv3d_fwd_gn: this shape does not produce output statistics");
        g9.stats = stats;
        return fwd9_launch(x, packed, bias, residual, y, g9, l9, gr9, stream);
    }
    const size_t n = (size_t)B * g9.Do * g9.Ho * g9.Wo * Cout;
    DIQT_REQUIRE(!stats && workspace_bytes >= (size_t)g9.ksplit * n * sizeof(float), DIQT_E_WORKSPACE,
                 "conv3d_fwd_gn: split-K launch: no statistics, workspace of %zu bytes", (size_t)g9.ksplit * n * sizeof(float));
    float* slabs = static_cast<float*>(workspace);
    g9.stats = nullptr;
    int rc = fwd9_launch(x, packed, nullptr, nullptr, slabs, g9, l9, gr9, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(conv_fwd_reduce_kernel, dim3(grid_for(n, 256, 2048)), dim3(256), 0, (hipStream_t)stream, slabs, bias, residual, y, n,
                       Cout, g9.ksplit);
    return check_launch("conv3d_fwd_gn(split-K reduce)");
}

// 'same' convolution (odd cubic filter k, padding k / 2) over the f^3 sub-volume batch x[f^3][A][A][A][Cin] of ONE merged volume, the
// halo of a sub-volume read in place from its neighbours (conv_src_voxel): what the reference computes as
// boundary_pad(x) -> unpadded Conv3d (imagen_pytorch3D.py:37-46, 550-566 with boundary=True).  Same kernels, same bits per output
// element as running the padded copies; workspace / stats as diqt_conv3d_fwd_ex for (B = f^3, D = H = W = A, pad = k / 2).
extern "C" int diqt_conv3d_fwd_neighbours(const float* x, const float* packed, const float* bias, const float* residual, float* y,
                                          float* stats, void* workspace, size_t workspace_bytes, int f, int A, int Cin, int Cout,
                                          int k, void* stream) {
    DIQT_REQUIRE(f >= 1 && A >= 1 && k >= 1 && (k & 1) && k / 2 <= A, DIQT_E_SHAPE, "conv3d_fwd_neighbours: bad shape (f %d, A %d, k %d)", f, A, k);
    const int B = f * f * f, p = k / 2;
    DIQT_REQUIRE((unsigned long long)B * A * A * A * (unsigned long long)(Cin > Cout ? Cin : Cout) * 4ull < (1ull << 30), DIQT_E_UNSUPPORTED,
                 "conv3d_fwd_neighbours: the sub-volume batch must stay below 1 GiB (it cannot be cut into independent launches)");
    DIQT_REQUIRE(!stats || diqt_conv3d_fwd_neighbours_stats_blocks(f, A, Cin, Cout, k) > 0, DIQT_E_UNSUPPORTED,
                 "conv3d_fwd_neighbours: this shape does not produce output statistics (diqt_conv3d_fwd_neighbours_stats_blocks == 0)");
    return conv3d_fwd_one(x, packed, bias, residual, y, workspace, workspace_bytes, B, A, A, A, Cin, Cout, k, k, k, p, p, p, 0, 0, 0, stream,
                          stats, f);
}

extern "C" int diqt_conv3d_fwd_ws(const float* x, const float* packed, const float* bias, const float* residual,
                                  float* y, void* workspace, size_t workspace_bytes, int B, int D, int H, int W, int Cin,
                                  int Cout, int kd, int kh, int kw, int pd, int ph, int pw, int epd, int eph, int epw,
                                  void* stream) {
    return conv3d_fwd_impl(x, packed, bias, residual, y, workspace, workspace_bytes, B, D, H, W, Cin, Cout, kd, kh, kw, pd, ph,
                           pw, epd, eph, epw, stream, nullptr);
}


// The fast kernels address x and y through 32-bit buffer descriptors (tensors < 1 GiB).  Larger launches -- big patch batches are the
// natural way to use 288 GB of HBM -- are cut into independent sub-launches below that size: row ranges for the flattened 1x1x1 /
// Linear geometry, batch-entry ranges otherwise (a single batch entry >= 1 GiB still takes the pointer-arithmetic kernel).
static int conv3d_fwd_impl(const float* x, const float* packed, const float* bias, const float* residual,
                           float* y, void* workspace, size_t workspace_bytes, int B, int D, int H, int W, int Cin, int Cout,
                           int kd, int kh, int kw, int pd, int ph, int pw, int epd, int eph, int epw, void* stream, float* stats) {
    const long long Do = (long long)D + 2 * pd + epd - kd + 1, Ho = (long long)H + 2 * ph + eph - kh + 1, Wo = (long long)W + 2 * pw + epw - kw + 1;
    const unsigned long long lim = (1ull << 30) - 1;
    if (B > 0 && D > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && Do > 0 && Ho > 0 && Wo > 0) {
        const unsigned long long xin = (unsigned long long)D * H * W * Cin * 4ull, yout = (unsigned long long)Do * Ho * Wo * Cout * 4ull;
        const bool flat = kd == 1 && kh == 1 && kw == 1 && !pd && !ph && !pw && !epd && !eph && !epw;
        if ((xin * B > lim || yout * B > lim) && !stats) {
            if (flat) {
                const long long rows = (long long)B * D * H * W;
                long long per = (long long)(lim / (4ull * (unsigned long long)(Cin > Cout ? Cin : Cout))) / MTILE * MTILE;
                if (per >= MTILE && rows > per && rows < (1ll << 31)) {
                    for (long long r = 0; r < rows; r += per) {
                        const int nr = (int)(rows - r < per ? rows - r : per);
                        const int rc = conv3d_fwd_one(x + r * Cin, packed, bias, residual ? residual + r * Cout : nullptr, y + r * Cout,
                                                      nullptr, 0, 1, 1, 1, nr, Cin, Cout, 1, 1, 1, 0, 0, 0, 0, 0, 0, stream, nullptr);
                        if (rc) return rc;
                    }
                    return DIQT_OK;
                }
            } else if (B > 1 && xin <= lim && yout <= lim) {
                unsigned long long per = lim / (xin > yout ? xin : yout);
                if (per < 1) per = 1;
                for (int b0 = 0; b0 < B; b0 += (int)per) {
                    const int nb = B - b0 < (int)per ? B - b0 : (int)per;
                    const int rc = conv3d_fwd_one(x + (size_t)b0 * (xin / 4), packed, bias, residual ? residual + (size_t)b0 * (yout / 4) : nullptr,
                                                  y + (size_t)b0 * (yout / 4), nullptr, 0, nb, D, H, W, Cin, Cout, kd, kh, kw, pd,
                                                  ph, pw, epd, eph, epw, stream, nullptr);
                    if (rc) return rc;
                }
                return DIQT_OK;
            }
        }
    }
    return conv3d_fwd_one(x, packed, bias, residual, y, workspace, workspace_bytes, B, D, H, W, Cin, Cout, kd, kh, kw, pd, ph, pw, epd, eph,
                          epw, stream, stats);
}

static int conv3d_fwd_one(const float* x, const float* packed, const float* bias, const float* residual,
                          float* y, void* workspace, size_t workspace_bytes, int B, int D, int H, int W, int Cin, int Cout,
                          int kd, int kh, int kw, int pd, int ph, int pw, int epd, int eph, int epw, void* stream, float* stats,
                          int subF) {
    DIQT_REQUIRE(x && packed && y, DIQT_E_ALIGN, "conv3d_fwd: null pointer");
    ConvGeom g;
    int rc = make_geom(g, B, D, H, W, Cin, Cout, kd, kh, kw, pd, ph, pw, epd, eph, epw);
    if (rc) return rc;
    g.subF = subF;
    g.chunksPerSplit = g.nChunks;
    g.slabStride = 0;
    DIQT_REQUIRE(aligned16(packed), DIQT_E_ALIGN, "conv3d_fwd: packed weights must be 16-byte aligned");
    const bool vec4 = (Cin % 4 == 0) && aligned16(x);
    const int HV = g.HD * g.HH * g.HWd;
    const unsigned nwg = (unsigned)((long long)g.B * g.tilesD * g.tilesH * g.tilesW * g.nNt);
    if (const int CINP = smallcin_pad(Cin, kd * kh * kw)) {
        const int nCh = cdiv(kd * kh * kw * CINP, CK);
        const size_t slds = ((size_t)((HV * CINP + 3) & ~3) + (size_t)MTILE * LDSROW + (size_t)NT * LDSROW) * sizeof(float) +
                            (MTILE + CK) * sizeof(int);
        if (slds <= 80 * 1024) {
            void (*ks)(const float*, const float*, const float*, const float*, float*, ConvGeom, int) =
                CINP == 1 ? conv_fwd_smallcin_kernel<1> : (CINP == 2 ? conv_fwd_smallcin_kernel<2> : conv_fwd_smallcin_kernel<4>);
            if (slds > 64 * 1024) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(ks), hipFuncAttributeMaxDynamicSharedMemorySize, (int)slds);
                DIQT_REQUIRE(e == hipSuccess, DIQT_E_LAUNCH, "conv3d_fwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
            }
            hipLaunchKernelGGL(ks, dim3(nwg), dim3(256), slds, (hipStream_t)stream, x, packed, bias, residual, y, g, nCh);
            return check_launch("conv3d_fwd(small Cin)");
        }
        DIQT_REQUIRE(false, DIQT_E_UNSUPPORTED, "conv3d_fwd: small-Cin halo needs %zu B of LDS", slds);
    }
    static const size_t ldspad = [] { const char* e = getenv("DIQT_CONV_LDSPAD"); return e ? (size_t)atoi(e) * 1024 : (size_t)0; }();   // occupancy experiment
    size_t lds = ((size_t)HV * (LDSROW + 1) + 2 * NT * LDSROW) * sizeof(float) + MTILE * sizeof(int) + ldspad;
    DIQT_REQUIRE(lds <= 160 * 1024, DIQT_E_UNSUPPORTED, "conv3d_fwd: halo tile needs %zu B of LDS", lds);
    const unsigned long long xb = (unsigned long long)g.B * g.D * g.H * g.W * g.Cin * 4ull;
    const unsigned long long yb = (unsigned long long)g.B * g.Do * g.Ho * g.Wo * g.Cout * 4ull;
    static const bool nobuf = [] { const char* e = getenv("DIQT_CONV_NOBUF"); return e && e[0] == '1'; }();
    const bool buf = vec4 && !nobuf && xb < (1ull << 30) && yb < (1ull << 30);
    if (buf) { g.xBytes = (unsigned)xb; g.yBytes = (unsigned)yb; }
    // opt-in experiment (DIQT_CONV_CK16=1): 16-channel chunks = 44 KB of LDS for a 3x3x3 filter = three workgroups per CU instead of
    // two.  Measured 2 % SLOWER on MI355X (64->64 @ 8x32^3: 467 vs 458 us; 128->128 @ 8x16^3: 243 vs 227 us): a third resident
    // workgroup does not fill the staging bubbles, the doubled barrier count costs more (profiles/r01_conv_ablation.md)
    static const int ck16_env = [] { const char* e = getenv("DIQT_CONV_CK16"); return e ? atoi(e) : 0; }();
    const size_t lds16 = ((size_t)HV * (CK / 2 + 4 + 1) + 2 * NT * (CK / 2 + 4)) * sizeof(float) + MTILE * sizeof(int);
    const bool ck16_ok = buf && lds16 * 3 <= 160 * 1024 && kd * kh * kw > 1;
    const bool ck16 = ck16_ok && ck16_env == 1;
    if (ck16) lds = lds16 + ldspad;
    auto kern = vec4 ? (buf ? (ck16 ? conv_fwd_kernel<true, true, 16> : conv_fwd_kernel<true, true, 32>) : conv_fwd_kernel<true, false, 32>)
                     : conv_fwd_kernel<false, false, 32>;
    static unsigned long long* dbg_buf = nullptr;
    static const bool dbg_on = [] { const char* e = getenv("DIQT_CONV_DBG"); return e && e[0] == '1'; }();
    if (dbg_on) {     // diagnostic build path only: cycle stamps per workgroup, read back with diqt_debug_conv_stamps()
        if (!dbg_buf) (void)hipMalloc(&dbg_buf, (size_t)65536 * 8 * sizeof(unsigned long long));
        if (nwg <= 65536) g.dbg = dbg_buf;
        g_dbg_ptr = dbg_buf; g_dbg_n = nwg <= 65536 ? nwg : 0;
    }
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        DIQT_REQUIRE(e == hipSuccess, DIQT_E_LAUNCH, "conv3d_fwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
    }
    if (buf && !ck16 && g.subF == 0) {
        // conv_fwd9_kernel first: whole rounds of 512- / 256-voxel tiles, or (small volumes) split-K slabs in the caller's workspace
        F9Geom g9;
        size_t l9;
        unsigned gr9;
        const bool maySplit = workspace && aligned16(workspace);
        if (fwd9_plan(g9, l9, gr9, B, D, H, W, Cin, Cout, kd, kh, kw, pd, ph, pw, epd, eph, epw, diqt_conv_packed_elems(Cout, Cin, kd, kh, kw),
                      maySplit)) {
            if (g9.ksplit == 1) {
                g9.stats = stats;
                return fwd9_launch(x, packed, bias, residual, y, g9, l9, gr9, stream);
            }
            const size_t n = (size_t)g.B * g.Do * g.Ho * g.Wo * g.Cout;
            if (!stats && workspace_bytes >= (size_t)g9.ksplit * n * sizeof(float)) {
                float* slabs = static_cast<float*>(workspace);
                g9.stats = nullptr;
                rc = fwd9_launch(x, packed, nullptr, nullptr, slabs, g9, l9, gr9, stream);
                if (rc) return rc;
                hipLaunchKernelGGL(conv_fwd_reduce_kernel, dim3(grid_for(n, 256, 2048)), dim3(256), 0, (hipStream_t)stream, slabs, bias,
                                   residual, y, n, g.Cout, g9.ksplit);
                return check_launch("conv3d_fwd(v9 split-K reduce)");
            }
        }
    }
    const int ks = workspace ? fwd_ksplit(g) : 1;
    static const int tpw_env = [] { const char* e = getenv("DIQT_CONV_TPW"); return e ? atoi(e) : 0; }();
    int tpw = tpw_env > 0 ? tpw_env : 1;      // measured: 2 or 4 tiles per workgroup change nothing (457 / 464 / 459 us), see profiles/r01_conv_ablation.md
    if (dbg_on) tpw = 1;
    g.tilesPerWg = tpw;
    const unsigned gridx = (nwg + tpw - 1) / tpw;
    if (ks > 1) {
        const size_t n = (size_t)g.B * g.Do * g.Ho * g.Wo * g.Cout;
        DIQT_REQUIRE(workspace_bytes >= (size_t)ks * n * sizeof(float) && aligned16(workspace), DIQT_E_WORKSPACE,
                     "conv3d_fwd: split-K workspace %zu < %zu", workspace_bytes, (size_t)ks * n * sizeof(float));
        g.chunksPerSplit = cdiv(g.nChunks, ks);
        g.slabStride = n;
        float* slabs = static_cast<float*>(workspace);
        hipLaunchKernelGGL(kern, dim3(gridx, ks), dim3(256), lds, (hipStream_t)stream, x, packed, nullptr, nullptr, slabs, g);
        rc = check_launch("conv3d_fwd(split-K)");
        if (rc) return rc;
        hipLaunchKernelGGL(conv_fwd_reduce_kernel, dim3(grid_for(n, 256, 2048)), dim3(256), 0, (hipStream_t)stream, slabs, bias,
                           residual, y, n, g.Cout, ks);
        return check_launch("conv3d_fwd(split-K reduce)");
    }
    DIQT_REQUIRE(!stats || buf, DIQT_E_UNSUPPORTED, "conv3d_fwd: output statistics need the buffer-path kernel");
    g.stats = stats;
    if (buf && !ck16) {
        ConvGeom g8;
        size_t lds8;
        if (fwd8_plan(g, g8, lds8)) {
            g8.stats = stats;
            g8.dbg = nullptr;
            if (lds8 > 64 * 1024) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_fwd8_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds8);
                DIQT_REQUIRE(e == hipSuccess, DIQT_E_LAUNCH, "conv3d_fwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
            }
            const unsigned nwg8 = (unsigned)((long long)g8.B * g8.tilesD * g8.tilesH * g8.tilesW * g8.nNt);
            // persistent tile walk: one workgroup per CU runs all its tiles (needs a 64-channel block that stays with the workgroup)
            static const bool nopersist = [] { const char* e = getenv("DIQT_CONV_W8_NOPERSIST"); return e && e[0] == '1'; }();
            const unsigned grid8 = (!nopersist && nwg8 > 256u && 256 % g8.nNt == 0) ? 256u : nwg8;
            hipLaunchKernelGGL(conv_fwd8_kernel, dim3(grid8), dim3(512), lds8, (hipStream_t)stream, x, packed, bias, residual, y, g8);
            return check_launch("conv3d_fwd(8 waves)");
        }
    }
    static const bool no1x1 = [] { const char* e = getenv("DIQT_CONV_NO1X1"); return e && e[0] == '1'; }();
    if (buf && !stats && !dbg_on && !no1x1 && kd * kh * kw == 1 && g.B == 1 && g.D == 1 && g.H == 1 && g.Wo == g.W && g.TW == MTILE) {     // the flattened-rows geometry of make_geom
        if (pw64_ok((long long)g.W, Cin, Cout, x, packed, y))      // few input channels, many output channels: x resident, persistent row walk
            return pw64_launch(x, packed, bias, residual, y, (long long)g.W, Cout, g.CoutPad, stream);
        hipLaunchKernelGGL(conv1x1_fwd_kernel, dim3(nwg), dim3(256), 0, (hipStream_t)stream, x, packed, bias, residual, y, g);
        return check_launch("conv3d_fwd(1x1x1)");
    }
    hipLaunchKernelGGL(kern, dim3(gridx), dim3(256), lds, (hipStream_t)stream, x, packed, bias, residual, y, g);
    return check_launch("conv3d_fwd");
}

static int bw_plan(BwGeom& bg, int B, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw, int pd,
                   int ph, int pw, int epd, int eph, int epw, int& ksplit) {
    int rc = make_geom(bg.g, B, D, H, W, Cin, Cout, kd, kh, kw, pd, ph, pw, epd, eph, epw);
    if (rc) return rc;
    const ConvGeom& g = bg.g;
    bg.tapGroups = cdiv(kh * kw, 2 * BW_MAXT);
    auto ilog2 = [](int v) { int l = 0; while ((1 << l) < v) ++l; return l; };
    bg.lTW = ilog2(g.TW); bg.lTH = ilog2(g.TH);
    DIQT_REQUIRE((1 << bg.lTW) == g.TW && (1 << bg.lTH) == g.TH, DIQT_E_UNSUPPORTED, "conv3d_bwd_weight: tile not a power of two");
    bg.MT = g.B * g.tilesD * g.tilesH * g.tilesW;
    const int gx = g.nChunks * g.nNt * g.kd * bg.tapGroups;
    // 2 workgroups fit a CU (LDS): aim at ONE full round of 512 resident workgroups, never 1.01 rounds
    ksplit = 512 / gx;
    if (ksplit > bg.MT) ksplit = bg.MT;
    if (ksplit < 1) ksplit = 1;
    bg.tilesPerSplit = cdiv(bg.MT, ksplit);
    ksplit = cdiv(bg.MT, bg.tilesPerSplit);
    return DIQT_OK;
}

constexpr int BW2_MAXT_A = 7, BW2_MAXT_B = 3;   // A: 4 SIMDs x 7 taps per group, 2 k-interleaved waves per SIMD
constexpr int BW2_KPAR_A = 1;                   // slabs written per workgroup in mode A (the k-interleaved wave pairs are combined in LDS)
// version-2 plan; returns false when the shape needs the generic (version-1) kernel
static bool bw2_plan(const ConvGeom& g, BwGeom2& b2, bool& splitCo, int& ksplit, size_t& lds) {
    static const bool off = [] { const char* e = getenv("DIQT_BWDW_V1"); return e && e[0] == '1'; }();
    if (off) return false;
    const int T = g.kd * g.kh * g.kw, HV = g.HD * g.HH * g.HWd;
    splitCo = T <= BW2_MAXT_B;
    if (splitCo ? (HV * 8 > 256 * 8) : (HV * 8 > 512 * 7)) return false;      // register-staged halo pieces: NRX = 8 / 7 per thread
    auto ilog2 = [](int v) { int l = 0; while ((1 << l) < v) ++l; return l; };
    b2.g = g;
    b2.lTW = ilog2(g.TW); b2.lTH = ilog2(g.TH);
    if ((1 << b2.lTW) != g.TW || (1 << b2.lTH) != g.TH) return false;
    // the MFMAs of a step are issued for every tap slot of a wave, used or not: filters with <= 12 taps ((1,3,3): 9) take
    // the 3-taps-per-SIMD instantiation instead of idling 4 of 7 slots
    b2.maxtA = (!splitCo && T <= 12) ? 3 : BW2_MAXT_A;
    const int COB = splitCo ? 128 : 32, TG = splitCo ? BW2_MAXT_B : 4 * b2.maxtA;
    b2.tapGroups = cdiv(T, TG);
    b2.coBlocks = cdiv(g.CoutPad, COB);
    b2.MT = g.B * g.tilesD * g.tilesH * g.tilesW;
    const int gx = g.nChunks * b2.coBlocks * b2.tapGroups;
    static const int wgs = [] { const char* e = getenv("DIQT_BWDW_WGS"); return e ? atoi(e) : 256; }();
    ksplit = wgs / gx;      // ONE workgroup per CU: the kernel needs >256 VGPRs to keep its LDS reads batched ahead of the MFMAs
    if (ksplit > b2.MT) ksplit = b2.MT;
    if (ksplit < 1) ksplit = 1;
    b2.tilesPerSplit = cdiv(b2.MT, ksplit);
    ksplit = cdiv(b2.MT, b2.tilesPerSplit);
    lds = ((size_t)HV * CK + (size_t)MTILE * COB) * sizeof(float) + (2 * (size_t)HV + 2 * MTILE) * sizeof(int);
    if (!splitCo) {                                   // room for the end-of-kernel combine of the wave pairs
        const size_t red = (size_t)4 * BW2_MAXT_A * 16 * 64 * sizeof(float);
        if (red > lds) lds = red;
    }
    return lds <= 160 * 1024;
}

// ---- 1x1x1 filters: dW = dY^T X is a plain GEMM with K = all voxels.  It runs on the batched GEMM (bgemm.hip) with the
//      K axis cut into `ks` slices mapped to the batch index (one output slab per slice) + a fixed-order slab sum. ----
extern "C" int diqt_bgemm(const float* A, const float* Bm, float* C, int batch, int M, int N, int K, int transA, int transB,
                          long long strideA, long long strideB, long long strideC, int lda, int ldb, int ldc, float alpha,
                          float beta, void* stream);
extern "C" int diqt_weighted_colsum(const float* x, const float* w, float* out, void* workspace, size_t workspace_bytes,
                                    int B, int rows, int C, void* stream);
extern "C" size_t diqt_reduce_workspace_bytes(int B, int C);

struct PwPlan { bool ok; long long V; int ks; bool xFirst; int M, N; };
static PwPlan pw_plan(int B, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw, int pd, int ph, int pw, int epd,
                      int eph, int epw) {
    static const bool off = [] { const char* e = getenv("DIQT_BWDW_NOGEMM"); return e && e[0] == '1'; }();
    PwPlan p{};
    p.ok = !off && kd == 1 && kh == 1 && kw == 1 && pd == 0 && ph == 0 && pw == 0 && epd == 0 && eph == 0 && epw == 0;
    if (!p.ok) return p;
    p.V = (long long)B * D * H * W;
    p.xFirst = Cin >= Cout;                      // the larger channel count takes the 128-row side of the tile
    p.M = p.xFirst ? Cin : Cout; p.N = p.xFirst ? Cout : Cin;
    if (p.M <= 64 || p.V < 4096) { p.ok = false; return p; }      // small problems stay on the conv kernel
    {
        // conv_wgrad3_kernel's pointwise variant (LDS-DMA double-buffered 64-row tiles, one wave per SIMD) takes the shapes it can:
        // the batched GEMM runs these K = all-rows products at ~30 TFLOP/s
        W3Geom g3;
        int v3 = 0, ks3 = 0;
        size_t lds3 = 0;
        if (wgrad3_plan(g3, v3, ks3, lds3, B, D, H, W, Cin, Cout, 1, 1, 1, 0, 0, 0, 0, 0, 0)) { p.ok = false; return p; }
    }
    const int tiles = cdiv(p.M, 128) * cdiv(p.N, 64);
    int ks = 1;
    while (ks * 2 * tiles <= 256 && p.V % (ks * 2) == 0 && p.V / (ks * 2) >= 256) ks *= 2;      // one workgroup per CU
    p.ks = ks;
    return p;
}

// dw[co][ci] = sum_s slab[s][...]; slabs are [M][N] = [ci][co] when xFirst, else [co][ci].  A block reduces 64 consecutive
// slab elements: 4 thread groups each sum a contiguous quarter of the slabs (coalesced 256-byte reads, 4 loads in flight),
// then the quarters are combined in a fixed order.
__global__ __launch_bounds__(256) void pw_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ dw, int Cout, int Cin,
                                                        int ks, int xFirst) {
    __shared__ float part[4][64];
    const int total = Cout * Cin;
    const int e = blockIdx.x * 64 + (threadIdx.x & 63), sg = threadIdx.x >> 6;
    const int k0 = sg * ks / 4, k1 = (sg + 1) * ks / 4;
    float s = 0.f;
    if (e < total) {
        const float* p = slabs + e;
        int k = k0;
        for (; k + 4 <= k1; k += 4) {
            const float a0 = p[(size_t)k * total], a1 = p[(size_t)(k + 1) * total];
            const float a2 = p[(size_t)(k + 2) * total], a3 = p[(size_t)(k + 3) * total];
            s += a0; s += a1; s += a2; s += a3;
        }
        for (; k < k1; ++k) s += p[(size_t)k * total];
    }
    part[sg][threadIdx.x & 63] = s;
    __syncthreads();
    if (sg == 0 && e < total) {
        const float v = ((part[0][threadIdx.x] + part[1][threadIdx.x]) + part[2][threadIdx.x]) + part[3][threadIdx.x];
        const int dst = xFirst ? (e % Cout) * Cin + e / Cout : e;       // slab element (ci, co) -> dw[co][ci]
        dw[dst] = v;
    }
}

// ---- <= 4 input channels (init convs, Family-B cross-embed): dW through an explicit im2col + split-K GEMM.  col[v][k], k = tap*CINP+ci
//      (the tap-packed K of conv_fwd_smallcin_kernel), then dW'[k][co] = sum_v col[v][k] dY[v][co]. ----
__global__ __launch_bounds__(256) void im2col_smallcin_kernel(const float* __restrict__ x, float* __restrict__ col, ConvGeom g,
                                                              int CINP, int Kp, size_t total) {
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int k = (int)(i % Kp);
        size_t v = i / Kp;
        const int ow = (int)(v % g.Wo); v /= g.Wo;
        const int oh = (int)(v % g.Ho); v /= g.Ho;
        const int od = (int)(v % g.Do);
        const int b = (int)(v / g.Do);
        const int tap = k / CINP, ci = k % CINP;
        const int kx = tap % g.kw, ky = (tap / g.kw) % g.kh, kz = tap / (g.kw * g.kh);
        const int iz = od + kz - g.pd, iy = oh + ky - g.ph, ix = ow + kx - g.pw;
        float val = 0.f;
        if (ci < g.Cin && iz >= 0 && iz < g.D && iy >= 0 && iy < g.H && ix >= 0 && ix < g.W)
            val = x[((((size_t)b * g.D + iz) * g.H + iy) * g.W + ix) * g.Cin + ci];
        col[i] = val;
    }
}
// dw[co][ci][tap] = sum_s slab[s][k = tap*CINP+ci][co]
__global__ __launch_bounds__(256) void smallcin_dw_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ dw, int Cout,
                                                                 int Cin, int T, int CINP, int Kp, int ks) {
    const int total = Cout * Cin * T;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int tap = i % T, ci = (i / T) % Cin, co = i / (T * Cin);
        const size_t src = (size_t)(tap * CINP + ci) * Cout + co;
        float s = 0.f;
        for (int k = 0; k < ks; ++k) s += slabs[(size_t)k * Kp * Cout + src];
        dw[i] = s;
    }
}
struct ScPlan { bool ok; int CINP, Kp, ks; long long V; };
static ScPlan sc_plan(const ConvGeom& g) {
    ScPlan p{};
    const int T = g.kd * g.kh * g.kw;
    p.CINP = smallcin_pad(g.Cin, T);
    p.V = (long long)g.B * g.Do * g.Ho * g.Wo;
    p.ok = p.CINP > 0 && p.V >= 4096 && p.V < (1ll << 31);
    if (!p.ok) return p;
    p.Kp = T * p.CINP;
    const int tiles = cdiv(p.Kp, p.Kp > 64 ? 128 : 64) * cdiv(g.Cout, 64);
    int ks = 1;
    while (ks * 2 * tiles <= 256 && p.V % (ks * 2) == 0 && p.V / (ks * 2) >= 256) ks *= 2;
    p.ks = ks;
    return p;
}

// ---- weight gradient with 16-bit MFMA operands (bf16 training): conv_wgrad_h.hip + the fixed-order slab sum of conv_wgrad3_kernel ----
extern "C" size_t diqt_conv3d_bwd_weight_h_workspace_bytes(int B, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw, int pd,
                                                           int ph, int pw, int epd, int eph, int epw) {
    WHGeom g;
    int ks = 0;
    if (!wgradh_plan(g, ks, B, D, H, W, Cin, Cout, kd, kh, kw, pd, ph, pw, epd, eph, epw)) return 0;       // 0: shape not taken
    return ((size_t)ks * Cout * Cin * kd * kh * kw + (size_t)ks * g.CoutPad) * sizeof(float);
}
extern "C" int diqt_conv3d_bwd_weight_h(const float* x, const float* dy, float* dw, float* dbias, void* workspace, size_t workspace_bytes,
                                        int B, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw, int pd, int ph, int pw,
                                        int epd, int eph, int epw, int bf16, void* stream) {
    DIQT_REQUIRE(x && dy && dw && workspace, DIQT_E_ALIGN, "conv3d_bwd_weight_h: null pointer");
    DIQT_REQUIRE(aligned16(x) && aligned16(dy) && aligned16(dw) && aligned16(workspace), DIQT_E_ALIGN, "conv3d_bwd_weight_h: pointers must be 16-byte aligned");
    WHGeom g;
    int ks = 0;
    const bool xHalf = (bf16 & 2) != 0, dyHalf = (bf16 & 4) != 0;      // bits 1, 2 of `bf16`: x / dY hold 16-bit values of the operand type
    bf16 &= 1;
    DIQT_REQUIRE(wgradh_plan(g, ks, B, D, H, W, Cin, Cout, kd, kh, kw, pd, ph, pw, epd, eph, epw, xHalf, dyHalf), DIQT_E_UNSUPPORTED,
                 "conv3d_bwd_weight_h: shape not taken (diqt_conv3d_bwd_weight_h_workspace_bytes == 0)");
    const int T = kd * kh * kw;
    const size_t need = ((size_t)ks * Cout * Cin * T + (size_t)ks * g.CoutPad) * sizeof(float);
    DIQT_REQUIRE(workspace_bytes >= need, DIQT_E_WORKSPACE, "conv3d_bwd_weight_h: workspace %zu < %zu", workspace_bytes, need);
    float* slabs = static_cast<float*>(workspace);
    float* bias_part = dbias ? slabs + (size_t)ks * Cout * Cin * T : nullptr;
    int rc = wgradh_launch(x, dy, slabs, bias_part, g, ks, bf16, stream);
    if (rc) return rc;
    const size_t n4 = (size_t)Cout * Cin * T / 4;
    hipLaunchKernelGGL(conv_reduce_dw3_kernel, dim3((unsigned)((n4 + 63) / 64)), dim3(512), 0, (hipStream_t)stream, slabs, dw, n4, ks, Cout,
                       g.CoutPad, bias_part, bias_part ? dbias : nullptr, ks);
    return check_launch("conv_reduce_dw3(h)");
}

extern "C" size_t diqt_conv3d_bwd_weight_workspace_bytes(int B, int D, int H, int W, int Cin, int Cout, int kd,
                                                         int kh, int kw, int pd, int ph, int pw, int epd, int eph,
                                                         int epw) {
    BwGeom bg;
    int ksplit;
    if (bw_plan(bg, B, D, H, W, Cin, Cout, kd, kh, kw, pd, ph, pw, epd, eph, epw, ksplit)) return 0;
    BwGeom2 b2;
    bool splitCo;
    int ks2 = 0;
    size_t lds2;
    if (bw2_plan(bg.g, b2, splitCo, ks2, lds2) && ks2 * (splitCo ? 1 : BW2_KPAR_A) > ksplit) ksplit = ks2 * (splitCo ? 1 : BW2_KPAR_A);
    {
        W3Geom g3;
        int v3 = 0, ks3 = 0;
        size_t lds3 = 0;
        if (wgrad3_plan(g3, v3, ks3, lds3, B, D, H, W, Cin, Cout, kd, kh, kw, pd, ph, pw, epd, eph, epw) && ks3 > ksplit) ksplit = ks3;
    }
    const size_t slab = (size_t)bg.g.nChunks * kd * kh * kw * bg.g.CoutPad * CK * sizeof(float);
    const size_t colsum = (size_t)1024 * Cout * sizeof(float);
    size_t need = (size_t)ksplit * slab + (size_t)ksplit * bg.g.CoutPad * sizeof(float);   // + bias partials (v2)
    const PwPlan pp = pw_plan(B, D, H, W, Cin, Cout, kd, kh, kw, pd, ph, pw, epd, eph, epw);
    if (pp.ok) {
        const size_t gemm = (size_t)pp.ks * Cout * Cin * sizeof(float);
        if (gemm > need) need = gemm;
    }
    if (Cout == 1 && kd * kh * kw == 1) {
        const size_t red = diqt_reduce_workspace_bytes(1, Cin);
        if (red > need) need = red;
    }
    const ScPlan sp = sc_plan(bg.g);
    if (sp.ok) {
        const size_t sc = ((size_t)sp.V * sp.Kp + (size_t)sp.ks * sp.Kp * Cout) * sizeof(float);
        if (sc > need) need = sc;
    }
    return need > colsum ? need : colsum;
}

// which kernel diqt_conv3d_bwd_weight dispatches this shape to (for profilers / bench.py, so that per-kernel numbers carry the names
// rocprofv3 reports): 3 conv_wgrad3_kernel, 2 conv_bwd_weight2_kernel, 1 conv_bwd_weight_kernel, 0 one of the GEMM / column-sum
// paths (1x1x1 filters, <= 4 input channels, Cout == 1), -1 bad shape
extern "C" int diqt_conv3d_bwd_weight_kernel_id(int B, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw, int pd, int ph,
                                                int pw, int epd, int eph, int epw) {
    BwGeom bg;
    int ksplit;
    if (bw_plan(bg, B, D, H, W, Cin, Cout, kd, kh, kw, pd, ph, pw, epd, eph, epw, ksplit)) return -1;
    if (Cout == 1 && kd * kh * kw == 1) return 0;
    if (sc_plan(bg.g).ok || pw_plan(B, D, H, W, Cin, Cout, kd, kh, kw, pd, ph, pw, epd, eph, epw).ok) return 0;
    W3Geom g3;
    int v3 = 0, ks3 = 0;
    size_t lds3 = 0;
    if (wgrad3_plan(g3, v3, ks3, lds3, B, D, H, W, Cin, Cout, kd, kh, kw, pd, ph, pw, epd, eph, epw)) return 3;
    BwGeom2 b2;
    bool splitCo;
    int ks2;
    size_t lds2;
    return bw2_plan(bg.g, b2, splitCo, ks2, lds2) ? 2 : 1;
}

extern "C" int diqt_conv3d_bwd_weight(const float* x, const float* dy, float* dw, float* dbias, void* workspace,
                                      size_t workspace_bytes, int B, int D, int H, int W, int Cin, int Cout, int kd,
                                      int kh, int kw, int pd, int ph, int pw, int epd, int eph, int epw, void* stream) {
    DIQT_REQUIRE(x && dy && dw && workspace, DIQT_E_ALIGN, "conv3d_bwd_weight: null pointer");
    BwGeom bg;
    int ksplit;
    int rc = bw_plan(bg, B, D, H, W, Cin, Cout, kd, kh, kw, pd, ph, pw, epd, eph, epw, ksplit);
    if (rc) return rc;
    const ConvGeom& g = bg.g;
    const int T = kd * kh * kw;
    const size_t need = diqt_conv3d_bwd_weight_workspace_bytes(B, D, H, W, Cin, Cout, kd, kh, kw, pd, ph, pw, epd, eph, epw);
    DIQT_REQUIRE(workspace_bytes >= need, DIQT_E_WORKSPACE, "conv3d_bwd_weight: workspace %zu < %zu", workspace_bytes, need);
    DIQT_REQUIRE(aligned16(workspace), DIQT_E_ALIGN, "conv3d_bwd_weight: workspace must be 16-byte aligned");
    const PwPlan pp = pw_plan(B, D, H, W, Cin, Cout, kd, kh, kw, pd, ph, pw, epd, eph, epw);
    if (Cout == 1 && kd == 1 && kh == 1 && kw == 1 && pd == 0 && ph == 0 && pw == 0 && epd == 0 && eph == 0 && epw == 0 &&
        workspace_bytes >= diqt_reduce_workspace_bytes(1, Cin) && (long long)B * D * H * W < (1ll << 31)) {
        // final 1x1x1 conv to one channel: dW[0][ci] = sum_v dY[v] X[v][ci] is a weighted column sum, dbias = sum_v dY[v]
        const int rows = (int)((long long)B * D * H * W);
        rc = diqt_weighted_colsum(x, dy, dw, workspace, workspace_bytes, 1, rows, Cin, stream);
        if (rc) return rc;
        if (dbias) {
            float* part = static_cast<float*>(workspace);
            unsigned nblk = (unsigned)((rows + 255) / 256);
            if (nblk > 1024) nblk = 1024;
            hipLaunchKernelGGL(colsum_stage1_kernel, dim3(nblk), dim3(256), 256 * sizeof(float), (hipStream_t)stream, dy, part, (size_t)rows, 1);
            rc = check_launch("colsum_stage1");
            if (rc) return rc;
            hipLaunchKernelGGL(colsum_stage2_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, part, dbias, (int)nblk, 1);
            rc = check_launch("colsum_stage2");
        }
        return rc;
    }
    const ScPlan sp = sc_plan(g);
    if (sp.ok) {
        float* col = static_cast<float*>(workspace);
        float* slabs = col + (size_t)sp.V * sp.Kp;
        const size_t total = (size_t)sp.V * sp.Kp;
        hipLaunchKernelGGL(im2col_smallcin_kernel, dim3(grid_for(total, 256, 16384)), dim3(256), 0, (hipStream_t)stream, x, col, g,
                           sp.CINP, sp.Kp, total);
        rc = check_launch("conv3d_bwd_weight(im2col)");
        if (rc) return rc;
        const long long kslice = sp.V / sp.ks;
        rc = diqt_bgemm(col, dy, slabs, sp.ks, sp.Kp, Cout, (int)kslice, 1, 0, kslice * sp.Kp, kslice * Cout, (long long)sp.Kp * Cout,
                        sp.Kp, Cout, Cout, 1.f, 0.f, stream);
        if (rc) return rc;
        hipLaunchKernelGGL(smallcin_dw_reduce_kernel, dim3(grid_for((size_t)Cout * Cin * T, 256, 1024)), dim3(256), 0, (hipStream_t)stream,
                           slabs, dw, Cout, Cin, T, sp.CINP, sp.Kp, sp.ks);
        rc = check_launch("conv3d_bwd_weight(small Cin reduce)");
        if (rc) return rc;
        if (dbias) {
            const size_t rows = (size_t)sp.V;
            unsigned nblk = (unsigned)((rows + 15) / 16);
            if (nblk > 1024) nblk = 1024;
            hipLaunchKernelGGL(colsum_stage1_kernel, dim3(nblk), dim3(256), 256 * sizeof(float), (hipStream_t)stream, dy, col, rows, Cout);
            rc = check_launch("colsum_stage1");
            if (rc) return rc;
            hipLaunchKernelGGL(colsum_stage2_kernel, dim3(cdiv(Cout, 4)), dim3(256), 0, (hipStream_t)stream, col, dbias, (int)nblk, Cout);
            rc = check_launch("colsum_stage2");
        }
        return rc;
    }
    if (pp.ok) {
        float* slabs = static_cast<float*>(workspace);
        const long long kslice = pp.V / pp.ks;
        const float* Am = pp.xFirst ? x : dy;
        const float* Bmm = pp.xFirst ? dy : x;
        const int ldA = pp.xFirst ? Cin : Cout, ldB = pp.xFirst ? Cout : Cin;
        rc = diqt_bgemm(Am, Bmm, slabs, pp.ks, pp.M, pp.N, (int)kslice, 1, 0, kslice * ldA, kslice * ldB, (long long)pp.M * pp.N,
                        ldA, ldB, pp.N, 1.f, 0.f, stream);
        if (rc) return rc;
        hipLaunchKernelGGL(pw_reduce_kernel, dim3((unsigned)cdiv(Cout * Cin, 64)), dim3(256), 0, (hipStream_t)stream, slabs,
                           dw, Cout, Cin, pp.ks, pp.xFirst ? 1 : 0);
        rc = check_launch("conv3d_bwd_weight(1x1x1 reduce)");
        if (rc) return rc;
        if (dbias) {
            const size_t rows = (size_t)pp.V;
            // wide layers at the coarse levels have few rows: slice them finely so the column sums still fill the chip
            unsigned nblk = (unsigned)((rows + 15) / 16);
            if (nblk > 1024) nblk = 1024;
            hipLaunchKernelGGL(colsum_stage1_kernel, dim3(nblk), dim3(256), 256 * sizeof(float), (hipStream_t)stream, dy, slabs, rows, Cout);
            rc = check_launch("colsum_stage1");
            if (rc) return rc;
            hipLaunchKernelGGL(colsum_stage2_kernel, dim3(cdiv(Cout, 4)), dim3(256), 0, (hipStream_t)stream, slabs, dbias, (int)nblk, Cout);
            rc = check_launch("colsum_stage2");
        }
        return rc;
    }
    const bool vec4 = (Cin % 4 == 0) && aligned16(x) && aligned16(dy);
    const size_t lds = ((size_t)g.TD * g.HH * g.HWd * (CK + 1) + (size_t)MTILE * (NT + 1)) * sizeof(float);
    DIQT_REQUIRE(lds <= 160 * 1024, DIQT_E_UNSUPPORTED, "conv3d_bwd_weight: tile needs %zu B of LDS", lds);
    auto kern = vec4 ? conv_bwd_weight_kernel<true> : conv_bwd_weight_kernel<false>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        DIQT_REQUIRE(e == hipSuccess, DIQT_E_LAUNCH, "conv3d_bwd_weight: hipFuncSetAttribute: %s", hipGetErrorString(e));
    }
    hipStream_t s = (hipStream_t)stream;
    float* slabs = static_cast<float*>(workspace);
    BwGeom2 b2;
    bool splitCo = false;
    int ks2 = 0;
    size_t lds2 = 0;
    float* bias_part = nullptr;      // v2 kernel: per-split-K bias-gradient partials behind the slabs
    int bias_parts = 0;
    W3Geom g3;
    int v3 = 0, ks3 = 0;
    size_t lds3 = 0;
    if (aligned16(x) && aligned16(dy) && aligned16(dw) && wgrad3_plan(g3, v3, ks3, lds3, B, D, H, W, Cin, Cout, kd, kh, kw, pd, ph, pw, epd, eph, epw)) {
        // version 3 (conv_wgrad.hip): one wave per SIMD, LDS-DMA double-buffered tiles; same slab layout and reduce
        if (dbias) { bias_part = slabs + (size_t)ks3 * Cout * Cin * T; bias_parts = ks3; }
        rc = wgrad3_launch(x, dy, slabs, bias_part, g3, v3, ks3, lds3, stream);
        if (rc) return rc;
        const size_t n4 = (size_t)Cout * Cin * T / 4;          // Cin % 4 == 0 (wgrad3_plan)
        hipLaunchKernelGGL(conv_reduce_dw3_kernel, dim3((unsigned)((n4 + 63) / 64)), dim3(512), 0, s, slabs, dw, n4, ks3, Cout, g.CoutPad,
                           bias_part, bias_part ? dbias : nullptr, bias_parts);
        return check_launch("conv_reduce_dw3");
    } else if (bw2_plan(g, b2, splitCo, ks2, lds2)) {
        void (*k2)(const float*, const float*, float*, float*, BwGeom2) =
            splitCo ? (vec4 ? conv_bwd_weight2_kernel<true, 8, 16, BW2_MAXT_B, true, 256> : conv_bwd_weight2_kernel<false, 8, 16, BW2_MAXT_B, true, 256>)
                    : (b2.maxtA == 3
                           ? (vec4 ? conv_bwd_weight2_kernel<true, 7, 2, 3, false, 512> : conv_bwd_weight2_kernel<false, 7, 2, 3, false, 512>)
                           : (vec4 ? conv_bwd_weight2_kernel<true, 7, 2, BW2_MAXT_A, false, 512> : conv_bwd_weight2_kernel<false, 7, 2, BW2_MAXT_A, false, 512>));
        if (lds2 > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k2), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
            DIQT_REQUIRE(e == hipSuccess, DIQT_E_LAUNCH, "conv3d_bwd_weight: hipFuncSetAttribute: %s", hipGetErrorString(e));
        }
        ksplit = ks2 * (splitCo ? 1 : BW2_KPAR_A);
        static unsigned long long* bdbg = nullptr;
        static const bool bdbg_on = [] { const char* e = getenv("DIQT_CONV_DBG"); return e && e[0] == '1'; }();
        if (bdbg_on) {
            if (!bdbg) (void)hipMalloc(&bdbg, (size_t)65536 * 8 * sizeof(unsigned long long));
            b2.g.dbg = bdbg; g_dbg_ptr = bdbg; g_dbg_n = g.nChunks * b2.coBlocks * b2.tapGroups * ks2 * (splitCo ? 4 : 8);
        }
        if (dbias) { bias_part = slabs + (size_t)ksplit * g.nChunks * T * g.CoutPad * CK; bias_parts = ks2; }
        hipLaunchKernelGGL(k2, dim3(g.nChunks * b2.coBlocks * b2.tapGroups, ks2), dim3(splitCo ? 256 : 512), lds2, s, x, dy, slabs,
                           bias_part, b2);
        rc = check_launch("conv3d_bwd_weight(v2)");
    } else {
        const dim3 grid(g.nChunks * g.nNt * g.kd * bg.tapGroups, ksplit);
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, x, dy, slabs, bg);
        rc = check_launch("conv3d_bwd_weight");
    }
    if (rc) return rc;
    const size_t total = (size_t)Cout * Cin * T;
    hipLaunchKernelGGL(conv_reduce_dw_kernel, dim3(grid_for(total, 256)), dim3(256), 0, s, slabs, dw, Cout, Cin, T,
                       g.CoutPad, g.nChunks, ksplit, bias_part, bias_part ? dbias : nullptr, bias_parts);
    rc = check_launch("conv_reduce_dw");
    if (rc) return rc;
    if (dbias && !bias_part) {
        const size_t rows = (size_t)g.B * g.Do * g.Ho * g.Wo;
        unsigned nblk = (unsigned)((rows + 15) / 16);
        if (nblk > 1024) nblk = 1024;
        if (nblk < 1) nblk = 1;
        // the slabs were consumed by the reduce kernel above (same stream) -> reuse the workspace
        hipLaunchKernelGGL(colsum_stage1_kernel, dim3(nblk), dim3(256), 256 * sizeof(float), s, dy, slabs, rows, Cout);
        rc = check_launch("colsum_stage1");
        if (rc) return rc;
        hipLaunchKernelGGL(colsum_stage2_kernel, dim3(cdiv(Cout, 4)), dim3(256), 0, s, slabs, dbias, (int)nblk, Cout);
        rc = check_launch("colsum_stage2");
    }
    return rc;
}
