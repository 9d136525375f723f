// Mixed-precision forward convolution for gfx950 on v_mfma_f32_32x32x16_{f16,bf16}: fp16 / bf16 operands, fp32 accumulate.
//
// This is the autocast path of the sampler (`torch.autocast` around `sample`; the reference gets it from
// ATen's autocast policy: conv3d / linear inputs are cast to the low-precision type, the result is rounded to it once).
// Activations stay fp32 NDHWC in HBM; the cast happens while the halo tile is staged, so no other kernel changes.
//
// GEMM view as in conv_mfma.hip (M = voxels, N = Cout, K = taps x Cin), re-tiled for a 16x faster MFMA:
//   * 512-thread workgroup (8 waves, two per SIMD), 256 output voxels x 64 output channels; wave w owns voxels 32w..32w+31
//     and both 32-channel halves (two 32x32 accumulators).
//   * halo tile (TD+kd-1)(TH+kh-1)(TW+kw-1) x 32 channels, converted to 16-bit, 80-byte LDS rows (64 B payload + 16 B pad:
//     row r starts at 16-byte slot 5r mod 16, so the 16 lanes of a ds_read_b128 pass hit distinct slots).
//   * weights of up to NINE taps (one kd-plane of a 3x3x3 filter) are staged per step, double-buffered: the next group's panels
//     are loaded global -> registers before the tap loop and written to the other buffer after it, so a step has ONE barrier per
//     36 MFMAs of a wave instead of one per 4.  The next chunk's halo is prefetched into registers during the last group.
//   * per tap and wave: 6 ds_read_b128 (A k-steps 0/1, B for both channel halves) feed 4 MFMAs.
// Packed weights: half/bf16 [chunk = ci/32][tap][co padded to 64][32 ci]  (conv_pack_weight_h_kernel).
#include "common.h"
#include "conv_f9h.h"
#include <stdlib.h>
#include <atomic>
#include <type_traits>

namespace diqt {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int HCK = 32;          // channels per K-chunk
constexpr int HROWB = 80;        // LDS row bytes (32 x 2 B + 16 B pad)
constexpr int HNT = 64;          // output channels per workgroup
constexpr int HMT = 256;         // output voxels per workgroup
constexpr int HTG = 9;           // taps per staged weight group
constexpr int HWREG = (HTG * 256 + 511) / 512;      // 16-byte weight pieces per thread and group
constexpr int HHREG = 10;        // 16-byte fp32 halo pieces per thread that can be prefetched (HV <= 640)
constexpr unsigned HBUF_OOB = 0x80000000u, HBUF_OOB_C = 0x40000000u;

struct HalfGeom {
    int B, D, H, W, Cin, Cout;
    int Do, Ho, Wo;
    int kd, kh, kw, pd, ph, pw;
    int TD, TH, TW;
    int tilesD, tilesH, tilesW;
    int nNt, nChunks, CoutPad;
    int HD, HH, HWd;
    int TG, nGroups;             // taps per weight group, groups per chunk
    int roundOut;                // 1: round the result to the operand type before the fp32 store (autocast semantics)
    int NS, realChunks;          // persistent kernel, pointwise filters: NS > 1 consecutive 32-channel chunks are staged per step and walked
                                 // like taps (nChunks then counts steps of NS chunks, realChunks the 32-channel chunks)
    unsigned xBytes, yBytes;
    unsigned long long* dbg;     // diagnostic cycle stamps per workgroup (DIQT_CONVH_DBG=1), NULL in production
};

__host__ __device__ inline int hcdiv(int a, int b) { return (a + b - 1) / b; }

// NOTE: __builtin_bit_cast applied directly to a vector COMPONENT lvalue (v.y) reinterprets the first bytes of the whole vector
// (hipcc 7.2: every component reads element 0); passing the component by value first is correct.
__device__ __forceinline__ float asf(unsigned u) { return __builtin_bit_cast(float, u); }

template <bool BF>
__device__ __forceinline__ unsigned pack2(float a, float b) {
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
__device__ __forceinline__ float round_through(float a) {
    return BF ? (float)(__bf16)a : (float)(_Float16)a;
}

template <bool BF>
__device__ __forceinline__ f32x16 mfma16(u32x4 a, u32x4 b, f32x16 c) {
    if (BF) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// packed[((chunk*T + tap)*CoutPad + o)*32 + k]:  mode 0: (half) w[o][chunk*32 + k][tap]  (effective out, in = Cout, Cin);
// mode 1 (backward-data: dX = conv(dY, flipped W)): (half) w[chunk*32 + k][o][T-1-tap]  (effective out, in = Cin, Cout)
template <bool BF>
__global__ void conv_pack_weight_h_kernel(const float* __restrict__ w, unsigned short* __restrict__ packed, int Cout, int Cin, int T,
                                          int mode, int CoutPad, size_t total) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int k = (int)(i % HCK);
        size_t r = i / HCK;
        const int o = (int)(r % CoutPad);
        r /= CoutPad;
        const int tap = (int)(r % T);
        const int in = (int)(r / T) * HCK + k;
        float v = 0.f;
        if (mode == 0) {
            if (o < Cout && in < Cin) v = w[((size_t)o * Cin + in) * T + tap];
        } else {
            if (o < Cin && in < Cout) v = w[((size_t)in * Cin + o) * T + (T - 1 - tap)];
        }
        packed[i] = (unsigned short)(pack2<BF>(v, 0.f) & 0xffffu);
    }
}

// the same for a whole set of weights in ONE launch (blockIdx.y = weight; the table travels in the kernel arguments): a captured training
// micro-step re-derives every packed copy on each replay -- 86 launches of 4.8 us for the C2 U-Net as single packs
constexpr int PK_ROWS = 64;
struct PackRow { const float* w; unsigned short* packed; int Cout, Cin, T, mode, CoutPad; unsigned total; };
struct PackTable { PackRow r[PK_ROWS]; };
static_assert(sizeof(PackTable) + 16 <= 4096, "kernel arguments are limited to 4 KiB");
template <bool BF>
__global__ __launch_bounds__(256) void conv_pack_weight_h_multi_kernel(const PackTable t) {
    const PackRow& e = t.r[blockIdx.y];
    const float* __restrict__ w = e.w;
    unsigned short* __restrict__ packed = e.packed;
    const int Cout = e.Cout, Cin = e.Cin, T = e.T, mode = e.mode, CoutPad = e.CoutPad;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < e.total; i += gridDim.x * 256u) {
        const int k = (int)(i % HCK);
        unsigned r = i / HCK;
        const int o = (int)(r % CoutPad);
        r /= CoutPad;
        const int tap = (int)(r % T);
        const int in = (int)(r / T) * HCK + k;
        float v = 0.f;
        if (mode == 0) {
            if (o < Cout && in < Cin) v = w[((size_t)o * Cin + in) * T + tap];
        } else {
            if (o < Cin && in < Cout) v = w[((size_t)in * Cin + o) * T + (T - 1 - tap)];
        }
        packed[i] = (unsigned short)(pack2<BF>(v, 0.f) & 0xffffu);
    }
}

template <bool BF, bool PREF>
__global__ __launch_bounds__(512, 1) void conv_fwd_h_kernel(const float* __restrict__ x, const unsigned short* __restrict__ wp,
                                                             const float* __restrict__ bias, const float* __restrict__ residual,
                                                             float* __restrict__ y, HalfGeom g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char hsm[];
    const int HV = g.HD * g.HH * g.HWd;
    unsigned char* halo = hsm;                                         // [HV][80 B]
    unsigned char* wbuf = hsm + (size_t)HV * HROWB;                    // [2][TG][64][80 B]
    const int wbufBytes = g.TG * HNT * HROWB;
    int* out_off = reinterpret_cast<int*>(wbuf + 2 * (size_t)wbufBytes);   // [256]
    int* halo_src = out_off + HMT;                                          // [HV]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int T = g.kd * g.kh * g.kw;

    const unsigned L = xcd_remap(blockIdx.x, gridDim.x);
    const int nt = L % g.nNt;
    int mt = L / g.nNt;
    const int tx = mt % g.tilesW; mt /= g.tilesW;
    const int ty = mt % g.tilesH; mt /= g.tilesH;
    const int tz = mt % g.tilesD;
    const int b = mt / g.tilesD;
    const int d0 = tz * g.TD, h0 = ty * g.TH, w0 = tx * g.TW;
    const int n0 = nt * HNT;

    if (tid < HMT) {
        const int tw = tid % g.TW, th = (tid / g.TW) % g.TH, td = tid / (g.TW * g.TH);
        const int od = d0 + td, oh = h0 + th, ow = w0 + tw;
        int off = (int)HBUF_OOB;
        if (od < g.Do && oh < g.Ho && ow < g.Wo) off = (((b * g.Do + od) * g.Ho + oh) * g.Wo + ow) * g.Cout * 4;
        out_off[tid] = off;
    }
    for (int hv = tid; hv < HV; hv += 512) {
        const int hx = hv % g.HWd, hy = (hv / g.HWd) % g.HH, hz = hv / (g.HWd * g.HH);
        const int iz = d0 + hz - g.pd, iy = h0 + hy - g.ph, ix = w0 + hx - g.pw;
        int src = (int)HBUF_OOB;
        if (iz >= 0 && iz < g.D && iy >= 0 && iy < g.H && ix >= 0 && ix < g.W) src = (((b * g.D + iz) * g.H + iy) * g.W + ix) * g.Cin * 4;
        halo_src[hv] = src;
    }
    const auto rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, (int)g.xBytes, 0x00020000);

    int a_base;      // byte offset of this lane's voxel row (A operand) in the halo image at tap (0,0,0), k-half h
    {
        const int v = wave * 32 + l31;
        const int tw = v % g.TW, th = (v / g.TW) % g.TH, td = v / (g.TW * g.TH);
        a_base = ((td * g.HH + th) * g.HWd + tw) * HROWB + h * 16;
    }
    const int b_base = l31 * HROWB + h * 16;

    f32x16 acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }

    const int hq = tid & 7;                         // this thread's channel quad inside a 32-channel chunk
    const int nHalo = HV * 8;                       // 16-byte fp32 pieces of a halo chunk
    __syncthreads();                                // tables visible

    // ---- halo chunk: global fp32 -> registers (batches of 8 loads in flight) -> 16-bit LDS rows ----
    auto stage_halo_sync = [&](int ci0) {
        const unsigned coff = (ci0 + hq * 4 < g.Cin) ? (unsigned)(ci0 + hq * 4) * 4u : HBUF_OOB_C;
        for (int base = 0; base < nHalo; base += 512 * 8) {
            u32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = base + u * 512 + tid;
                const unsigned t = (unsigned)halo_src[min(idx >> 3, HV - 1)] + coff;
                v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, idx < nHalo ? t : HBUF_OOB, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = base + u * 512 + tid;
                if (idx < nHalo) {
                    u32x2 p;
                    p.x = pack2<BF>(asf(v[u].x), asf(v[u].y));
                    p.y = pack2<BF>(asf(v[u].z), asf(v[u].w));
                    *reinterpret_cast<u32x2*>(halo + (idx >> 3) * HROWB + hq * 8) = p;
                }
            }
        }
    };
    // ---- weight group (chunk, taps t0..t0+n-1): global -> registers; registers -> LDS buffer ----
    u32x4 wr[HWREG];
    auto load_wgroup = [&](int chunk, int t0, int n) {
        const unsigned short* src = wp + ((size_t)chunk * T * g.CoutPad + n0) * HCK;
#pragma unroll
        for (int u = 0; u < HWREG; ++u) {
            const int idx = u * 512 + tid;
            const int tap = min(idx >> 8, n - 1), row = (idx >> 2) & 63, q = idx & 3;
            // UNCONDITIONAL (clamped tap): a guarded load becomes an exec-masked branch with an s_waitcnt vmcnt(0) in front of
            // every load, i.e. one serialized L2 round trip per piece; the store below is the guarded side
            wr[u] = *reinterpret_cast<const u32x4*>(src + ((size_t)(t0 + tap) * g.CoutPad + row) * HCK + q * 8);
        }
    };
    auto store_wgroup = [&](int buf, int n) {
        unsigned char* dst = wbuf + buf * wbufBytes;
#pragma unroll
        for (int u = 0; u < HWREG; ++u) {
            const int idx = u * 512 + tid;
            const int tap = idx >> 8, row = (idx >> 2) & 63, q = idx & 3;
            if (tap < n) *reinterpret_cast<u32x4*>(dst + (tap * HNT + row) * HROWB + q * 16) = wr[u];
        }
    };

    const long long ts0 = g.dbg ? (long long)__builtin_readcyclecounter() : 0;
    long long tTap = 0, tSync = 0, tq = 0;
    stage_halo_sync(0);
    load_wgroup(0, 0, min(g.TG, T));
    store_wgroup(0, min(g.TG, T));
    __syncthreads();

    const long long ts1 = g.dbg ? (long long)__builtin_readcyclecounter() : 0;
    u32x4 hr[PREF ? HHREG : 1];
    int step = 0;
    for (int chunk = 0; chunk < g.nChunks; ++chunk) {
        for (int grp = 0; grp < g.nGroups; ++grp, ++step) {
            const int t0 = grp * g.TG, nTap = min(g.TG, T - t0);
            const bool lastGrp = grp + 1 == g.nGroups, more = !(lastGrp && chunk + 1 == g.nChunks);
            const int nchunk = lastGrp ? chunk + 1 : chunk, nt0 = lastGrp ? 0 : t0 + g.TG, nn = min(g.TG, T - nt0);
            if (more) load_wgroup(nchunk, nt0, nn);
            const bool prefHalo = PREF && lastGrp && more;
            if constexpr (PREF) if (prefHalo) {
                const int ci0 = (chunk + 1) * HCK;
                const unsigned coff = (ci0 + hq * 4 < g.Cin) ? (unsigned)(ci0 + hq * 4) * 4u : HBUF_OOB_C;
#pragma unroll
                for (int u = 0; u < HHREG; ++u) {
                    const int idx = u * 512 + tid;
                    const unsigned t = (unsigned)halo_src[min(idx >> 3, HV - 1)] + coff;
                    hr[u] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, idx < nHalo ? t : HBUF_OOB, 0, 0);
                }
            }
            if (g.dbg) tq = (long long)__builtin_readcyclecounter();
            // ---- taps of this group: fragments of tap t+1 are read while the MFMAs of tap t issue ----
            const unsigned char* wcur = wbuf + (step & 1) * wbufBytes + b_base;
            const unsigned char* ap = halo + a_base;
            // tap (kz,ky,kx) -> byte offset in the halo image, advanced with wave-uniform counters (an LDS table read per tap
            // would put a dependent LDS round trip in front of every tap's fragment reads)
            int kx = t0 % g.kw, ky = (t0 / g.kw) % g.kh, kz = t0 / (g.kw * g.kh);
            int toff = ((kz * g.HH + ky) * g.HWd + kx) * HROWB;
            auto next_tap = [&]() {
                toff += HROWB;
                if (++kx == g.kw) {
                    kx = 0; toff += (g.HWd - g.kw) * HROWB;
                    if (++ky == g.kh) { ky = 0; toff += (g.HH - g.kh) * g.HWd * HROWB; }
                }
            };
            u32x4 a0 = *reinterpret_cast<const u32x4*>(ap + toff), a1 = *reinterpret_cast<const u32x4*>(ap + toff + 32);
            u32x4 b00 = *reinterpret_cast<const u32x4*>(wcur), b01 = *reinterpret_cast<const u32x4*>(wcur + 32);
            u32x4 b10 = *reinterpret_cast<const u32x4*>(wcur + 32 * HROWB), b11 = *reinterpret_cast<const u32x4*>(wcur + 32 * HROWB + 32);
            // the first tap's fragments land HERE: without this explicit wait hipcc's counter tracking merges "fragments pending"
            // from the loop entry into the loop header and puts an s_waitcnt lgkmcnt(0) in front of every tap's MFMAs, i.e.
            // behind the next tap's reads, which undoes the software pipelining
            __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0)
            for (int t = 0; t < nTap; ++t) {
                u32x4 na0, na1, nb00, nb01, nb10, nb11;
                if (t + 1 < nTap) {
                    next_tap();
                    const unsigned char* wn = wcur + (t + 1) * (HNT * HROWB);
                    na0 = *reinterpret_cast<const u32x4*>(ap + toff); na1 = *reinterpret_cast<const u32x4*>(ap + toff + 32);
                    nb00 = *reinterpret_cast<const u32x4*>(wn); nb01 = *reinterpret_cast<const u32x4*>(wn + 32);
                    nb10 = *reinterpret_cast<const u32x4*>(wn + 32 * HROWB); nb11 = *reinterpret_cast<const u32x4*>(wn + 32 * HROWB + 32);
                }
                __builtin_amdgcn_sched_barrier(0);
                acc0 = mfma16<BF>(a0, b00, acc0);
                acc1 = mfma16<BF>(a0, b10, acc1);
                acc0 = mfma16<BF>(a1, b01, acc0);
                acc1 = mfma16<BF>(a1, b11, acc1);
                if (t + 1 < nTap) { a0 = na0; a1 = na1; b00 = nb00; b01 = nb01; b10 = nb10; b11 = nb11; }
            }
            if (g.dbg) { const long long tn = (long long)__builtin_readcyclecounter(); tTap += tn - tq; tq = tn; }
            if (more) store_wgroup((step + 1) & 1, nn);      // that buffer was last read in step-1, retired by its barrier
            if (lastGrp && more) {
                __syncthreads();                             // every wave is done with this chunk's halo image
                bool done = false;
                if constexpr (PREF) if (prefHalo) {
                    done = true;
#pragma unroll
                    for (int u = 0; u < HHREG; ++u) {
                        const int idx = u * 512 + tid;
                        if (idx < nHalo) {
                            u32x2 p;
                            p.x = pack2<BF>(asf(hr[u].x), asf(hr[u].y));
                            p.y = pack2<BF>(asf(hr[u].z), asf(hr[u].w));
                            *reinterpret_cast<u32x2*>(halo + (idx >> 3) * HROWB + hq * 8) = p;
                        }
                    }
                }
                if (!done) stage_halo_sync((chunk + 1) * HCK);
            }
            __syncthreads();
            if (g.dbg) tSync += (long long)__builtin_readcyclecounter() - tq;
        }
    }
    const long long ts2 = g.dbg ? (long long)__builtin_readcyclecounter() : 0;

    // ---- epilogue: D[row = voxel][col = co]; row = (r&3) + 8*(r>>2) + 4*h ----
    const int co0 = n0 + l31, co1 = n0 + 32 + l31;
    const float bias0 = (bias && co0 < g.Cout) ? bias[co0] : 0.f;
    const float bias1 = (bias && co1 < g.Cout) ? bias[co1] : 0.f;
    const auto rs_y = __builtin_amdgcn_make_buffer_rsrc(y, 0, (int)g.yBytes, 0x00020000);
    const auto rs_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(residual), 0, residual ? (int)g.yBytes : 0, 0x00020000);
    const unsigned c0 = co0 < g.Cout ? (unsigned)co0 * 4u : HBUF_OOB_C, c1 = co1 < g.Cout ? (unsigned)co1 * 4u : HBUF_OOB_C;
    // residual: all 32 loads in flight before the first add (a load + wait + add per element serialises 16 L2 round trips; the
    // pseudo-3D blocks under autocast pass their `+ res` here)
    float rr0[16], rr1[16];
    if (residual) {            // kernel-uniform
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
        if (g.roundOut) { v0 = round_through<BF>(v0); v1 = round_through<BF>(v1); }
        if (residual) { v0 += rr0[r]; v1 += rr1[r]; }
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v0), rs_y, off + c0, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v1), rs_y, off + c1, 0, 0);
    }
    if (g.dbg && tid == 0) {
        unsigned long long* d = g.dbg + (size_t)blockIdx.x * 8;
        d[0] = (unsigned long long)(ts1 - ts0); d[1] = (unsigned long long)tTap; d[2] = (unsigned long long)tSync;
        d[3] = (unsigned long long)((long long)__builtin_readcyclecounter() - ts2); d[4] = (unsigned long long)((long long)__builtin_readcyclecounter() - ts0);
        d[5] = (unsigned long long)step;
    }
}

// The same kernel, PERSISTENT: a workgroup walks a contiguous range of (tile, channel-block) units, so that the prologue of the
// one-unit kernel above -- coordinate tables, the first halo chunk and the first weight group, ~25 % of a workgroup's lifetime with
// nothing else resident on the CU to hide it -- is paid once: the next unit's first halo chunk is prefetched into registers during
// the last tap group of the current unit, the weight pipeline runs on across units, and the output stores of a unit leave before
// its closing barrier.  Per-thread halo coordinates are fixed for the whole kernel (the tile geometry is), so a unit's source
// offsets are a few integer operations per piece instead of an LDS table.  Only for halo tiles that fit the register prefetch.
// XH: x holds 16-bit values (the operand type) instead of fp32 -- a halo piece is 8 channels and goes to the LDS image as it is;
// YH: y is stored as 16-bit values (round_out, no residual).  The pair of convs of a pseudo-3D block passes its intermediate tensor this
// way: the values are the ones the fp32 tensor would hold (already rounded to the operand type), at half the bytes.
// MW waves per workgroup: 8 (a wave owns 32 voxels x 64 channels) or 4 with TWO workgroups per CU (a wave owns 64 voxels x 64 channels:
// 8 fragment reads per 8 MFMAs instead of 6 per 4 -- the 8-wave form asks 187 B/clk of the 128 B/clk LDS; one weight buffer, so that
// two workgroups fit, and the other workgroup covers the refill)
template <bool BF, int OCC, int NHR, int TGM, bool XH, bool YH, bool STATS = false, int MW = 8>       // workgroups per CU, halo pieces per thread, most taps per weight group
__global__ __launch_bounds__(64 * MW, MW == 4 ? 2 : 2 * OCC) void conv_fwd_hp_kernel(const float* __restrict__ x, const unsigned short* __restrict__ wp,
                                                              const float* __restrict__ bias, const float* __restrict__ residual,
                                                              float* __restrict__ y, HalfGeom g, int nUnits, int perWg,
                                                              float* __restrict__ stats) {
    // stats (optional): column sums (sum, sum of squares) of the STORED values per (tile, wave) for the consumer's GroupNorm:
    // [B][tiles per batch * 8][2][Cout]
    constexpr int NT = 64 * MW, RB = 8 / MW, WB = MW == 4 ? 1 : 2;      // threads, 32-voxel row blocks per wave, weight buffers
    constexpr int NWR = (TGM * 256 + NT - 1) / NT;
    extern __shared__ __attribute__((aligned(16))) unsigned char hsm[];
    const int HV = g.HD * g.HH * g.HWd;
    unsigned char* halo = hsm;                                         // [HV][80 B]
    const int sliceBytes = HV * HROWB;                                 // one 32-channel halo image
    unsigned char* wbuf = hsm + (size_t)g.NS * sliceBytes;             // [2][TG][64][80 B]
    const int wbufBytes = g.TG * HNT * HROWB;
    int* out_off = reinterpret_cast<int*>(wbuf + WB * (size_t)wbufBytes);  // [2][256]: by unit parity

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int T = g.kd * g.kh * g.kw;
    constexpr int PPR = XH ? 4 : 8, PSH = XH ? 2 : 3;      // 16-byte pieces per 32-channel halo row
    constexpr unsigned XE = XH ? 2u : 4u, YE = YH ? 2u : 4u;   // bytes per element
    const int hq = tid & (PPR - 1);
    const int nSlice = HV * PPR;                    // pieces of one 32-channel image
    const int nHalo = nSlice * g.NS;

    const int uBegin = blockIdx.x * perWg, uEnd = min(uBegin + perWg, nUnits);
    if (uBegin >= uEnd) return;

    // halo pieces of this thread, fixed for the kernel: the coordinates inside the halo tile as three 10-bit fields (range tests of a
    // unit's source voxel are then two subtractions on guarded fields instead of six compares) and the byte offset inside the tile
    constexpr unsigned GUARD = (1u << 9) | (1u << 19) | (1u << 29), ONES = 16u | (16u << 10) | (16u << 20);      // ONES: PADB in every field
    unsigned pc[NHR], prel[NHR];
#pragma unroll
    for (int u = 0; u < NHR; ++u) {
        const int idx = u * NT + tid, hv = min((idx >= nSlice ? idx - nSlice : idx) >> PSH, HV - 1);      // NS <= 2
        const int hx = hv % g.HWd, hy = (hv / g.HWd) % g.HH, hz = hv / (g.HWd * g.HH);
        pc[u] = idx < nHalo ? ((unsigned)hz | ((unsigned)hy << 10) | ((unsigned)hx << 20)) : 0x3fffffffu;     // past the tile: never in range
        prel[u] = (unsigned)(((hz * g.H + hy) * g.W + hx) * g.Cin) * XE;
    }
    constexpr int PADB = 16;
    const unsigned limits = (unsigned)(g.D + PADB - 1) | ((unsigned)(g.H + PADB - 1) << 10) | ((unsigned)(g.W + PADB - 1) << 20);
    const bool wide = g.D > 448 || g.H > 448 || g.W > 448 || g.pd > PADB || g.ph > PADB || g.pw > PADB;
    const int otw = tid % g.TW, oth = (tid / g.TW) % g.TH, otd = tid / (g.TW * g.TH);      // tid < 256: this thread's out_off entry

    struct Unit { int b, d0, h0, w0, n0, nt, tx, ty, tz; };
    auto decode = [&](int L) {
        Unit t;
        t.nt = L % g.nNt;
        int mt = L / g.nNt;
        t.tx = mt % g.tilesW; mt /= g.tilesW;
        t.ty = mt % g.tilesH; mt /= g.tilesH;
        t.tz = mt % g.tilesD;
        t.b = mt / g.tilesD;
        t.n0 = t.nt * HNT; t.w0 = t.tx * g.TW; t.h0 = t.ty * g.TH; t.d0 = t.tz * g.TD;
        return t;
    };
    auto successor = [&](Unit t) {          // the next unit of the walk, without the divisions of decode()
        if (++t.nt == g.nNt) {
            t.nt = 0;
            if (++t.tx == g.tilesW) {
                t.tx = 0;
                if (++t.ty == g.tilesH) {
                    t.ty = 0;
                    if (++t.tz == g.tilesD) { t.tz = 0; ++t.b; }
                }
            }
        }
        t.n0 = t.nt * HNT; t.w0 = t.tx * g.TW; t.h0 = t.ty * g.TH; t.d0 = t.tz * g.TD;
        return t;
    };
    auto write_out_table = [&](const Unit& t, int par) {
        if (tid < HMT) {
            const int od = t.d0 + otd, oh = t.h0 + oth, ow = t.w0 + otw;
            int off = (int)HBUF_OOB;
            if (od < g.Do && oh < g.Ho && ow < g.Wo) off = (((t.b * g.Do + od) * g.Ho + oh) * g.Wo + ow) * g.Cout * (int)YE;
            out_off[par * HMT + tid] = off;
        }
    };
    unsigned srcv[NHR];
    auto unit_sources = [&](const Unit& t) {
        // source voxel of a piece = tile origin - pad + piece coordinates, per axis in [0, extent): with c = coordinate + PADB >= 0 in a
        // guarded field, (c | guard) - PADB keeps the guard bit iff c >= PADB and (extent + PADB - 1 | guard) - c keeps it iff c <= that
        const unsigned org = (unsigned)(t.d0 - g.pd + PADB) + ((unsigned)(t.h0 - g.ph + PADB) << 10) + ((unsigned)(t.w0 - g.pw + PADB) << 20);
        const unsigned base = (unsigned)((((t.b * g.D + t.d0 - g.pd) * g.H + t.h0 - g.ph) * g.W + t.w0 - g.pw) * g.Cin) * XE;
        if (!wide) {
#pragma unroll
            for (int u = 0; u < NHR; ++u) {
                const unsigned c = pc[u] + org;                   // fields stay below 512 (extents <= 448, PADB, a halo of a few voxels)
                const unsigned ok = ((c | GUARD) - ONES) & ((limits | GUARD) - c) & GUARD;
                srcv[u] = (ok == GUARD && pc[u] != 0x3fffffffu) ? base + prel[u] : HBUF_OOB;
            }
        } else {                                                  // an extent beyond the fields (1x1x1 convs arrive as one long row axis)
#pragma unroll
            for (int u = 0; u < NHR; ++u) {
                const int hz = pc[u] & 1023, hy = (pc[u] >> 10) & 1023, hx = (pc[u] >> 20) & 1023;
                const int iz = t.d0 + hz - g.pd, iy = t.h0 + hy - g.ph, ix = t.w0 + hx - g.pw;
                const bool ok = pc[u] != 0x3fffffffu && iz >= 0 && iz < g.D && iy >= 0 && iy < g.H && ix >= 0 && ix < g.W;
                srcv[u] = ok ? base + prel[u] : HBUF_OOB;
            }
        }
    };
    const auto rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, (int)g.xBytes, 0x00020000);
    const auto rs_y = __builtin_amdgcn_make_buffer_rsrc(y, 0, (int)g.yBytes, 0x00020000);
    const auto rs_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(residual), 0, residual ? (int)g.yBytes : 0, 0x00020000);

    u32x4 hr[NHR];
    auto halo_load = [&](int ci0) {
#pragma unroll
        for (int u = 0; u < NHR; ++u) {
            const int ch = ci0 + (u * NT + tid >= nSlice ? HCK : 0) + hq * (32 / PPR);   // second slice of a pointwise step: the next 32 channels
            const unsigned coff = ch < g.Cin ? (unsigned)ch * XE : HBUF_OOB_C;
            hr[u] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, srcv[u] + coff, 0, 0);
        }
    };
    auto halo_store = [&]() {
#pragma unroll
        for (int u = 0; u < NHR; ++u) {
            const int idx = u * NT + tid;
            if (idx < nHalo) {
                unsigned char* img = idx >= nSlice ? halo + sliceBytes + ((idx - nSlice) >> PSH) * HROWB : halo + (idx >> PSH) * HROWB;
                if constexpr (XH) {
                    *reinterpret_cast<u32x4*>(img + hq * 16) = hr[u];
                } else {
                    u32x2 p;
                    p.x = pack2<BF>(asf(hr[u].x), asf(hr[u].y));
                    p.y = pack2<BF>(asf(hr[u].z), asf(hr[u].w));
                    *reinterpret_cast<u32x2*>(img + hq * 8) = p;
                }
            }
        }
    };
    u32x4 wr[NWR];
    auto load_wgroup = [&](int chunk, int t0, int n, int n0) {
        const unsigned short* src = wp + ((size_t)chunk * T * g.CoutPad + n0) * HCK;
#pragma unroll
        for (int u = 0; u < NWR; ++u) {
            const int idx = u * NT + tid;
            const int tap = min(idx >> 8, n - 1), row = (idx >> 2) & 63, q = idx & 3;
            wr[u] = *reinterpret_cast<const u32x4*>(src + ((size_t)(t0 + tap) * g.CoutPad + row) * HCK + q * 8);
        }
    };
    auto store_wgroup = [&](int buf, int n) {
        unsigned char* dst = wbuf + buf * wbufBytes;
#pragma unroll
        for (int u = 0; u < NWR; ++u) {
            const int idx = u * NT + tid;
            const int tap = idx >> 8, row = (idx >> 2) & 63, q = idx & 3;
            if (tap < n) *reinterpret_cast<u32x4*>(dst + (tap * HNT + row) * HROWB + q * 16) = wr[u];
        }
    };

    int a_base[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        const int v = (wave * RB + rb) * 32 + l31;
        const int tw = v % g.TW, th = (v / g.TW) % g.TH, td = v / (g.TW * g.TH);
        a_base[rb] = ((td * g.HH + th) * g.HWd + tw) * HROWB + h * 16;
    }
    const int b_base = l31 * HROWB + h * 16;

    f32x16 accA[RB], accB[RB];                      // output channels n0 + l31 / n0 + 32 + l31 of each row block
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int i = 0; i < 16; ++i) { accA[rb][i] = 0.f; accB[rb][i] = 0.f; }

    Unit cur = decode(uBegin);
    write_out_table(cur, 0);
    unit_sources(cur);
    const bool pww = g.NS > 1;                      // pointwise filter: a step's "taps" are NS consecutive channel chunks
    const int firstTaps = pww ? min(g.NS, g.realChunks) : min(g.TG, T);
    halo_load(0);
    load_wgroup(0, 0, firstTaps, cur.n0);
    halo_store();
    store_wgroup(0, firstTaps);
    __syncthreads();

    int step = 0;
    for (int un = uBegin; un < uEnd; ++un) {
        const bool hasNext = un + 1 < uEnd;
        Unit nxt = cur;
        if (hasNext) {
            nxt = successor(cur);
            write_out_table(nxt, (un + 1 - uBegin) & 1);       // read in the next unit's epilogue: at least this unit's closing barrier between
        }
        for (int chunk = 0; chunk < g.nChunks; ++chunk) {
            for (int grp = 0; grp < g.nGroups; ++grp, ++step) {
                const int t0 = grp * g.TG, nTap = pww ? min(g.NS, g.realChunks - chunk * g.NS) : min(g.TG, T - t0);
                const bool lastGrp = grp + 1 == g.nGroups, lastStep = lastGrp && chunk + 1 == g.nChunks;
                const bool more = !lastStep || hasNext;
                const int nchunk = lastStep ? 0 : (lastGrp ? chunk + 1 : chunk), nt0 = lastGrp ? 0 : t0 + g.TG;
                const int nn = pww ? min(g.NS, g.realChunks - nchunk * g.NS) : min(g.TG, T - nt0);
                if (more && WB == 2) load_wgroup(nchunk * g.NS, nt0, nn, lastStep ? nxt.n0 : cur.n0);
                const bool pref = lastGrp && more;
                if (pref) {
                    if (lastStep) unit_sources(nxt);
                    halo_load(lastStep ? 0 : (chunk + 1) * HCK * g.NS);
                }
                const unsigned char* wcur = wbuf + (WB == 2 ? (step & 1) : 0) * wbufBytes + b_base;
                const unsigned char* ap = halo;
                int kx = t0 % g.kw, ky = (t0 / g.kw) % g.kh, kz = t0 / (g.kw * g.kh);
                int toff = ((kz * g.HH + ky) * g.HWd + kx) * HROWB;
                auto next_tap = [&]() {
                    if (pww) { toff += sliceBytes; return; }
                    toff += HROWB;
                    if (++kx == g.kw) {
                        kx = 0; toff += (g.HWd - g.kw) * HROWB;
                        if (++ky == g.kh) { ky = 0; toff += (g.HH - g.kh) * g.HWd * HROWB; }
                    }
                };
                if constexpr (MW == 4) {
                    // two workgroups per CU: the other workgroup's waves cover a tap's fragment reads (no register double-buffering)
                    for (int t = 0; t < nTap; ++t) {
                        const unsigned char* wn = wcur + t * (HNT * HROWB);
                        const u32x4 b00 = *reinterpret_cast<const u32x4*>(wn), b01 = *reinterpret_cast<const u32x4*>(wn + 32);
                        const u32x4 b10 = *reinterpret_cast<const u32x4*>(wn + 32 * HROWB), b11 = *reinterpret_cast<const u32x4*>(wn + 32 * HROWB + 32);
#pragma unroll
                        for (int rb = 0; rb < RB; ++rb) {
                            const u32x4 a0 = *reinterpret_cast<const u32x4*>(ap + a_base[rb] + toff), a1 = *reinterpret_cast<const u32x4*>(ap + a_base[rb] + toff + 32);
                            accA[rb] = mfma16<BF>(a0, b00, accA[rb]);
                            accB[rb] = mfma16<BF>(a0, b10, accB[rb]);
                            accA[rb] = mfma16<BF>(a1, b01, accA[rb]);
                            accB[rb] = mfma16<BF>(a1, b11, accB[rb]);
                        }
                        next_tap();
                    }
                } else {
                u32x4 a0 = *reinterpret_cast<const u32x4*>(ap + a_base[0] + toff), a1 = *reinterpret_cast<const u32x4*>(ap + a_base[0] + toff + 32);
                u32x4 b00 = *reinterpret_cast<const u32x4*>(wcur), b01 = *reinterpret_cast<const u32x4*>(wcur + 32);
                u32x4 b10 = *reinterpret_cast<const u32x4*>(wcur + 32 * HROWB), b11 = *reinterpret_cast<const u32x4*>(wcur + 32 * HROWB + 32);
                __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0): see conv_fwd_h_kernel
                for (int t = 0; t < nTap; ++t) {
                    u32x4 na0, na1, nb00, nb01, nb10, nb11;
                    if (t + 1 < nTap) {
                        next_tap();
                        const unsigned char* wn = wcur + (t + 1) * (HNT * HROWB);
                        na0 = *reinterpret_cast<const u32x4*>(ap + a_base[0] + toff); na1 = *reinterpret_cast<const u32x4*>(ap + a_base[0] + toff + 32);
                        nb00 = *reinterpret_cast<const u32x4*>(wn); nb01 = *reinterpret_cast<const u32x4*>(wn + 32);
                        nb10 = *reinterpret_cast<const u32x4*>(wn + 32 * HROWB); nb11 = *reinterpret_cast<const u32x4*>(wn + 32 * HROWB + 32);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    accA[0] = mfma16<BF>(a0, b00, accA[0]);
                    accB[0] = mfma16<BF>(a0, b10, accB[0]);
                    accA[0] = mfma16<BF>(a1, b01, accA[0]);
                    accB[0] = mfma16<BF>(a1, b11, accB[0]);
                    if (t + 1 < nTap) { a0 = na0; a1 = na1; b00 = nb00; b01 = nb01; b10 = nb10; b11 = nb11; }
                }
                }
                if (lastStep) {
                    // ---- epilogue of this unit (before its closing barrier: the stores drain while the next unit is staged) ----
                    const int* oo0 = out_off + ((un - uBegin) & 1) * HMT + wave * (32 * RB) + 4 * h;
                    const int co0 = cur.n0 + l31, co1 = cur.n0 + 32 + l31;
                    const float bias0 = (bias && co0 < g.Cout) ? bias[co0] : 0.f;
                    const float bias1 = (bias && co1 < g.Cout) ? bias[co1] : 0.f;
                    const unsigned c0 = co0 < g.Cout ? (unsigned)co0 * YE : HBUF_OOB_C, c1 = co1 < g.Cout ? (unsigned)co1 * YE : HBUF_OOB_C;
                    float st0 = 0.f, sq0 = 0.f, st1 = 0.f, sq1 = 0.f;
                    auto epilogue = [&](auto ROUND, auto RES) {
#pragma unroll
                      for (int rb = 0; rb < RB; ++rb) {
                        const int* oo = oo0 + 32 * rb;
                        f32x16& acc0 = accA[rb];
                        f32x16& acc1 = accB[rb];
                        float rr0[16], rr1[16];
                        if constexpr (RES.value) {
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const unsigned off = (unsigned)oo[(r & 3) + 8 * (r >> 2)];
                                rr0[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_r, off + c0, 0, 0));
                                rr1[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_r, off + c1, 0, 0));
                            }
                        }
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const unsigned off = (unsigned)oo[(r & 3) + 8 * (r >> 2)];
                            float v0 = acc0[r] + bias0, v1 = acc1[r] + bias1;
                            if constexpr (ROUND.value) { v0 = round_through<BF>(v0); v1 = round_through<BF>(v1); }
                            if constexpr (RES.value) { v0 += rr0[r]; v1 += rr1[r]; }
                            if constexpr (STATS) {
                                const float m = off != HBUF_OOB ? 1.f : 0.f;      // rows of a ragged tile beyond the volume
                                if constexpr (YH) { v0 = round_through<BF>(v0); v1 = round_through<BF>(v1); }
                                st0 += m * v0; sq0 += m * v0 * v0; st1 += m * v1; sq1 += m * v1 * v1;
                            }
                            if constexpr (YH) {
                                __builtin_amdgcn_raw_buffer_store_b16((unsigned short)(pack2<BF>(v0, 0.f) & 0xffffu), rs_y, off + c0, 0, 0);
                                __builtin_amdgcn_raw_buffer_store_b16((unsigned short)(pack2<BF>(v1, 0.f) & 0xffffu), rs_y, off + c1, 0, 0);
                            } else {
                                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v0), rs_y, off + c0, 0, 0);
                                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v1), rs_y, off + c1, 0, 0);
                            }
                            acc0[r] = 0.f; acc1[r] = 0.f;
                        }
                      }
                    };
                    // kernel-uniform switches as branches (as selects they cost two v_cndmask per output element)
                    if (g.roundOut) {
                        if (residual) epilogue(std::true_type{}, std::true_type{}); else epilogue(std::true_type{}, std::false_type{});
                    } else {
                        if (residual) epilogue(std::false_type{}, std::true_type{}); else epilogue(std::false_type{}, std::false_type{});
                    }
                    if constexpr (STATS) {
                        st0 += __shfl_xor(st0, 32, 64); sq0 += __shfl_xor(sq0, 32, 64);
                        st1 += __shfl_xor(st1, 32, 64); sq1 += __shfl_xor(sq1, 32, 64);
                        if (h == 0) {
                            const int nblk = g.tilesD * g.tilesH * g.tilesW * MW;
                            const int blk = ((cur.tz * g.tilesH + cur.ty) * g.tilesW + cur.tx) * MW + wave;
                            float* sp = stats + ((size_t)cur.b * nblk + blk) * 2 * g.Cout;
                            if (co0 < g.Cout) { sp[co0] = st0; sp[g.Cout + co0] = sq0; }
                            if (co1 < g.Cout) { sp[co1] = st1; sp[g.Cout + co1] = sq1; }
                        }
                    }
                }
                if (WB == 2) {
                    if (more) store_wgroup((step + 1) & 1, nn);
                    if (pref) {
                        __syncthreads();                 // every wave is done with this chunk's halo image
                        halo_store();
                    }
                } else {                                 // one weight buffer: refilled, like the halo image, once every wave has left the tap loop
                    __syncthreads();
                    if (more) {                          // loaded HERE, not before the tap loop: 36 registers less across the loop
                        load_wgroup(nchunk * g.NS, nt0, nn, lastStep ? nxt.n0 : cur.n0);
                        store_wgroup(0, nn);
                    }
                    if (pref) halo_store();
                }
                __syncthreads();
            }
        }
        cur = nxt;
    }
}

static size_t half_lds_bytes(const HalfGeom& g, int wbufs = 2) {
    const size_t HV = (size_t)g.HD * g.HH * g.HWd;
    const size_t tables = (HMT + HV) > 2 * HMT ? (HMT + HV) : 2 * HMT;      // one-unit kernel: out_off + halo_src; persistent: out_off x 2
    return (size_t)g.NS * HV * HROWB + (size_t)wbufs * g.TG * HNT * HROWB + tables * sizeof(int);
}

static bool half_geom(HalfGeom& g, int B, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw, int pd, int ph, int pw,
                      int epd, int eph, int epw, int xe = 4, int ye = 4) {      // xe, ye: bytes per element of x and y
    if (B <= 0 || D <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || kd <= 0 || kh <= 0 || kw <= 0) return false;
    if (pd < 0 || ph < 0 || pw < 0 || pd + epd < 0 || ph + eph < 0 || pw + epw < 0) return false;
    if (Cin % 4 != 0) return false;
    if (Cin < 8 && kd * kh * kw > 1) return false;          // the tap-packed fp32 kernel (conv_fwd_smallcin_kernel) is the better fit
    if (kd == 1 && kh == 1 && kw == 1 && pd == 0 && ph == 0 && pw == 0 && epd == 0 && eph == 0 && epw == 0) {   // 1x1x1: flatten all voxels into W
        const long long rows = (long long)B * D * H * W;
        if (rows >= (1ll << 31)) return false;
        B = 1; D = 1; H = 1; W = (int)rows;
    }
    g.B = B; g.D = D; g.H = H; g.W = W; g.Cin = Cin; g.Cout = Cout;
    g.kd = kd; g.kh = kh; g.kw = kw; g.pd = pd; g.ph = ph; g.pw = pw;
    g.Do = D + 2 * pd + epd - kd + 1; g.Ho = H + 2 * ph + eph - kh + 1; g.Wo = W + 2 * pw + epw - kw + 1;
    if (g.Do <= 0 || g.Ho <= 0 || g.Wo <= 0) return false;
    const int T = kd * kh * kw;
    g.TG = T < HTG ? T : HTG;
    g.nGroups = hcdiv(T, g.TG);
    g.nNt = hcdiv(Cout, HNT); g.CoutPad = g.nNt * HNT; g.nChunks = hcdiv(Cin, HCK);
    g.roundOut = 1;
    g.dbg = nullptr;
    g.NS = 1; g.realChunks = g.nChunks;
    static const int cand[][3] = {{4, 8, 8}, {8, 8, 4}, {8, 4, 8}, {2, 8, 16}, {2, 16, 8}, {1, 16, 16}, {16, 4, 4}, {4, 4, 16}, {4, 16, 4},
                                  {16, 16, 1}, {16, 1, 16}, {32, 4, 2}, {64, 2, 2}, {256, 1, 1}, {1, 1, 256}, {1, 256, 1}, {1, 8, 32},
                                  {1, 32, 8}, {8, 32, 1}, {32, 8, 1}, {1, 4, 64}, {1, 2, 128}, {128, 2, 1}, {128, 1, 2}};
    double best = 1e300;
    bool found = false;
    for (auto& c : cand) {
        HalfGeom t = g;
        t.TD = c[0]; t.TH = c[1]; t.TW = c[2];
        t.HD = c[0] + kd - 1; t.HH = c[1] + kh - 1; t.HWd = c[2] + kw - 1;
        if (half_lds_bytes(t) > 160 * 1024) continue;
        const double tiles = (double)hcdiv(g.Do, c[0]) * hcdiv(g.Ho, c[1]) * hcdiv(g.Wo, c[2]);
        const double halo = (double)t.HD * t.HH * t.HWd;
        const double cost = tiles * (halo * 1.0 + 256.0 * T);     // staging is relatively 16x dearer than on the f32 kernel
        if (cost < best) { best = cost; g.TD = c[0]; g.TH = c[1]; g.TW = c[2]; found = true; }
    }
    if (!found) return false;
    g.HD = g.TD + kd - 1; g.HH = g.TH + kh - 1; g.HWd = g.TW + kw - 1;
    g.tilesD = hcdiv(g.Do, g.TD); g.tilesH = hcdiv(g.Ho, g.TH); g.tilesW = hcdiv(g.Wo, g.TW);
    const long long nwg = (long long)g.B * g.tilesD * g.tilesH * g.tilesW * g.nNt;
    if (nwg >= (1ll << 31)) return false;
    const unsigned long long xb = (unsigned long long)g.B * g.D * g.H * g.W * g.Cin * (unsigned long long)xe;
    const unsigned long long yb = (unsigned long long)g.B * g.Do * g.Ho * g.Wo * g.Cout * (unsigned long long)ye;
    if (xb >= (1ull << 30) || yb >= (1ull << 30)) return false;       // buffer-descriptor addressing
    g.xBytes = (unsigned)xb; g.yBytes = (unsigned)yb;
    return true;
}


// ---------------------------------------------------------------------------------------------------------------------------------
// Pointwise (1x1x1) convs with MANY output channels as a K-blocked GEMM (round 3).  The tap-oriented kernels above give a workgroup
// 256 rows x 64 output channels: x is staged (and converted) once per 64 channels -- eight times for the 256 -> 512 query projection of
// the joint attentions -- and a step has 4-8 MFMAs of a wave per barrier pair (106-157 TFLOP/s inside the C5 cascade).  Here a
// workgroup of four waves (two workgroups per CU) owns 256 rows x 32 NCT = 128 channels: wave w = rows 64 w .. + 63, every channel
// tile; per 32-channel K chunk 4 NCT = 16 MFMAs of a wave between two barriers; x (fp32 rows, converted while
// staged) and the packed weight panel [chunk][co][32] are double-buffered in LDS, the next chunk's pieces in registers during the MFMAs.
// The channel tiles of a row tile go to the SAME XCD (workgroup ids 8 apart), so x comes from HBM once.
// ---------------------------------------------------------------------------------------------------------------------------------
struct PwhGeom {
    long long R;                 // rows (voxels)
    int Cin, Cout, CoutPad, nChunks, nRowTiles, nCoTiles, roundOut;
    int T, pd, HW, D;            // taps along the frame axis (1 or kd), their low pad, rows per frame, frames
    unsigned xBytes, yBytes, wBytes;
};

// The same kernel takes the (kd,1,1) TEMPORAL convs of the pseudo-3D blocks (imagen_video.py Conv3d: kernel (k,1,1) after the per-frame
// conv): in channels-last rows a frame shift is a shift by H W rows, so K = taps x Cin with the A rows of K step (chunk, tap) read
// (tap - pad) H W rows away (zero where the frame index leaves the volume: causal / symmetric padding); the packed weights
// [chunk][tap][co][32] are already in that K order.  XH: x rows are 16-bit (copied verbatim).
template <bool BF, int NCT, bool XH>
__global__ __launch_bounds__(256, 2) void conv_pw_h_kernel(const float* __restrict__ x, const unsigned short* __restrict__ wp,
                                                           const float* __restrict__ bias, const float* __restrict__ residual,
                                                           float* __restrict__ y, PwhGeom g) {
    constexpr int NTC = 32 * NCT, ABYTES = 256 * HROWB, BBYTES = NTC * HROWB;
    constexpr int XE = XH ? 2 : 4;
    constexpr int NPA = XH ? 4 : 8, NPB = NTC * 4 / 256;      // 16-byte pieces per thread and K step: x (256 rows x 32 ci), weights (NTC rows x 32 ci)
    constexpr int PSH = XH ? 2 : 3;
    extern __shared__ __attribute__((aligned(16))) unsigned char smpw[];
    unsigned char* const As = smpw;                  // [2][256 rows][80 B]
    unsigned char* const Bs = smpw + 2 * ABYTES;     // [2][NTC co][80 B]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const unsigned id = blockIdx.x;
    const int coT = (int)((id >> 3) % (unsigned)g.nCoTiles);
    const int rowT = (int)((id / (8u * (unsigned)g.nCoTiles)) * 8u + (id & 7u));
    if (rowT >= g.nRowTiles) return;                 // (the whole workgroup: the grid is padded to eight row tiles per XCD round)
    const long long r0 = (long long)rowT * 256;
    const int n0 = coT * NTC;
    const auto rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, (int)g.xBytes, 0x00020000);
    const auto rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(wp), 0, (int)g.wBytes, 0x00020000);
    const auto rs_y = __builtin_amdgcn_make_buffer_rsrc(y, 0, (int)g.yBytes, 0x00020000);
    const auto rs_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(residual), 0, residual ? (int)g.yBytes : 0, 0x00020000);

    unsigned aoff[NPA], boff[NPB];
    int afr[NPA];                                    // frame index of the piece's row (-1000: no row)
#pragma unroll
    for (int u = 0; u < NPA; ++u) {
        const int idx = u * 256 + tid, row = idx >> PSH, q = idx & ((1 << PSH) - 1);
        const bool ok = r0 + row < g.R;
        aoff[u] = ok ? (unsigned)((r0 + row) * g.Cin * XE + q * 16) : HBUF_OOB;
        afr[u] = ok ? (int)(((r0 + row) / g.HW) % g.D) : -1000;
    }
#pragma unroll
    for (int u = 0; u < NPB; ++u) {
        const int idx = u * 256 + tid, co = idx >> 2, q = idx & 3;
        boff[u] = n0 + co < g.CoutPad ? (unsigned)(((n0 + co) * HCK + q * 8) * 2) : HBUF_OOB;
    }
    const unsigned wStep = (unsigned)g.CoutPad * HCK * 2;      // bytes of one (chunk, tap) weight panel
    const int frameBytes = g.HW * g.Cin * XE;

    f32x16 acc[2][NCT];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[rt][ct][i] = 0.f;

    u32x4 pa[NPA], pb[NPB];
    auto gload = [&](int st, int chunk, int tap) {
        const int sh = tap - g.pd;                   // frames
#pragma unroll
        for (int u = 0; u < NPA; ++u) {
            const int f = afr[u] + sh;
            const unsigned off = (f >= 0 && f < g.D) ? aoff[u] + (unsigned)(sh * frameBytes) + (unsigned)chunk * (HCK * XE) : HBUF_OOB;
            pa[u] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, off, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < NPB; ++u) pb[u] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, boff[u] + (unsigned)st * wStep, 0, 0);
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int u = 0; u < NPA; ++u) {
            const int idx = u * 256 + tid, row = idx >> PSH, q = idx & ((1 << PSH) - 1);
            if constexpr (XH) {
                *reinterpret_cast<u32x4*>(As + buf * ABYTES + row * HROWB + q * 16) = pa[u];
            } else {
                u32x2 p;
                p.x = pack2<BF>(asf(pa[u].x), asf(pa[u].y));
                p.y = pack2<BF>(asf(pa[u].z), asf(pa[u].w));
                *reinterpret_cast<u32x2*>(As + buf * ABYTES + row * HROWB + q * 8) = p;
            }
        }
#pragma unroll
        for (int u = 0; u < NPB; ++u) {
            const int idx = u * 256 + tid, co = idx >> 2, q = idx & 3;
            *reinterpret_cast<u32x4*>(Bs + buf * BBYTES + co * HROWB + q * 16) = pb[u];
        }
    };
    const int nSteps = g.nChunks * g.T;
    int nchunk = 0, ntap = 0;                        // (chunk, tap) of the step being prefetched
    gload(0, 0, 0);
    lstore(0);
    __syncthreads();
    for (int st = 0; st < nSteps; ++st) {
        const bool more = st + 1 < nSteps;
        if (more) {                                  // in flight behind this step's MFMAs
            if (++ntap == g.T) { ntap = 0; ++nchunk; }
            gload(st + 1, nchunk, ntap);
        }
        const unsigned char* ap = As + (st & 1) * ABYTES + (64 * wave + l31) * HROWB + h * 16;
        const unsigned char* bp = Bs + (st & 1) * BBYTES + l31 * HROWB + h * 16;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const u32x4 a0 = *reinterpret_cast<const u32x4*>(ap + ks * 32), a1 = *reinterpret_cast<const u32x4*>(ap + 32 * HROWB + ks * 32);
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                const u32x4 b = *reinterpret_cast<const u32x4*>(bp + ct * 32 * HROWB + ks * 32);
                acc[0][ct] = mfma16<BF>(a0, b, acc[0][ct]);
                acc[1][ct] = mfma16<BF>(a1, b, acc[1][ct]);
            }
        }
        if (more) lstore((st + 1) & 1);              // the other buffer: every wave left it before the previous barrier
        __syncthreads();
    }
    // ---- epilogue: D[row = voxel][col = co], row = (r & 3) + 8 (r >> 2) + 4 h of the wave's row tile ----
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
        const int co = n0 + 32 * ct + l31;
        const bool cok = co < g.Cout;
        const float bv = (bias && cok) ? bias[co] : 0.f;
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            // residual: the 16 loads of a tile in flight before the first add (a load + wait + add per element serialises 16 L2 round trips)
            unsigned offs[16];
            float rr[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long row = r0 + 64 * wave + 32 * rt + (r & 3) + 8 * (r >> 2) + 4 * h;
                offs[r] = (cok && row < g.R) ? (unsigned)((row * g.Cout + co) * 4) : HBUF_OOB;
                rr[r] = 0.f;
            }
            if (residual) {                          // kernel-uniform
#pragma unroll
                for (int r = 0; r < 16; ++r) rr[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_r, offs[r], 0, 0));
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[rt][ct][r] + bv;
                if (g.roundOut) v = round_through<BF>(v);
                v += rr[r];
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs_y, offs[r], 0, 0);
            }
        }
    }
}

}  // namespace diqt

using namespace diqt;

static unsigned long long* g_hdbg = nullptr;    // diagnostic only (DIQT_CONVH_DBG=1)
static unsigned g_hdbg_n = 0;
// diagnostic only (not part of include/diqt.h): per-workgroup cycle stamps of the last DIQT_CONVH_DBG=1 launch:
// [prologue, tap loops, store + barrier waits, epilogue, lifetime, steps, -, -]
extern "C" int diqt_debug_convh_stamps(unsigned long long* host_out, unsigned max_wg) {
    if (!g_hdbg || !g_hdbg_n) return 0;
    const unsigned n = g_hdbg_n < max_wg ? g_hdbg_n : max_wg;
    if (hipDeviceSynchronize() != hipSuccess) return 0;
    if (hipMemcpy(host_out, g_hdbg, (size_t)n * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return 0;
    return (int)n;
}

// number of 16-bit elements of the packed low-precision weight (same [chunk][tap][co pad 64][32] order as the fp32 packing)
extern "C" size_t diqt_conv_packed_h_elems(int Cout, int Cin, int kd, int kh, int kw) {
    if (Cout <= 0 || Cin <= 0 || kd <= 0 || kh <= 0 || kw <= 0) return 0;
    return (size_t)hcdiv(Cin, HCK) * kd * kh * kw * (hcdiv(Cout, HNT) * HNT) * HCK;
}

extern "C" int diqt_conv_pack_weight_h(const float* w, void* packed, int Cout, int Cin, int kd, int kh, int kw, int mode, int bf16,
                                       void* stream) {
    DIQT_REQUIRE(w && packed, DIQT_E_ALIGN, "conv_pack_weight_h: null pointer");
    DIQT_REQUIRE(Cout > 0 && Cin > 0 && kd > 0 && kh > 0 && kw > 0 && (mode == 0 || mode == 1), DIQT_E_SHAPE, "conv_pack_weight_h: bad shape/mode");
    const int T = kd * kh * kw, outEff = mode == 0 ? Cout : Cin, inEff = mode == 0 ? Cin : Cout;
    const int CoutPad = hcdiv(outEff, HNT) * HNT;
    const size_t total = (size_t)hcdiv(inEff, HCK) * T * CoutPad * HCK;
    auto k = bf16 ? conv_pack_weight_h_kernel<true> : conv_pack_weight_h_kernel<false>;
    hipLaunchKernelGGL(k, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, w, static_cast<unsigned short*>(packed), Cout,
                       Cin, T, mode, CoutPad, total);
    return check_launch("conv_pack_weight_h");
}

// host_table: rows of 8 x int64 {w (device pointer), packed (device pointer), Cout, Cin, kd, kh, kw, mode}; the same results as `count`
// calls of diqt_conv_pack_weight_h
extern "C" int diqt_conv_pack_weight_h_multi(const long long* host_table, int count, int bf16, void* stream) {
    DIQT_REQUIRE(host_table && count >= 0, DIQT_E_ALIGN, "conv_pack_weight_h_multi: null table");
    for (int lo = 0; lo < count; lo += PK_ROWS) {
        const int m = count - lo < PK_ROWS ? count - lo : PK_ROWS;
        PackTable t;
        for (int i = 0; i < PK_ROWS; ++i) {
            const long long* e = host_table + 8 * (size_t)(lo + (i < m ? i : 0));
            const int Cout = (int)e[2], Cin = (int)e[3], kd = (int)e[4], kh = (int)e[5], kw = (int)e[6], mode = (int)e[7];
            DIQT_REQUIRE(e[0] && e[1], DIQT_E_ALIGN, "conv_pack_weight_h_multi: null pointer in row %d", lo + i);
            DIQT_REQUIRE(Cout > 0 && Cin > 0 && kd > 0 && kh > 0 && kw > 0 && (mode == 0 || mode == 1), DIQT_E_SHAPE,
                         "conv_pack_weight_h_multi: bad shape/mode in row %d", lo + i);
            const int T = kd * kh * kw, outEff = mode == 0 ? Cout : Cin, inEff = mode == 0 ? Cin : Cout;
            const int CoutPad = hcdiv(outEff, HNT) * HNT;
            const size_t total = (size_t)hcdiv(inEff, HCK) * T * CoutPad * HCK;
            DIQT_REQUIRE(total < (1ull << 31), DIQT_E_SHAPE, "conv_pack_weight_h_multi: weight too large in row %d", lo + i);
            t.r[i] = PackRow{reinterpret_cast<const float*>(e[0]), reinterpret_cast<unsigned short*>(e[1]), Cout, Cin, T, mode, CoutPad,
                             (unsigned)total};
        }
        auto k = bf16 ? conv_pack_weight_h_multi_kernel<true> : conv_pack_weight_h_multi_kernel<false>;
        hipLaunchKernelGGL(k, dim3(128, m), dim3(256), 0, (hipStream_t)stream, t);     // 128 workgroups per weight: the largest (256 x 256 x 27) sets the pace
        const int rc = check_launch("conv_pack_weight_h_multi");
        if (rc != DIQT_OK) return rc;
    }
    return DIQT_OK;
}

// 1 when diqt_conv3d_fwd_h takes this shape (Cin % 4 == 0, tensors < 1 GiB, halo tile within the LDS), else 0: the caller then
// stays on the fp32 kernel, which is always correct under autocast (more precise than the reference)
extern "C" int diqt_conv3d_fwd_h_supported(int B, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw, int pd, int ph, int pw,
                                           int epd, int eph, int epw) {
    HalfGeom g;
    return half_geom(g, B, D, H, W, Cin, Cout, kd, kh, kw, pd, ph, pw, epd, eph, epw) ? 1 : 0;
}

// workgroups of the persistent low-precision kernel (default 256 = one per CU; DIQT_CONVH_WGS): a launch of fewer than twice as many
// (tile, channel-block) units stays on the one-unit kernel.  n > 0 sets it (tests force the persistent walk on small shapes), n <= 0
// only queries; returns the previous value.
static std::atomic<int> g_convh_wgs{0};
extern "C" int diqt_set_convh_workgroups(int n) {
    int cur = g_convh_wgs.load();
    if (cur == 0) {
        const char* e = getenv("DIQT_CONVH_WGS");
        const int v = e && atoi(e) > 0 ? atoi(e) : 256;
        g_convh_wgs.compare_exchange_strong(cur, v);
        cur = g_convh_wgs.load();
    }
    if (n > 0) g_convh_wgs.store(n);
    return cur;
}

static bool convh_persistent_takes(const HalfGeom& g, unsigned nwg) {
    static const bool persist = [] { const char* e = getenv("DIQT_CONVH_PERSIST"); return !(e && e[0] == '0'); }();
    static const bool nopref = [] { const char* e = getenv("DIQT_CONVH_NOPREF"); return e && e[0] == '1'; }();
    const int HV = g.HD * g.HH * g.HWd;
    // (filters of more than one tap group -- 3x3x3 -- stay on the one-unit kernel: 93.9 vs 102.3 us on 64 -> 64 @ 8 x 32^3, the walk's prefetch
    // registers spill there and 4 units per workgroup amortise little)
    return persist && !nopref && g.nGroups == 1 && HV * 8 <= 512 * HHREG && nwg >= 2u * (unsigned)diqt_set_convh_workgroups(0);
}

// 16-bit input on the persistent kernel: 4-wave workgroups (64 voxels x 64 channels per wave, one weight buffer), two per CU, when both fit
static bool convh_four_waves(const HalfGeom& g, bool xh) {
    static const bool off = [] { const char* e = getenv("DIQT_CONVH_W8"); return e && e[0] == '1'; }();
    static const bool pw4 = [] { const char* e = getenv("DIQT_CONVH_PW4"); return !(e && e[0] == '0'); }();
    const int HV = g.HD * g.HH * g.HWd;
    if (off || half_lds_bytes(g, 1) > 80 * 1024 - 1024) return false;
    if (g.kd * g.kh * g.kw > 1) return xh && HV * 4 <= 256 * 6;
    return pw4 && !xh && g.NS == 1 && HV * 8 <= 256 * 8;     // pointwise, fp32 rows, one chunk per step: 8 pieces per thread
}

// 1 when diqt_conv3d_fwd_h_io takes 16-bit x and / or y for this shape: the persistent kernel's conditions and Cin, Cout % 8 == 0
extern "C" int diqt_conv3d_fwd_h_io16_supported(int B, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw, int pd, int ph,
                                                int pw, int epd, int eph, int epw, int x_half, int y_half) {
    HalfGeom g;
    if (x_half) {                 // the LDS-DMA kernel (conv_f9h_kernel): 3x3x3 and (1,3,3) filters over 16-bit x
        H9Geom g9; size_t l9; unsigned gr9;
        if (f9h_plan(g9, l9, gr9, B, D, H, W, Cin, Cout, kd, kh, kw, pd, ph, pw, epd, eph, epw, y_half != 0)) return 1;
    }
    if (Cin % 8 != 0 || Cout % 8 != 0 ||
        !half_geom(g, B, D, H, W, Cin, Cout, kd, kh, kw, pd, ph, pw, epd, eph, epw, x_half ? 2 : 4, y_half ? 2 : 4))
        return 0;
    const unsigned nwg = (unsigned)((long long)g.B * g.tilesD * g.tilesH * g.tilesW * g.nNt);
    return convh_persistent_takes(g, nwg) ? 1 : 0;
}

// rows of per-(tile, wave) column sums diqt_conv3d_fwd_h_io writes per batch entry when given `stats` (0: this shape emits none)
extern "C" int diqt_conv3d_fwd_h_stats_blocks(int B, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw, int pd, int ph,
                                              int pw, int epd, int eph, int epw, int x_half, int y_half) {
    HalfGeom g;
    if (x_half) {
        H9Geom g9; size_t l9; unsigned gr9;
        if (f9h_plan(g9, l9, gr9, B, D, H, W, Cin, Cout, kd, kh, kw, pd, ph, pw, epd, eph, epw, y_half != 0)) return f9h_stats_blocks(g9);
    }
    if (!half_geom(g, B, D, H, W, Cin, Cout, kd, kh, kw, pd, ph, pw, epd, eph, epw, x_half ? 2 : 4, y_half ? 2 : 4)) return 0;
    if ((kd == 1 && kh == 1 && kw == 1) || !x_half) return 0;          // flattened rows: tiles straddle batch entries
    const unsigned nwg = (unsigned)((long long)g.B * g.tilesD * g.tilesH * g.tilesW * g.nNt);
    return convh_persistent_takes(g, nwg) ? g.tilesD * g.tilesH * g.tilesW * (convh_four_waves(g, x_half != 0) ? 4 : 8) : 0;
}


// pointwise convs with fp32 rows at both ends and temporal convs with 16-bit input rows: the K-blocked GEMM (conv_pw_h_kernel)
static bool pwh_takes(const HalfGeom& g, bool xh, bool yh, const float* stats) {
    static const bool off = [] { const char* e = getenv("DIQT_NO_PWH"); return e && e[0] == '1'; }();
    static const bool notemporal = [] { const char* e = getenv("DIQT_NO_PWH_T"); return e && e[0] == '1'; }();
    if (off || yh || stats) return false;
    if (g.kh != 1 || g.kw != 1 || g.ph || g.pw || g.Do != g.D || g.Ho != g.H || g.Wo != g.W) return false;
    if (g.Cin % HCK != 0 || g.Cin < 64) return false;
    const long long rows = (long long)g.B * g.D * g.H * g.W;
    if (rows < 2048 || rows * g.Cin * (xh ? 2 : 4) >= (1ll << 31) || rows * g.Cout * 4 >= (1ll << 31)) return false;
    if (g.kd == 1) return !xh && g.pd == 0 && g.Cout >= 32;          // pointwise, fp32 rows (<= 64 channels: the 64-channel build, 5-20 % faster than the tap-oriented kernel)
    return !notemporal && xh && g.kd <= 4 && g.pd < g.kd && g.Cout >= 32;      // temporal: the 16-bit output of the per-frame conv
}
static int pwh_launch(const void* x, const unsigned short* wp, const float* bias, const float* residual, float* y, const HalfGeom& g, int bf16,
                      bool xh, hipStream_t s) {
    PwhGeom p;
    p.R = (long long)g.B * g.D * g.H * g.W;
    p.Cin = g.Cin; p.Cout = g.Cout; p.CoutPad = g.CoutPad; p.nChunks = g.Cin / HCK; p.roundOut = g.roundOut;
    p.T = g.kd; p.pd = g.pd; p.HW = g.kd == 1 ? 1 : g.H * g.W; p.D = g.kd == 1 ? 1 : g.D;
    p.xBytes = (unsigned)(p.R * g.Cin * (xh ? 2 : 4)); p.yBytes = (unsigned)(p.R * g.Cout * 4);
    p.wBytes = (unsigned)((size_t)p.nChunks * p.T * g.CoutPad * HCK * 2);
    // 128 channels per workgroup, TWO workgroups per CU (230 registers): one's loads and stores run under the other's MFMAs.  256 channels at
    // one workgroup per CU (accumulators in AGPRs) stage x half as often but measure slower on every shape of the cascade (121 vs 117 us
    // for 256 -> 512 on 131072 rows, 264 vs 208 us for 128 -> 128 on 1 M rows): a chunk step there lasts as long as its loads' latency
    const int nct = g.Cout > 64 ? 4 : 2;
    p.nRowTiles = (int)((p.R + 255) / 256);
    p.nCoTiles = (g.Cout + 32 * nct - 1) / (32 * nct);
    const unsigned grid = (unsigned)((p.nRowTiles + 7) / 8 * 8) * (unsigned)p.nCoTiles;
    const size_t lds = (size_t)2 * 256 * HROWB + (size_t)2 * 32 * nct * HROWB;
    typedef void (*KP)(const float*, const unsigned short*, const float*, const float*, float*, PwhGeom);
    const KP kp = xh ? (nct == 4 ? (bf16 ? conv_pw_h_kernel<true, 4, true> : conv_pw_h_kernel<false, 4, true>)
                                 : (bf16 ? conv_pw_h_kernel<true, 2, true> : conv_pw_h_kernel<false, 2, true>))
                     : (nct == 4 ? (bf16 ? conv_pw_h_kernel<true, 4, false> : conv_pw_h_kernel<false, 4, false>)
                                 : (bf16 ? conv_pw_h_kernel<true, 2, false> : conv_pw_h_kernel<false, 2, false>));
    hipLaunchKernelGGL(kp, dim3(grid), dim3(256), lds, s, static_cast<const float*>(x), wp, bias, residual, y, p);
    return check_launch("conv3d_fwd_h(gemm)");
}

static int convh_launch(const void* x, const void* packed_h, const float* bias, const float* residual, void* y, int B, int D, int H, int W,
                        int Cin, int Cout, int kd, int kh, int kw, int pd, int ph, int pw, int epd, int eph, int epw, int bf16,
                        int round_out, bool xh, bool yh, float* stats, void* stream) {
    DIQT_REQUIRE(x && packed_h && y, DIQT_E_ALIGN, "conv3d_fwd_h: null pointer");
    DIQT_REQUIRE(aligned16(x) && aligned16(packed_h), DIQT_E_ALIGN, "conv3d_fwd_h: x and the packed weights must be 16-byte aligned");
    if (xh && round_out && !(yh && residual)) {
        // 16-bit x, 3x3x3 / (1,3,3): the LDS-DMA kernel on conv_fwd9_kernel's structure (conv_f9h_kernel.h)
        H9Geom g9; size_t l9; unsigned gr9;
        if (f9h_plan(g9, l9, gr9, B, D, H, W, Cin, Cout, kd, kh, kw, pd, ph, pw, epd, eph, epw, yh)) {
            DIQT_REQUIRE(aligned16(y) && (!residual || aligned16(residual)), DIQT_E_ALIGN, "conv3d_fwd_h_io: y / residual must be 16-byte aligned");
            g9.stats = stats;
            return f9h_launch(x, static_cast<const unsigned short*>(packed_h), bias, residual, y, g9, l9, gr9, bf16, yh, stream);
        }
    }
    HalfGeom g;
    DIQT_REQUIRE(half_geom(g, B, D, H, W, Cin, Cout, kd, kh, kw, pd, ph, pw, epd, eph, epw, xh ? 2 : 4, yh ? 2 : 4), DIQT_E_UNSUPPORTED,
                 "conv3d_fwd_h: shape not supported by the low-precision kernel (diqt_conv3d_fwd_h_supported == 0)");
    g.roundOut = round_out ? 1 : 0;
    if (pwh_takes(g, xh, yh, stats))
        return pwh_launch(x, static_cast<const unsigned short*>(packed_h), bias, residual, static_cast<float*>(y), g, bf16, xh, (hipStream_t)stream);
    const unsigned nwg = (unsigned)((long long)g.B * g.tilesD * g.tilesH * g.tilesW * g.nNt);
    static const bool dbg_on = [] { const char* e = getenv("DIQT_CONVH_DBG"); return e && e[0] == '1'; }();
    if (dbg_on && nwg <= 65536 && !xh && !yh) {
        if (!g_hdbg) DIQT_REQUIRE(hipMalloc(&g_hdbg, (size_t)65536 * 8 * sizeof(unsigned long long)) == hipSuccess, DIQT_E_LAUNCH, "conv3d_fwd_h: debug buffer");
        g.dbg = g_hdbg; g_hdbg_n = nwg;
    }
    size_t lds = half_lds_bytes(g);
    const int HV = g.HD * g.HH * g.HWd;
    static const bool nopref = [] { const char* e = getenv("DIQT_CONVH_NOPREF"); return e && e[0] == '1'; }();
    const bool pref = !nopref && HV * 8 <= 512 * HHREG;
    hipStream_t s = (hipStream_t)stream;
    if (!g.dbg && convh_persistent_takes(g, nwg)) {
        // (two workgroups per CU -- 6 halo pieces per thread, weight groups of <= 5 taps, 128 registers per wave -- measured 1.4x SLOWER
        // on the 64^3 level-0 shapes: the kernel moves 3.5 TB/s of fp32 activations, 0.75 of what a plain copy reaches)
        if (xh || yh) {
            DIQT_REQUIRE(Cin % 8 == 0 && Cout % 8 == 0 && (!yh || (round_out && !residual)), DIQT_E_UNSUPPORTED,
                         "conv3d_fwd_h_io: 16-bit tensors need Cin, Cout %% 8 == 0; a 16-bit output needs round_out and no residual");
        }
        DIQT_REQUIRE(!stats || !(kd == 1 && kh == 1 && kw == 1), DIQT_E_UNSUPPORTED, "conv3d_fwd_h_io: no statistics from a 1x1x1 conv");
        static const bool nowide = [] { const char* e = getenv("DIQT_CONVH_NOWIDE"); return e && e[0] == '1'; }();
        // (fp32 rows of a pointwise conv: the 4-wave build below at one chunk per step measures slightly faster than two chunks per step on 8 waves)
        static const bool pw4first = [] { const char* e = getenv("DIQT_CONVH_PW4"); return !(e && e[0] == '0'); }();
        if (kd * kh * kw == 1 && g.nChunks >= 2 && !nowide && !(pw4first && !xh)) {
            // pointwise: 4 MFMAs of a wave per 32-channel chunk and two barriers around them -- stage TWO chunks per step and walk them like
            // taps (their weight panels are consecutive in the packed layout [chunk][tap = 1][co][32])
            g.NS = 2; g.realChunks = g.nChunks; g.nChunks = (g.realChunks + 1) / 2; g.TG = 2; g.nGroups = 1;
        }
        lds = half_lds_bytes(g);
        typedef void (*KP)(const float*, const unsigned short*, const float*, const float*, float*, HalfGeom, int, int, float*);
        const int sel = (bf16 ? 4 : 0) + (xh ? 2 : 0) + (yh ? 1 : 0);
        static const KP tab[8] = {conv_fwd_hp_kernel<false, 1, HHREG, HTG, false, false>, conv_fwd_hp_kernel<false, 1, HHREG, HTG, false, true>,
                                  conv_fwd_hp_kernel<false, 1, HHREG / 2, HTG, true, false>,  conv_fwd_hp_kernel<false, 1, HHREG / 2, HTG, true, true>,
                                  conv_fwd_hp_kernel<true, 1, HHREG, HTG, false, false>,  conv_fwd_hp_kernel<true, 1, HHREG, HTG, false, true>,
                                  conv_fwd_hp_kernel<true, 1, HHREG / 2, HTG, true, false>,   conv_fwd_hp_kernel<true, 1, HHREG / 2, HTG, true, true>};
        DIQT_REQUIRE(!stats || xh, DIQT_E_UNSUPPORTED, "conv3d_fwd_h_io: statistics are built for x_half = 1");
        const bool four = convh_four_waves(g, xh);
        KP kp;
        if (four && !xh) {
            DIQT_REQUIRE(!stats, DIQT_E_UNSUPPORTED, "conv3d_fwd_h_io: statistics are built for x_half = 1");
            lds = half_lds_bytes(g, 1);
            kp = yh ? (bf16 ? conv_fwd_hp_kernel<true, 1, 8, HTG, false, true, false, 4> : conv_fwd_hp_kernel<false, 1, 8, HTG, false, true, false, 4>)
                    : (bf16 ? conv_fwd_hp_kernel<true, 1, 8, HTG, false, false, false, 4> : conv_fwd_hp_kernel<false, 1, 8, HTG, false, false, false, 4>);
        } else if (four) {
            lds = half_lds_bytes(g, 1);
            kp = stats ? (yh ? (bf16 ? conv_fwd_hp_kernel<true, 1, 6, HTG, true, true, true, 4> : conv_fwd_hp_kernel<false, 1, 6, HTG, true, true, true, 4>)
                             : (bf16 ? conv_fwd_hp_kernel<true, 1, 6, HTG, true, false, true, 4> : conv_fwd_hp_kernel<false, 1, 6, HTG, true, false, true, 4>))
                       : (yh ? (bf16 ? conv_fwd_hp_kernel<true, 1, 6, HTG, true, true, false, 4> : conv_fwd_hp_kernel<false, 1, 6, HTG, true, true, false, 4>)
                             : (bf16 ? conv_fwd_hp_kernel<true, 1, 6, HTG, true, false, false, 4> : conv_fwd_hp_kernel<false, 1, 6, HTG, true, false, false, 4>));
        } else {
            kp = !stats ? tab[sel]
               : yh ? (bf16 ? conv_fwd_hp_kernel<true, 1, HHREG / 2, HTG, true, true, true> : conv_fwd_hp_kernel<false, 1, HHREG / 2, HTG, true, true, true>)
                    : (bf16 ? conv_fwd_hp_kernel<true, 1, HHREG / 2, HTG, true, false, true> : conv_fwd_hp_kernel<false, 1, HHREG / 2, HTG, true, false, true>);
        }
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kp), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            DIQT_REQUIRE(e == hipSuccess, DIQT_E_LAUNCH, "conv3d_fwd_h: hipFuncSetAttribute: %s", hipGetErrorString(e));
        }
        const int wgs = diqt_set_convh_workgroups(0) * (four ? 2 : 1);
        const int perWg = (int)((nwg + wgs - 1) / wgs), grid = (int)((nwg + perWg - 1) / perWg);
        hipLaunchKernelGGL(kp, dim3(grid), dim3(four ? 256 : 512), lds, s, static_cast<const float*>(x), static_cast<const unsigned short*>(packed_h), bias,
                           residual, static_cast<float*>(y), g, (int)nwg, perWg, stats);
        return check_launch("conv3d_fwd_h(persistent)");
    }
    DIQT_REQUIRE(!xh && !yh && !stats, DIQT_E_UNSUPPORTED, "conv3d_fwd_h_io: 16-bit tensors only on the persistent kernel (diqt_conv3d_fwd_h_io16_supported)");
    void (*kern)(const float*, const unsigned short*, const float*, const float*, float*, HalfGeom) =
        bf16 ? (pref ? conv_fwd_h_kernel<true, true> : conv_fwd_h_kernel<true, false>)
             : (pref ? conv_fwd_h_kernel<false, true> : conv_fwd_h_kernel<false, false>);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        DIQT_REQUIRE(e == hipSuccess, DIQT_E_LAUNCH, "conv3d_fwd_h: hipFuncSetAttribute: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(512), lds, s, static_cast<const float*>(x), static_cast<const unsigned short*>(packed_h), bias,
                       residual, static_cast<float*>(y), g);
    return check_launch("conv3d_fwd_h");
}

extern "C" int diqt_conv3d_fwd_h(const float* x, const void* packed_h, const float* bias, const float* residual, float* y, int B, int D,
                                 int H, int W, int Cin, int Cout, int kd, int kh, int kw, int pd, int ph, int pw, int epd, int eph,
                                 int epw, int bf16, int round_out, void* stream) {
    return convh_launch(x, packed_h, bias, residual, y, B, D, H, W, Cin, Cout, kd, kh, kw, pd, ph, pw, epd, eph, epw, bf16, round_out, false,
                        false, nullptr, stream);
}

extern "C" int diqt_conv3d_fwd_h_io(const void* x, const void* packed_h, const float* bias, const float* residual, void* y, int B, int D,
                                    int H, int W, int Cin, int Cout, int kd, int kh, int kw, int pd, int ph, int pw, int epd, int eph,
                                    int epw, int bf16, int round_out, int x_half, int y_half, float* stats, void* stream) {
    return convh_launch(x, packed_h, bias, residual, y, B, D, H, W, Cin, Cout, kd, kh, kw, pd, ph, pw, epd, eph, epw, bf16, round_out,
                        x_half != 0, y_half != 0, stats, stream);
}
