// Weight gradient of the stride-1 3x3x3 / 1x3x3 / 3x1x1 convolutions with 16-bit MFMA operands (bf16 training, `ImagenTrainer(precision='bf16')`:
// the reference's autocast backward computes the weight gradient in the forward's type, trainer.py:293-311):
//   dW[co][ci][tap] = sum_v dY[v][co] * X[v + tap][ci]      M = co, N = ci, K = voxels, on v_mfma_f32_32x32x16_{bf16,f16}, fp32 accumulate
//
// Both operands are summed over VOXELS, the slow axis of the channels-last tensors, so neither is k-contiguous in memory.  The fp32 kernel
// (conv_wgrad.hip) reads them as scalars per lane (v_mfma_f32_32x32x2_f32 takes one k per lane); for the 8-wide k of the 16-bit MFMA the
// LDS images stay ROW-MAJOR [voxel][channel] -- exactly as staged from HBM, converted to 16 bit on the way -- and the fragments come out
// of `ds_read_b64_tr_b16`: a 16-lane group reads 4 voxels x 16 channels and hands lane i channel i of the 4 voxels.  A filter tap is then a
// whole-row offset in the halo image (no alignment constraints on the shifted reads).
//   * 256 threads, one wave per SIMD; a workgroup owns 64 co x 32 ci x all taps and walks a split-K range of 128-voxel tiles (2 x 4 x 16);
//     wave w = (co half, tap parity): 14 / 13 accumulator tiles for 3x3x3 (224 AGPRs), 5 / 4 for 1x3x3, 2 / 1 for 3x1x1;
//   * the next tile's dY rows and X halo rows are loaded global -> registers (range-checked buffer loads: zero padding, ragged tiles)
//     before the current tile's MFMAs and converted + written to LDS behind them (two barriers per tile);
//   * LDS rows: X 64 B (32 ci), dY 192 B (64 co + pad): the 4 rows x 2 channel blocks of a 32-lane half fall on disjoint bank groups;
//   * tried and dropped (round 4, in-situ timing of the bf16 training step on one box): 256-voxel tiles (4 x 4 x 16: 59.7 vs 60 us) and a
//     wave split by tap residue mod 4 with both co halves per wave, so that a B fragment read from the LDS feeds two MFMAs (61-64 us):
//     the kernel is not bound by its LDS fragment reads but by what sits around the MFMA loop (staging of the next tile between two
//     barriers, 221 KB of slab per workgroup through the LDS at the end of a 16-tile walk);
//   * partial slabs [slice][Cout][Cin][taps] and bias partials [slice][CoutPad] as conv_wgrad3_kernel writes them: the same fixed-order
//     reduce (conv_reduce_dw3_kernel) finishes the gradient -- deterministic.
#include "common.h"
#include "conv_wgrad_h.h"
#include <stdlib.h>

namespace diqt {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4w __attribute__((ext_vector_type(4)));
typedef unsigned u32x2w __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8w __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8w __attribute__((ext_vector_type(8)));
typedef short s16x4w __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4w lds_s16x4w;

constexpr unsigned WH_OOB = 0x80000000u;
constexpr int WTD = 2, WTH = 4, WTW = 16, WMV = WTD * WTH * WTW;        // 128-voxel tile: 8 k-blocks of 16 voxels (one row of 16 along W)
constexpr int XROW = 64, YROW = 192;                                    // LDS row bytes
constexpr int NPY = WMV * 16 / 256;                                     // dY float4 pieces per thread: 8
template <int KD, int KH, int KW>
struct WHCfg {                                                          // 3x3x3: 4 x 6 x 18 = 432 halo voxels, 14 X pieces per thread
    static constexpr int KD_ = KD, KH_ = KH, KW_ = KW;
    static constexpr int T = KD * KH * KW, NA = (T + 1) / 2;            // taps, accumulator tiles of a wave (tap parity halves)
    static constexpr int HD = WTD + KD - 1, HH = WTH + KH - 1, HW = WTW + KW - 1, HV = HD * HH * HW;
    static constexpr int NPX = (HV * 8 + 255) / 256;
    static constexpr int LDS_TILE = HV * XROW + WMV * YROW, LDS_STAGE = 16 * 32 * T * 4;
    static constexpr int LDS = LDS_TILE > LDS_STAGE ? LDS_TILE : LDS_STAGE;
    static constexpr int PD = NA < 5 ? NA : 5;                          // B fragments in flight ahead of their MFMA
};

// (by value: __builtin_bit_cast on a vector COMPONENT lvalue reads element 0 for every component on this hipcc, see conv_half.hip)
__device__ __forceinline__ float wasf(unsigned u) { return __builtin_bit_cast(float, u); }

template <bool BF>
__device__ __forceinline__ unsigned wpack2(float a, float b) {
    if (BF) {
        typedef __bf16 b2 __attribute__((ext_vector_type(2)));
        b2 v = {(__bf16)a, (__bf16)b};
        return __builtin_bit_cast(unsigned, v);
    } else {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        h2 v = {(_Float16)a, (_Float16)b};
        return __builtin_bit_cast(unsigned, v);
    }
}
template <bool BF>
__device__ __forceinline__ f32x16 wmfma(u32x4w a, u32x4w b, f32x16 c) {
    if (BF) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8w, a), __builtin_bit_cast(bf16x8w, b), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8w, a), __builtin_bit_cast(f16x8w, b), c, 0, 0, 0);
}
// 8 k-values (voxels r0 .. r0 + 7 of the image, this lane's channel) as one MFMA operand: two transposing reads of 4 rows each
__device__ __forceinline__ u32x4w tr_frag(const unsigned char* p, int rowBytes) {
    const s16x4w lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4w*)p);
    const s16x4w hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4w*)(p + 4 * rowBytes));
    const unsigned long long l = __builtin_bit_cast(unsigned long long, lo), h = __builtin_bit_cast(unsigned long long, hi);
    u32x4w r;
    r.x = (unsigned)l; r.y = (unsigned)(l >> 32); r.z = (unsigned)h; r.w = (unsigned)(h >> 32);
    return r;
}

// XH: x holds 16-bit values of the operand type (the GroupNorm-apply output a bf16 training step saved in that type, ops._GnActConvHFn):
// a piece is 8 channels and goes to the LDS image as it is -- the same bits the fp32 tensor's values round to while staged.
// DYH: dY holds 16-bit values as well (the gradient a low-precision training step keeps in the operand type between a GroupNorm backward
// and the conv backward that consumes it): 8 channels per piece, copied verbatim; the bias gradient sums exactly those values.
template <class C, bool BF, bool XH = false, bool DYH = false>
__global__ __launch_bounds__(256, 1) void conv_wgrad_h_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                              float* __restrict__ slabs, float* __restrict__ bias_part, WHGeom g) {
    constexpr int WT = C::T, NA = C::NA, WHH = C::HH, WHW = C::HW, WHV = C::HV, NPX = XH ? (C::HV * 4 + 255) / 256 : C::NPX;
    constexpr int XPS = XH ? 2 : 3, XCH = XH ? 8 : 4;      // log2 pieces per 32-channel row, channels per 16-byte piece
    constexpr unsigned XE = XH ? 2u : 4u;
    constexpr int NPYK = DYH ? WMV * 8 / 256 : NPY, YPS = DYH ? 3 : 4, YCH = DYH ? 8 : 4;      // dY pieces per thread, log2 pieces per 64-channel row, channels per piece
    constexpr unsigned YE = DYH ? 2u : 4u;
    extern __shared__ __attribute__((aligned(16))) unsigned char smw[];
    unsigned char* Xs = smw;                         // [432][64 B]   halo voxels x 32 ci
    unsigned char* Ys = smw + WHV * XROW;            // [128][192 B]  tile voxels x 64 co
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hf = lane >> 5;
    const int ct = wave & 1, tpar = wave >> 1;       // co half, tap parity: taps tpar, tpar + 2, ...
    const int NTW = tpar ? WT / 2 : (WT + 1) / 2;    // 3x3x3: 13 / 14 taps
    int bx = blockIdx.x;
    const int cob = bx % g.nCoB; bx /= g.nCoB;
    const int cib = bx;
    const int co0 = cob * 64, ci0 = cib * 32;
    const int mtBegin = blockIdx.y * g.tilesPerSplit, mtEnd = min(mtBegin + g.tilesPerSplit, g.MT);

    const auto rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, (int)g.xBytes, 0x00020000);
    const auto rs_y = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dy), 0, (int)g.yBytes, 0x00020000);

    // ---- this thread's staging pieces (tile-independent part): halo coordinates as guarded 10-bit fields (range tests of a tile are two
    // subtractions, see conv_fwd_hp_kernel) and the element offset inside the tile's halo ----
    constexpr unsigned GUARD = (1u << 9) | (1u << 19) | (1u << 29), PADB = 16u, LOW = PADB | (PADB << 10) | (PADB << 20);
    unsigned xpc[NPX], xrel[NPX], xdst[NPX];
#pragma unroll
    for (int u = 0; u < NPX; ++u) {
        const int idx = u * 256 + tid, row = idx >> XPS, q4 = idx & ((1 << XPS) - 1);
        const bool ok = row < WHV && ci0 + q4 * XCH < g.Cin;
        const int hx = row % WHW, hy = (row / WHW) % WHH, hz = row / (WHW * WHH);
        xpc[u] = ok ? ((unsigned)hz | ((unsigned)hy << 10) | ((unsigned)hx << 20)) : 0x3fffffffu;      // no piece: never in range
        xrel[u] = (unsigned)(((hz * g.H + hy) * g.W + hx) * g.Cin + ci0 + q4 * XCH) * XE;
        xdst[u] = row < WHV ? (unsigned)(row * XROW + q4 * (XH ? 16 : 8)) : 0xffffffffu;
    }
    const unsigned limits = (unsigned)(g.D + PADB - 1) | ((unsigned)(g.H + PADB - 1) << 10) | ((unsigned)(g.W + PADB - 1) << 20);
    unsigned yrel[NPYK], ypc[NPYK];
    const int yq = tid & ((1 << YPS) - 1);            // this thread's co quad (octet) of a dY row (the same for all its pieces)
#pragma unroll
    for (int u = 0; u < NPYK; ++u) {
        const int row = (u * 256 + tid) >> YPS;
        const int tw = row % WTW, th = (row / WTW) % WTH, td = row / (WTW * WTH);
        ypc[u] = (unsigned)td | ((unsigned)th << 10) | ((unsigned)tw << 20);
        yrel[u] = (unsigned)(((td * g.Ho + th) * g.Wo + tw) * g.Cout + co0 + yq * YCH) * YE;
    }
    const unsigned ylimits = (unsigned)(g.Do + PADB - 1) | ((unsigned)(g.Ho + PADB - 1) << 10) | ((unsigned)(g.Wo + PADB - 1) << 20);
    const bool yok = co0 + yq * YCH < g.Cout;
    // ---- fragment addresses ----
    const int trq = (lane & 15) >> 2, trp = lane & 3, trc = (lane >> 4) & 1;
    const unsigned char* aBase = Ys + (8 * hf + trq) * YROW + (32 * ct + 16 * trc + 4 * trp) * 2;
    const unsigned char* bBase = Xs + (8 * hf + trq) * XROW + (16 * trc + 4 * trp) * 2;
    int tapOff[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int t = min(2 * i + tpar, WT - 1);
        tapOff[i] = (((t / (C::KH_ * C::KW_)) * WHH + (t / C::KW_) % C::KH_) * WHW + t % C::KW_) * XROW;
    }

    f32x16 acc[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f), bsum2 = make_float4(0.f, 0.f, 0.f, 0.f);      // (bsum2: channels 4..7 of a 16-bit octet)

    u32x4w px[NPX];                                   // raw pieces: 4 fp32 channels, or (XH) 8 channels of the operand type
    u32x4w py[NPYK];
    auto load_tile = [&](int mt) {
        int m = mt;
        const int tx = m % g.tilesW; m /= g.tilesW;
        const int ty = m % g.tilesH; m /= g.tilesH;
        const int tz = m % g.tilesD;
        const int b = m / g.tilesD;
        const int d0 = tz * WTD, h0 = ty * WTH, w0 = tx * WTW;
        const unsigned xorg = (unsigned)(d0 - g.pd + (int)PADB) + ((unsigned)(h0 - g.ph + (int)PADB) << 10) + ((unsigned)(w0 - g.pw + (int)PADB) << 20);
        const unsigned xbase = (unsigned)((((b * g.D + d0 - g.pd) * g.H + h0 - g.ph) * g.W + w0 - g.pw) * g.Cin) * XE;
#pragma unroll
        for (int u = 0; u < NPX; ++u) {
            const unsigned c = xpc[u] + xorg;                     // fields < 512: extents <= 255 (plan), PADB, a halo of 2
            const unsigned okm = ((c | GUARD) - LOW) & ((limits | GUARD) - c) & GUARD;
            const unsigned off = (okm == GUARD && xpc[u] != 0x3fffffffu) ? xbase + xrel[u] : WH_OOB;
            px[u] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, off, 0, 0);
        }
        const unsigned yorg = (unsigned)(d0 + (int)PADB) + ((unsigned)(h0 + (int)PADB) << 10) + ((unsigned)(w0 + (int)PADB) << 20);
        const unsigned ybase = (unsigned)((((b * g.Do + d0) * g.Ho + h0) * g.Wo + w0) * g.Cout) * YE;
#pragma unroll
        for (int u = 0; u < NPYK; ++u) {
            const unsigned c = ypc[u] + yorg;
            const unsigned okm = ((c | GUARD) - LOW) & ((ylimits | GUARD) - c) & GUARD;
            const unsigned off = (yok && okm == GUARD) ? ybase + yrel[u] : WH_OOB;
            py[u] = __builtin_amdgcn_raw_buffer_load_b128(rs_y, off, 0, 0);
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int u = 0; u < NPX; ++u) {
            if (xdst[u] != 0xffffffffu) {
                if constexpr (XH) {
                    *reinterpret_cast<u32x4w*>(Xs + xdst[u]) = px[u];
                } else {
                    const u32x4w v = px[u];
                    u32x2w w;
                    w.x = wpack2<BF>(wasf(v.x), wasf(v.y)); w.y = wpack2<BF>(wasf(v.z), wasf(v.w));
                    *reinterpret_cast<u32x2w*>(Xs + xdst[u]) = w;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < NPYK; ++u) {
            const int row = (u * 256 + tid) >> YPS;
            const u32x4w v = py[u];
            if constexpr (DYH) {
                *reinterpret_cast<u32x4w*>(Ys + row * YROW + yq * 16) = v;
                float f[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const unsigned wd = e == 0 ? v.x : e == 1 ? v.y : e == 2 ? v.z : v.w;
                    if (BF) { f[2 * e] = wasf(wd << 16); f[2 * e + 1] = wasf(wd & 0xffff0000u); }
                    else {
                        typedef _Float16 h2w __attribute__((ext_vector_type(2)));
                        const h2w hv = __builtin_bit_cast(h2w, wd);
                        f[2 * e] = (float)hv[0]; f[2 * e + 1] = (float)hv[1];
                    }
                }
                bsum.x += f[0]; bsum.y += f[1]; bsum.z += f[2]; bsum.w += f[3];
                bsum2.x += f[4]; bsum2.y += f[5]; bsum2.z += f[6]; bsum2.w += f[7];
            } else {
                const float4 pf = make_float4(wasf(v.x), wasf(v.y), wasf(v.z), wasf(v.w));
                u32x2w w;
                w.x = wpack2<BF>(pf.x, pf.y); w.y = wpack2<BF>(pf.z, pf.w);
                *reinterpret_cast<u32x2w*>(Ys + row * YROW + yq * 8) = w;
                bsum.x += pf.x; bsum.y += pf.y; bsum.z += pf.z; bsum.w += pf.w;      // exact fp32 dY: the bias gradient
            }
        }
    };

    if (mtBegin < mtEnd) load_tile(mtBegin);
    for (int mt = mtBegin; mt < mtEnd; ++mt) {
        store_tile();
        __syncthreads();
        if (mt + 1 < mtEnd) load_tile(mt + 1);              // in flight behind this tile's MFMAs
        // fully unrolled (was `#pragma unroll 1`): the scheduler then requests a k-block's first fragments under the previous k-block's last
        // MFMAs -- 59.8 -> 54.5 us on 64 -> 64 @ 8 x 32^3 inside the bf16 training step; an explicit flat software pipeline over all
        // (k-block, tap) steps, and 7 instead of 5 fragments in flight, measured the same
#pragma unroll
        for (int kb = 0; kb < WMV / 16; ++kb) {              // 16 voxels: row (d, h) = (kb / 4, kb % 4) of the tile, all 16 w
            const u32x4w a = tr_frag(aBase + kb * 16 * YROW, YROW);
            const unsigned char* bp = bBase + (((kb >> 2) * WHH + (kb & 3)) * WHW) * XROW;
            // the B fragments run PD taps ahead of their MFMA: an LDS read is ~150 cycles away, an MFMA 32 -- one tap ahead (what the
            // compiler makes of the plain loop) left every MFMA waiting for its own operands (57 % of the wave's cycles in s_waitcnt)
            constexpr int PD = C::PD;
            u32x4w bq[PD];
#pragma unroll
            for (int i = 0; i < PD; ++i) bq[i] = tr_frag(bp + tapOff[i], XROW);
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                if (i < NTW) {                                   // wave-uniform (3x3x3: 13 or 14 taps)
                    const u32x4w b = bq[i % PD];
                    if (i + PD < NA && i + PD < NTW) bq[i % PD] = tr_frag(bp + tapOff[(i + PD) % NA], XROW);
                    acc[i] = wmfma<BF>(a, b, acc[i]);
                }
            }
        }
        __syncthreads();                                      // every wave is done with the images
    }

    // ---- slab [slice][Cout][Cin][taps]: this wave's co half x 32 ci x its taps ----
    // Through LDS in four rounds of 16 output channels, so that the slab leaves in its [co][ci][tap] order as 16-byte stores of contiguous
    // runs (a (co, this ci block) pair is 32 x 27 contiguous floats).  Straight from the accumulators every lane would write 216 single
    // floats 108 bytes apart: 14 M partial 32-byte sectors per launch -- that alone was about half of the kernel's time.
    float* slab = slabs + (size_t)blockIdx.y * g.Cout * g.Cin * WT;
    float* stage = reinterpret_cast<float*>(smw);           // [16 co][32 ci][27 taps]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (ct == (r >> 1)) {                               // wave-uniform: the two waves (tap parities) that hold these channels
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                if (i < NTW) {
                    const int tap = 2 * i + tpar;
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) {
                        const int j = 8 * (r & 1) + jj;                           // accumulator rows 16 (r & 1) .. + 15 of this wave's 32
                        const int cl = (jj & 3) + 8 * (jj >> 2) + 4 * hf;          // channel inside the group of 16
                        stage[(cl * 32 + l31) * WT + tap] = acc[i][j];
                    }
                }
            }
        }
        __syncthreads();
        for (int f = tid; f < 16 * 32 * WT / 4; f += 256) {
            const int cl = f / (32 * WT / 4), k4 = f % (32 * WT / 4);
            const int co = co0 + 16 * r + cl;
            if (co < g.Cout)
                *reinterpret_cast<float4*>(slab + ((size_t)co * g.Cin + ci0) * WT + 4 * k4) = *reinterpret_cast<const float4*>(stage + (cl * 32 * WT) + 4 * k4);
        }
        __syncthreads();
    }
    // ---- bias partial of this slice (workgroups of the first ci block): 16 threads share a co quad ----
    if (bias_part && cib == 0) {
        float4* red = reinterpret_cast<float4*>(smw);
        if constexpr (DYH) {            // 32 threads share a channel octet: [256][2] quads
            red[2 * tid] = bsum; red[2 * tid + 1] = bsum2;
            __syncthreads();
            if (tid < 16) {             // quad tid of the 64 channels: octet tid / 2, half tid % 2
                float4 s = red[2 * (tid >> 1) + (tid & 1)];
                for (int k = 1; k < 32; ++k) { const float4 t = red[2 * ((tid >> 1) + 8 * k) + (tid & 1)]; s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w; }
                float* bp = bias_part + (size_t)blockIdx.y * g.CoutPad + co0 + tid * 4;
                if (co0 + tid * 4 < g.Cout) { bp[0] = s.x; bp[1] = s.y; bp[2] = s.z; bp[3] = s.w; }
            }
        } else {
            red[tid] = bsum;
            __syncthreads();
            if (tid < 16) {
                float4 s = red[tid];
                for (int k = 1; k < 16; ++k) { const float4 t = red[tid + 16 * k]; s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w; }
                float* bp = bias_part + (size_t)blockIdx.y * g.CoutPad + co0 + tid * 4;
                if (co0 + tid * 4 < g.Cout) { bp[0] = s.x; bp[1] = s.y; bp[2] = s.z; bp[3] = s.w; }
            }
        }
    }
}

}  // namespace

bool wgradh_plan(WHGeom& g, int& ksplit, int B, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw, int pd, int ph, int pw,
                 int epd, int eph, int epw, bool xHalf, bool dyHalf) {
    static const bool off = [] { const char* e = getenv("DIQT_NO_WGRADH"); return e && e[0] == '1'; }();
    const bool filt = (kd == 3 && kh == 3 && kw == 3) || (kd == 1 && kh == 3 && kw == 3) || (kd == 3 && kh == 1 && kw == 1);
    if (off || !filt || Cin % 32 != 0 || Cout % 4 != 0 || Cin < 32 || Cout < 32) return false;
    if (B <= 0 || D <= 0 || H <= 0 || W <= 0 || pd < 0 || ph < 0 || pw < 0 || pd > 16 || ph > 16 || pw > 16) return false;
    if (D > 255 || H > 255 || W > 255) return false;                   // packed 10-bit coordinate fields in the kernel
    g.B = B; g.D = D; g.H = H; g.W = W; g.Cin = Cin; g.Cout = Cout; g.pd = pd; g.ph = ph; g.pw = pw;
    g.kd = kd; g.kh = kh; g.kw = kw;
    g.Do = D + 2 * pd + epd - kd + 1; g.Ho = H + 2 * ph + eph - kh + 1; g.Wo = W + 2 * pw + epw - kw + 1;
    if (g.Do <= 0 || g.Ho <= 0 || g.Wo <= 0) return false;
    g.tilesD = (g.Do + WTD - 1) / WTD; g.tilesH = (g.Ho + WTH - 1) / WTH; g.tilesW = (g.Wo + WTW - 1) / WTW;
    const long long mt = (long long)B * g.tilesD * g.tilesH * g.tilesW;
    if (mt >= (1ll << 30)) return false;
    g.MT = (int)mt;
    g.nCoB = (Cout + 63) / 64; g.nCiB = Cin / 32; g.CoutPad = g.nCoB * 64;
    const unsigned long long xb = (unsigned long long)B * D * H * W * Cin * (xHalf ? 2ull : 4ull), yb = (unsigned long long)B * g.Do * g.Ho * g.Wo * Cout * (dyHalf ? 2ull : 4ull);
    if (xb >= (1ull << 30) || yb >= (1ull << 30) || (xHalf && Cin % 8 != 0) || (dyHalf && Cout % 8 != 0)) return false;
    g.xBytes = (unsigned)xb; g.yBytes = (unsigned)yb; g.xHalf = xHalf ? 1 : 0; g.dyHalf = dyHalf ? 1 : 0;
    const int blocks = g.nCoB * g.nCiB;
    ksplit = 256 / blocks;
    if (ksplit > g.MT) ksplit = g.MT;
    if (ksplit < 1) ksplit = 1;
    g.tilesPerSplit = (g.MT + ksplit - 1) / ksplit;
    ksplit = (g.MT + g.tilesPerSplit - 1) / g.tilesPerSplit;
    return true;
}

template <class C>
static int wgradh_launch_t(const float* x, const float* dy, float* slabs, float* bias_part, const WHGeom& g, int ksplit, int bf16, void* stream) {
    auto kern = g.xHalf && g.dyHalf ? (bf16 ? conv_wgrad_h_kernel<C, true, true, true> : conv_wgrad_h_kernel<C, false, true, true>)
              : g.xHalf ? (bf16 ? conv_wgrad_h_kernel<C, true, true> : conv_wgrad_h_kernel<C, false, true>)
                        : (bf16 ? conv_wgrad_h_kernel<C, true> : conv_wgrad_h_kernel<C, false>);
    DIQT_REQUIRE(!g.dyHalf || g.xHalf, DIQT_E_UNSUPPORTED, "conv3d_bwd_weight_h: a 16-bit dY is built together with a 16-bit x");
    if (C::LDS > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
        DIQT_REQUIRE(e == hipSuccess, DIQT_E_LAUNCH, "conv3d_bwd_weight_h: hipFuncSetAttribute: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(kern, dim3(g.nCoB * g.nCiB, ksplit), dim3(256), C::LDS, (hipStream_t)stream, x, dy, slabs, bias_part, g);
    return check_launch("conv3d_bwd_weight_h");
}

int wgradh_launch(const float* x, const float* dy, float* slabs, float* bias_part, const WHGeom& g, int ksplit, int bf16, void* stream) {
    if (g.kd == 3 && g.kh == 3) return wgradh_launch_t<WHCfg<3, 3, 3>>(x, dy, slabs, bias_part, g, ksplit, bf16, stream);
    if (g.kd == 1) return wgradh_launch_t<WHCfg<1, 3, 3>>(x, dy, slabs, bias_part, g, ksplit, bf16, stream);
    return wgradh_launch_t<WHCfg<3, 1, 1>>(x, dy, slabs, bias_part, g, ksplit, bf16, stream);
}

}  // namespace diqt
