// Stride-1 convolution with a HANDFUL of output channels (the U-Nets' final convs: dim -> 1 image channel, imagen_video.py:1560
// `final_conv`, imagen_pytorch3D.py:1485; the 1x1x1 `to_k` of GlobalContext, imagen_video.py:585-601, dim -> 1).
//
// On the MFMA kernels such a conv pads Cout to a 32- or 64-wide tile: 1 useful column in 32 (the 65 -> 1 final conv of the 64^3
// stage took 1.9 ms for a tensor that streams in 0.12 ms).  It is a memory-bound reduction, done here on the vector ALU in two steps
// per tile of 256 output voxels:
//   1. per HALO voxel the responses of all taps  z[v][t, o] = sum_c x[v][c] w[o][c][t]   (x read ONCE, coalesced, staged through LDS
//      in channel chunks; one thread per halo voxel walks its row -- row stride odd, so the 64 lanes hit 64 banks);
//   2. per output voxel  y[v][o] = bias[o] + sum_t z[v + t][t, o]  (+ residual)           (a T-point gather in LDS).
// Exact fp32 FMAs; any Cin (rows need not be 16-byte aligned), T * Cout <= 32.
#include "common.h"
#include <stdlib.h>

namespace diqt {
namespace {

struct SCGeom {
    int B, D, H, W, Cin, Do, Ho, Wo, pd, ph, pw;
    int TD, TH, TW, HD, HH, HWd, HV;
    int tilesD, tilesH, tilesW;
    int cch, nChunks, rowS;          // channels per staged chunk, chunks, LDS row stride (floats, odd)
    unsigned xBytes, yBytes;
};

constexpr unsigned SC_OOB = 0x80000000u;

template <int KD, int KH, int KW, int CO>
__global__ __launch_bounds__(256) void conv_smallcout_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ bias, const float* __restrict__ residual,
                                                             float* __restrict__ y, SCGeom g) {
    constexpr int T = KD * KH * KW, TC = T * CO, TCP = TC | 1;      // odd row stride of the z image
    constexpr int NV = 3;                                          // halo voxels per thread (HV <= 768)
    extern __shared__ __attribute__((aligned(16))) float sm_sc[];
    float* wS = sm_sc;                                   // [Cin][TC]
    float* zS = wS + (size_t)g.Cin * TC;                 // [HV][TCP]
    float* xS = zS + (size_t)g.HV * TCP;                 // [HV][rowS]
    const int tid = threadIdx.x;

    const unsigned L = xcd_remap(blockIdx.x, gridDim.x);
    int mt = (int)L;
    const int tx = mt % g.tilesW; mt /= g.tilesW;
    const int ty = mt % g.tilesH; mt /= g.tilesH;
    const int tz = mt % g.tilesD;
    const int b = mt / g.tilesD;
    const int d0 = tz * g.TD, h0 = ty * g.TH, w0 = tx * g.TW;

    for (int e = tid; e < g.Cin * TC; e += 256) {
        const int c = e / TC, j = e % TC, t = j / CO, o = j % CO;
        wS[e] = w[((size_t)o * g.Cin + c) * T + t];
    }
    // global byte offset of each halo voxel this thread owns (step 1) -- also the rows it helps to stage
    unsigned src[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int hv = tid + 256 * k;
        const int hx = hv % g.HWd, hy = (hv / g.HWd) % g.HH, hz = hv / (g.HWd * g.HH);
        const int iz = d0 + hz - g.pd, iy = h0 + hy - g.ph, ix = w0 + hx - g.pw;
        const bool ok = hv < g.HV && iz >= 0 && iz < g.D && iy >= 0 && iy < g.H && ix >= 0 && ix < g.W;
        src[k] = ok ? (unsigned)((((b * g.D + iz) * g.H + iy) * g.W + ix) * g.Cin) * 4u : SC_OOB;
    }
    // the staging pass needs every row's offset: through LDS (the z image is free until step 1 ends)
    unsigned* srcS = reinterpret_cast<unsigned*>(zS);
#pragma unroll
    for (int k = 0; k < NV; ++k)
        if (tid + 256 * k < g.HV) srcS[tid + 256 * k] = src[k];
    const auto rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, (int)g.xBytes, 0x00020000);

    float acc[NV][TC];
#pragma unroll
    for (int k = 0; k < NV; ++k)
#pragma unroll
        for (int j = 0; j < TC; ++j) acc[k][j] = 0.f;
    __syncthreads();

    for (int ch = 0; ch < g.nChunks; ++ch) {
        const int c0 = ch * g.cch, cn = min(g.cch, g.Cin - c0);
        // ---- stage x[halo][c0 .. c0 + cn) : consecutive lanes read consecutive floats of a row ----
        const int total = g.HV * cn;
        for (int e0 = 0; e0 < total; e0 += 256 * 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = e0 + u * 256 + tid;
                const int hv = min(e / cn, g.HV - 1), c = e % cn;
                const unsigned off = e < total ? srcS[hv] : SC_OOB;
                v[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_x, off + (unsigned)(c0 + c) * 4u, 0, 0));
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = e0 + u * 256 + tid;
                if (e < total) xS[(e / cn) * g.rowS + e % cn] = v[u];
            }
        }
        __syncthreads();
        // ---- step 1: this thread's halo voxels ----
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int hv = tid + 256 * k;
            if (hv < g.HV && src[k] != SC_OOB) {
                const float* xr = xS + hv * g.rowS;
                const float* wr = wS + (size_t)c0 * TC;
                for (int c = 0; c < cn; ++c) {
                    const float xv = xr[c];
#pragma unroll
                    for (int j = 0; j < TC; ++j) acc[k][j] = fmaf(xv, wr[c * TC + j], acc[k][j]);
                }
            }
        }
        __syncthreads();
    }
    // ---- z image ----
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int hv = tid + 256 * k;
        if (hv < g.HV) {
#pragma unroll
            for (int j = 0; j < TC; ++j) zS[hv * TCP + j] = acc[k][j];
        }
    }
    __syncthreads();
    // ---- step 2: one output voxel per thread ----
    {
        const int tw = tid % g.TW, th = (tid / g.TW) % g.TH, td = tid / (g.TW * g.TH);
        const int od = d0 + td, oh = h0 + th, ow = w0 + tw;
        if (od < g.Do && oh < g.Ho && ow < g.Wo) {
            const int hbase = (td * g.HH + th) * g.HWd + tw;
            float o[CO];
#pragma unroll
            for (int q = 0; q < CO; ++q) o[q] = bias ? bias[q] : 0.f;
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const int kz = t / (KW * KH), ky = (t / KW) % KH, kx = t % KW;
                const float* zr = zS + (hbase + (kz * g.HH + ky) * g.HWd + kx) * TCP + t * CO;
#pragma unroll
                for (int q = 0; q < CO; ++q) o[q] += zr[q];
            }
            const size_t oi = ((((size_t)b * g.Do + od) * g.Ho + oh) * g.Wo + ow) * CO;
#pragma unroll
            for (int q = 0; q < CO; ++q) y[oi + q] = o[q] + (residual ? residual[oi + q] : 0.f);
        }
    }
}

bool sc_geom(SCGeom& g, size_t& lds, int B, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw, int pd, int ph, int pw,
             int epd, int eph, int epw) {
    if (B <= 0 || D <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return false;
    if (pd < 0 || ph < 0 || pw < 0 || pd + epd < 0 || ph + eph < 0 || pw + epw < 0) return false;
    g.B = B; g.D = D; g.H = H; g.W = W; g.Cin = Cin; g.pd = pd; g.ph = ph; g.pw = pw;
    g.Do = D + 2 * pd + epd - kd + 1; g.Ho = H + 2 * ph + eph - kh + 1; g.Wo = W + 2 * pw + epw - kw + 1;
    if (g.Do <= 0 || g.Ho <= 0 || g.Wo <= 0) return false;
    const int T = kd * kh * kw, TC = T * Cout, TCP = TC | 1;
    static const int cand[][3] = {{4, 8, 8}, {1, 16, 16}, {2, 8, 16}, {16, 4, 4}, {8, 8, 4}, {1, 8, 32}, {1, 4, 64}, {1, 1, 256}, {32, 4, 2}, {64, 2, 2}};
    double best = 1e300;
    bool found = false;
    for (auto& c : cand) {
        const int hv = (c[0] + kd - 1) * (c[1] + kh - 1) * (c[2] + kw - 1);
        if (hv > 768) continue;
        const double tiles = (double)((g.Do + c[0] - 1) / c[0]) * ((g.Ho + c[1] - 1) / c[1]) * ((g.Wo + c[2] - 1) / c[2]);
        const double cost = tiles * hv;
        if (cost < best) { best = cost; g.TD = c[0]; g.TH = c[1]; g.TW = c[2]; found = true; }
    }
    if (!found) return false;
    g.HD = g.TD + kd - 1; g.HH = g.TH + kh - 1; g.HWd = g.TW + kw - 1; g.HV = g.HD * g.HH * g.HWd;
    g.tilesD = (g.Do + g.TD - 1) / g.TD; g.tilesH = (g.Ho + g.TH - 1) / g.TH; g.tilesW = (g.Wo + g.TW - 1) / g.TW;
    if ((long long)B * g.tilesD * g.tilesH * g.tilesW >= (1ll << 31)) return false;
    const unsigned long long xb = (unsigned long long)B * D * H * W * Cin * 4ull, yb = (unsigned long long)B * g.Do * g.Ho * g.Wo * Cout * 4ull;
    if (xb >= (1ull << 31) || yb >= (1ull << 31)) return false;
    g.xBytes = (unsigned)xb; g.yBytes = (unsigned)yb;
    // channel chunk: the staged x image within what is left of an LDS budget that keeps three, two or one workgroup per CU
    const size_t fixed = ((size_t)Cin * TC + (size_t)g.HV * TCP) * sizeof(float);
    int cch = 0;
    for (size_t budget : {(size_t)52 * 1024, (size_t)78 * 1024, (size_t)150 * 1024}) {
        const size_t room = budget > fixed ? budget - fixed : 0;
        cch = (int)(room / sizeof(float) / g.HV) - 1;
        if (cch > Cin) cch = Cin;
        if (cch >= 16 || (cch >= 1 && cch == Cin)) break;
    }
    if (cch < 8 && cch < Cin) return false;
    g.cch = cch; g.nChunks = (Cin + cch - 1) / cch; g.rowS = cch | 1;
    lds = fixed + (size_t)g.HV * g.rowS * sizeof(float);
    return lds <= 160 * 1024;
}

// (filter, Cout) pairs with an instantiation
int sc_variant(int kd, int kh, int kw, int Cout) {
    if (Cout == 1 && kd == 1 && kh == 1 && kw == 1) return 1;
    if (Cout == 1 && kd == 1 && kh == 3 && kw == 3) return 2;
    if (Cout == 1 && kd == 3 && kh == 3 && kw == 3) return 3;
    if (Cout == 1 && kd == 3 && kh == 1 && kw == 1) return 4;
    if (Cout == 2 && kd == 1 && kh == 3 && kw == 3) return 5;
    if (Cout == 2 && kd == 1 && kh == 1 && kw == 1) return 6;
    return 0;
}

}  // namespace
}  // namespace diqt

using namespace diqt;

extern "C" int diqt_conv3d_fwd_smallcout_supported(int B, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw, int pd, int ph,
                                                   int pw, int epd, int eph, int epw) {
    static const bool off = [] { const char* e = getenv("DIQT_NO_SMALLCOUT"); return e && e[0] == '1'; }();
    if (off || !sc_variant(kd, kh, kw, Cout) || Cin < 16) return 0;
    SCGeom g;
    size_t lds;
    return sc_geom(g, lds, B, D, H, W, Cin, Cout, kd, kh, kw, pd, ph, pw, epd, eph, epw) ? 1 : 0;
}

extern "C" int diqt_conv3d_fwd_smallcout(const float* x, const float* w_oidhw, const float* bias, const float* residual, float* y, int B,
                                         int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw, int pd, int ph, int pw, int epd,
                                         int eph, int epw, void* stream) {
    DIQT_REQUIRE(x && w_oidhw && y, DIQT_E_ALIGN, "conv3d_fwd_smallcout: null pointer");
    SCGeom g;
    size_t lds = 0;
    const int v = sc_variant(kd, kh, kw, Cout);
    DIQT_REQUIRE(v && sc_geom(g, lds, B, D, H, W, Cin, Cout, kd, kh, kw, pd, ph, pw, epd, eph, epw), DIQT_E_UNSUPPORTED,
                 "conv3d_fwd_smallcout: shape not supported (diqt_conv3d_fwd_smallcout_supported == 0)");
    void (*kern)(const float*, const float*, const float*, const float*, float*, SCGeom) =
        v == 1 ? conv_smallcout_kernel<1, 1, 1, 1> : v == 2 ? conv_smallcout_kernel<1, 3, 3, 1> : v == 3 ? conv_smallcout_kernel<3, 3, 3, 1>
      : v == 4 ? conv_smallcout_kernel<3, 1, 1, 1> : v == 5 ? conv_smallcout_kernel<1, 3, 3, 2> : conv_smallcout_kernel<1, 1, 1, 2>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        DIQT_REQUIRE(e == hipSuccess, DIQT_E_LAUNCH, "conv3d_fwd_smallcout: hipFuncSetAttribute: %s", hipGetErrorString(e));
    }
    const unsigned nwg = (unsigned)((long long)B * g.tilesD * g.tilesH * g.tilesW);
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), lds, (hipStream_t)stream, x, w_oidhw, bias, residual, y, g);
    return check_launch("conv3d_fwd_smallcout");
}
