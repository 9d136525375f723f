// Direct grouped / strided 3-D convolution (VALU) for the FLOP-trivial shapes of the attention blocks:
// depthwise 3^3, patchify k=s=p, temporal depthwise (3,1,1).  Channels-last, one thread per output
// element with the channel index fastest (coalesced); backward-weight accumulates voxel slices with
// float atomics into a zeroed buffer (these layers are <0.1% of the path's FLOPs).
#include "common.h"

namespace diqt {
struct DGeom {
    int B, D, H, W, Cin, Cout, groups, kd, kh, kw, sd, sh, sw, pd, ph, pw, Do, Ho, Wo;
};

__global__ __launch_bounds__(256) void direct_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ y, DGeom g) {
    const int cig = g.Cin / g.groups, cog = g.Cout / g.groups, T = g.kd * g.kh * g.kw;
    const size_t total = (size_t)g.B * g.Do * g.Ho * g.Wo * g.Cout;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int co = (int)(i % g.Cout);
        size_t r = i / g.Cout;
        const int ow = (int)(r % g.Wo); r /= g.Wo;
        const int oh = (int)(r % g.Ho); r /= g.Ho;
        const int od = (int)(r % g.Do);
        const int b = (int)(r / g.Do);
        const int grp = co / cog;
        float acc = bias ? bias[co] : 0.f;
        for (int kz = 0; kz < g.kd; ++kz) {
            const int iz = od * g.sd - g.pd + kz;
            if (iz < 0 || iz >= g.D) continue;
            for (int ky = 0; ky < g.kh; ++ky) {
                const int iy = oh * g.sh - g.ph + ky;
                if (iy < 0 || iy >= g.H) continue;
                for (int kx = 0; kx < g.kw; ++kx) {
                    const int ix = ow * g.sw - g.pw + kx;
                    if (ix < 0 || ix >= g.W) continue;
                    const float* xp = x + ((((size_t)b * g.D + iz) * g.H + iy) * g.W + ix) * g.Cin + grp * cig;
                    const float* wp = w + (size_t)co * cig * T + (kz * g.kh + ky) * g.kw + kx;
                    for (int c = 0; c < cig; ++c) acc += xp[c] * wp[(size_t)c * T];
                }
            }
        }
        y[i] = acc;
    }
}

__global__ __launch_bounds__(256) void direct_bwd_data_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                              float* __restrict__ dx, DGeom g) {
    const int cig = g.Cin / g.groups, cog = g.Cout / g.groups, T = g.kd * g.kh * g.kw;
    const size_t total = (size_t)g.B * g.D * g.H * g.W * g.Cin;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int ci = (int)(i % g.Cin);
        size_t r = i / g.Cin;
        const int ix = (int)(r % g.W); r /= g.W;
        const int iy = (int)(r % g.H); r /= g.H;
        const int iz = (int)(r % g.D);
        const int b = (int)(r / g.D);
        const int grp = ci / cig, cl = ci % cig;
        float acc = 0.f;
        for (int kz = 0; kz < g.kd; ++kz) {
            const int tz = iz + g.pd - kz;
            if (tz < 0 || tz % g.sd) continue;
            const int od = tz / g.sd;
            if (od >= g.Do) continue;
            for (int ky = 0; ky < g.kh; ++ky) {
                const int ty = iy + g.ph - ky;
                if (ty < 0 || ty % g.sh) continue;
                const int oh = ty / g.sh;
                if (oh >= g.Ho) continue;
                for (int kx = 0; kx < g.kw; ++kx) {
                    const int tx = ix + g.pw - kx;
                    if (tx < 0 || tx % g.sw) continue;
                    const int ow = tx / g.sw;
                    if (ow >= g.Wo) continue;
                    const float* dp = dy + ((((size_t)b * g.Do + od) * g.Ho + oh) * g.Wo + ow) * g.Cout + grp * cog;
                    const float* wp = w + ((size_t)(grp * cog) * cig + cl) * T + (kz * g.kh + ky) * g.kw + kx;
                    for (int c = 0; c < cog; ++c) acc += dp[c] * wp[(size_t)c * cig * T];
                }
            }
        }
        dx[i] = acc;
    }
}

// grid.x: weight elements (co fastest) ; grid.y: voxel slices.  part == NULL: atomics into zeroed dw / dbias (no workspace);
// otherwise every slice writes part[slice][nW (+ Cout bias sums)] and direct_dw_reduce_kernel sums the slices in a fixed order.
// Voxel coordinates are advanced incrementally (no div/mod per voxel).
__global__ __launch_bounds__(256) void direct_bwd_weight_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                float* __restrict__ dw, float* __restrict__ dbias,
                                                                float* __restrict__ part, DGeom g) {
    const int cig = g.Cin / g.groups, cog = g.Cout / g.groups, T = g.kd * g.kh * g.kw;
    const size_t nW = (size_t)g.Cout * cig * T;
    const size_t widx = blockIdx.x * (size_t)256 + threadIdx.x;
    if (widx >= nW) return;
    // order: co fastest so that neighbouring lanes read neighbouring dy channels
    const int co = (int)(widx % g.Cout);
    size_t r = widx / g.Cout;
    const int cl = (int)(r % cig);
    const int tap = (int)(r / cig);
    const int kx = tap % g.kw, ky = (tap / g.kw) % g.kh, kz = tap / (g.kw * g.kh);
    const int grp = co / cog, ci = grp * cig + cl;
    const size_t nvox = (size_t)g.B * g.Do * g.Ho * g.Wo;
    const size_t per = (nvox + gridDim.y - 1) / gridDim.y;
    const size_t v0 = blockIdx.y * per;
    size_t v1 = v0 + per;
    if (v1 > nvox) v1 = nvox;
    float acc = 0.f, bacc = 0.f;
    size_t q = v0;
    int ow = (int)(q % g.Wo); q /= g.Wo;
    int oh = (int)(q % g.Ho); q /= g.Ho;
    int od = (int)(q % g.Do);
    int b = (int)(q / g.Do);
    for (size_t v = v0; v < v1; ++v) {
        const float d = dy[v * g.Cout + co];
        if (tap == 0 && cl == 0) bacc += d;
        const int iz = od * g.sd - g.pd + kz, iy = oh * g.sh - g.ph + ky, ix = ow * g.sw - g.pw + kx;
        if (iz >= 0 && iz < g.D && iy >= 0 && iy < g.H && ix >= 0 && ix < g.W)
            acc = fmaf(d, x[((((size_t)b * g.D + iz) * g.H + iy) * g.W + ix) * g.Cin + ci], acc);
        if (++ow == g.Wo) { ow = 0; if (++oh == g.Ho) { oh = 0; if (++od == g.Do) { od = 0; ++b; } } }
    }
    const size_t dst = ((size_t)co * cig + cl) * T + tap;
    if (part) {
        float* ps = part + (size_t)blockIdx.y * (nW + g.Cout);
        ps[dst] = acc;
        if (tap == 0 && cl == 0) ps[nW + co] = bacc;
    } else {
        atomicAdd(dw + dst, acc);
        if (dbias && tap == 0 && cl == 0) atomicAdd(dbias + co, bacc);
    }
}
__global__ __launch_bounds__(256) void direct_dw_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw,
                                                               float* __restrict__ dbias, size_t nW, int Cout, int slices) {
    const size_t i = blockIdx.x * (size_t)256 + threadIdx.x;
    if (i >= nW + Cout) return;
    float s = 0.f;
    for (int k = 0; k < slices; ++k) s += part[(size_t)k * (nW + Cout) + i];
    if (i < nW) dw[i] = s;
    else if (dbias) dbias[i - nW] = s;
}

static int dgeom(DGeom& g, int B, int D, int H, int W, int Cin, int Cout, int groups, int kd, int kh, int kw, int sd,
                 int sh, int sw, int pd, int ph, int pw, int epd, int eph, int epw) {
    DIQT_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && groups > 0, DIQT_E_SHAPE, "conv3d_direct: bad extent");
    DIQT_REQUIRE(Cin % groups == 0 && Cout % groups == 0, DIQT_E_SHAPE, "conv3d_direct: channels not divisible by groups");
    DIQT_REQUIRE(kd > 0 && kh > 0 && kw > 0 && sd > 0 && sh > 0 && sw > 0 && pd >= 0 && ph >= 0 && pw >= 0, DIQT_E_SHAPE,
                 "conv3d_direct: bad filter/stride/pad");
    g = DGeom{B, D, H, W, Cin, Cout, groups, kd, kh, kw, sd, sh, sw, pd, ph, pw,
              (D + 2 * pd + epd - kd) / sd + 1, (H + 2 * ph + eph - kh) / sh + 1, (W + 2 * pw + epw - kw) / sw + 1};
    DIQT_REQUIRE(pd + epd >= 0 && ph + eph >= 0 && pw + epw >= 0, DIQT_E_SHAPE, "conv3d_direct: negative high-side pad");
    DIQT_REQUIRE(g.Do > 0 && g.Ho > 0 && g.Wo > 0, DIQT_E_SHAPE, "conv3d_direct: empty output");
    return DIQT_OK;
}
}  // namespace diqt

using namespace diqt;
extern "C" int diqt_conv3d_direct_fwd(const float* x, const float* w, const float* bias, float* y, int B, int D, int H,
                                      int W, int Cin, int Cout, int groups, int kd, int kh, int kw, int sd, int sh,
                                      int sw, int pd, int ph, int pw, int epd, int eph, int epw, void* stream) {
    DIQT_REQUIRE(x && w && y, DIQT_E_ALIGN, "conv3d_direct_fwd: null pointer");
    DGeom g;
    int rc = dgeom(g, B, D, H, W, Cin, Cout, groups, kd, kh, kw, sd, sh, sw, pd, ph, pw, epd, eph, epw);
    if (rc) return rc;
    const size_t total = (size_t)g.B * g.Do * g.Ho * g.Wo * g.Cout;
    hipLaunchKernelGGL(direct_fwd_kernel, dim3(grid_for(total, 256, 65536)), dim3(256), 0, (hipStream_t)stream, x, w, bias, y, g);
    return check_launch("conv3d_direct_fwd");
}
extern "C" int diqt_conv3d_direct_bwd_data(const float* dy, const float* w, float* dx, int B, int D, int H, int W,
                                           int Cin, int Cout, int groups, int kd, int kh, int kw, int sd, int sh,
                                           int sw, int pd, int ph, int pw, int epd, int eph, int epw, void* stream) {
    DIQT_REQUIRE(dy && w && dx, DIQT_E_ALIGN, "conv3d_direct_bwd_data: null pointer");
    DGeom g;
    int rc = dgeom(g, B, D, H, W, Cin, Cout, groups, kd, kh, kw, sd, sh, sw, pd, ph, pw, epd, eph, epw);
    if (rc) return rc;
    const size_t total = (size_t)g.B * g.D * g.H * g.W * g.Cin;
    hipLaunchKernelGGL(direct_bwd_data_kernel, dim3(grid_for(total, 256, 65536)), dim3(256), 0, (hipStream_t)stream, dy, w, dx, g);
    return check_launch("conv3d_direct_bwd_data");
}
static unsigned direct_slices(size_t nvox) {
    unsigned slices = (unsigned)((nvox + 255) / 256);
    if (slices > 256) slices = 256;
    return slices < 1 ? 1 : slices;
}
extern "C" size_t diqt_conv3d_direct_bwd_weight_workspace_bytes(int B, int D, int H, int W, int Cin, int Cout, int groups, int kd,
                                                                int kh, int kw, int sd, int sh, int sw, int pd, int ph, int pw,
                                                                int epd, int eph, int epw) {
    DGeom g;
    if (dgeom(g, B, D, H, W, Cin, Cout, groups, kd, kh, kw, sd, sh, sw, pd, ph, pw, epd, eph, epw)) return 0;
    const size_t nW = (size_t)Cout * (Cin / groups) * kd * kh * kw;
    return (size_t)direct_slices((size_t)g.B * g.Do * g.Ho * g.Wo) * (nW + Cout) * sizeof(float);
}
static int direct_bwd_weight_impl(const float* x, const float* dy, float* dw, float* dbias, void* workspace, size_t workspace_bytes,
                                  int B, int D, int H, int W, int Cin, int Cout, int groups, int kd, int kh, int kw, int sd, int sh,
                                  int sw, int pd, int ph, int pw, int epd, int eph, int epw, void* stream) {
    DIQT_REQUIRE(x && dy && dw, DIQT_E_ALIGN, "conv3d_direct_bwd_weight: null pointer");
    DGeom g;
    int rc = dgeom(g, B, D, H, W, Cin, Cout, groups, kd, kh, kw, sd, sh, sw, pd, ph, pw, epd, eph, epw);
    if (rc) return rc;
    const size_t nW = (size_t)Cout * (Cin / groups) * kd * kh * kw;
    hipStream_t s = (hipStream_t)stream;
    const size_t nvox = (size_t)g.B * g.Do * g.Ho * g.Wo;
    const unsigned slices = direct_slices(nvox);
    float* part = nullptr;
    if (workspace && workspace_bytes >= (size_t)slices * (nW + Cout) * sizeof(float)) part = static_cast<float*>(workspace);
    if (!part) {
        hipError_t e = hipMemsetAsync(dw, 0, nW * sizeof(float), s);
        if (e == hipSuccess && dbias) e = hipMemsetAsync(dbias, 0, (size_t)Cout * sizeof(float), s);
        DIQT_REQUIRE(e == hipSuccess, DIQT_E_LAUNCH, "conv3d_direct_bwd_weight: memset: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(direct_bwd_weight_kernel, dim3((unsigned)((nW + 255) / 256), slices), dim3(256), 0, s, x, dy, dw, dbias, part, g);
    rc = check_launch("conv3d_direct_bwd_weight");
    if (rc || !part) return rc;
    hipLaunchKernelGGL(direct_dw_reduce_kernel, dim3((unsigned)((nW + Cout + 255) / 256)), dim3(256), 0, s, part, dw, dbias, nW, Cout,
                       (int)slices);
    return check_launch("conv3d_direct_bwd_weight(reduce)");
}
extern "C" int diqt_conv3d_direct_bwd_weight(const float* x, const float* dy, float* dw, float* dbias, int B, int D,
                                             int H, int W, int Cin, int Cout, int groups, int kd, int kh, int kw,
                                             int sd, int sh, int sw, int pd, int ph, int pw, int epd, int eph, int epw, void* stream) {
    return direct_bwd_weight_impl(x, dy, dw, dbias, nullptr, 0, B, D, H, W, Cin, Cout, groups, kd, kh, kw, sd, sh, sw, pd, ph, pw, epd,
                                  eph, epw, stream);
}
extern "C" int diqt_conv3d_direct_bwd_weight_ws(const float* x, const float* dy, float* dw, float* dbias, void* workspace,
                                                size_t workspace_bytes, int B, int D, int H, int W, int Cin, int Cout, int groups,
                                                int kd, int kh, int kw, int sd, int sh, int sw, int pd, int ph, int pw, int epd,
                                                int eph, int epw, void* stream) {
    return direct_bwd_weight_impl(x, dy, dw, dbias, workspace, workspace_bytes, B, D, H, W, Cin, Cout, groups, kd, kh, kw, sd, sh, sw,
                                  pd, ph, pw, epd, eph, epw, stream);
}
