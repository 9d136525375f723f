// Training data path and validation metrics on the device:
//   * random-crop patch pairs out of HBM-resident volume pairs (data.py:50-137 supervisedIQT.__getitem__: the crop, the
//     z-score / min-max normalisation; the crop origins and the non-zero rejection are decided by the host from a
//     summed-area table, diffusioniqt_amd/data.py),
//   * PSNR / SSIM of valid_step (metrics.py:19-31 -> torchmetrics 0.9.0 peak_signal_noise_ratio and
//     StructuralSimilarityIndexMeasure on min-max normalised 5-D tensors).
// All reductions are two-stage with a fixed order (bit-reproducible).
#include "common.h"

namespace diqt {
#define STREAM ((hipStream_t)stream)

struct MinMax { float lo, hi; };

__device__ __forceinline__ MinMax wg_minmax(float lo, float hi, MinMax* sh) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, o, 64));
        hi = fmaxf(hi, __shfl_xor(hi, o, 64));
    }
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = MinMax{lo, hi};
    __syncthreads();
    MinMax r = sh[0];
    for (unsigned w = 1; w < blockDim.x / 64; ++w) { r.lo = fminf(r.lo, sh[w].lo); r.hi = fmaxf(r.hi, sh[w].hi); }
    __syncthreads();
    return r;
}

// sel[n] = {volume, i0, j0, k0}.  grid (nb, n, 2): z = 0 low-res, 1 high-res.
__global__ __launch_bounds__(256) void crop_minmax_kernel(const float* __restrict__ lr, const float* __restrict__ hr,
                                                          const int* __restrict__ sel, MinMax* __restrict__ part, int D, int H,
                                                          int W, int P) {
    __shared__ MinMax sh[4];
    const int n = blockIdx.y;
    const float* vol = (blockIdx.z ? hr : lr) + (size_t)sel[4 * n] * D * H * W;
    const int i0 = sel[4 * n + 1], j0 = sel[4 * n + 2], k0 = sel[4 * n + 3];
    const size_t per = (size_t)P * P * P;
    float lo = INFINITY, hi = -INFINITY;
    for (size_t e = blockIdx.x * (size_t)256 + threadIdx.x; e < per; e += (size_t)gridDim.x * 256) {
        const int k = (int)(e % P), j = (int)((e / P) % P), i = (int)(e / ((size_t)P * P));
        const float v = vol[((size_t)(i0 + i) * H + (j0 + j)) * W + (k0 + k)];
        lo = fminf(lo, v);
        hi = fmaxf(hi, v);
    }
    const MinMax r = wg_minmax(lo, hi, sh);
    if (threadIdx.x == 0) part[((size_t)blockIdx.z * gridDim.y + n) * gridDim.x + blockIdx.x] = r;
}
// mode 0: (v - mean) / std ; mode 1: 2 * ((v - min) / (max - min) - 0.5) with the patch's own min / max (data.py:82-86)
__global__ __launch_bounds__(256) void crop_apply_kernel(const float* __restrict__ lr, const float* __restrict__ hr,
                                                         const int* __restrict__ sel, const MinMax* __restrict__ part,
                                                         float* __restrict__ lr_out, float* __restrict__ hr_out, int D, int H, int W,
                                                         int P, int mode, float mean, float stdv) {
    const int n = blockIdx.y;
    const float* vol = (blockIdx.z ? hr : lr) + (size_t)sel[4 * n] * D * H * W;
    float* out = (blockIdx.z ? hr_out : lr_out) + (size_t)n * P * P * P;
    const int i0 = sel[4 * n + 1], j0 = sel[4 * n + 2], k0 = sel[4 * n + 3];
    const size_t per = (size_t)P * P * P;
    float lo = 0.f, range = 1.f;
    if (mode == 1) {
        const MinMax* pp = part + ((size_t)blockIdx.z * gridDim.y + n) * gridDim.x;
        float hi = -INFINITY;
        lo = INFINITY;
        for (unsigned b = 0; b < gridDim.x; ++b) { lo = fminf(lo, pp[b].lo); hi = fmaxf(hi, pp[b].hi); }
        range = hi - lo;
    }
    for (size_t e = blockIdx.x * (size_t)256 + threadIdx.x; e < per; e += (size_t)gridDim.x * 256) {
        const int k = (int)(e % P), j = (int)((e / P) % P), i = (int)(e / ((size_t)P * P));
        const float v = vol[((size_t)(i0 + i) * H + (j0 + j)) * W + (k0 + k)];
        out[e] = mode == 1 ? 2.f * ((v - lo) / range - 0.5f) : (v - mean) / stdv;
    }
}

// ---- global min / max -----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void minmax_stage1_kernel(const float* __restrict__ x, MinMax* __restrict__ part, size_t n) {
    __shared__ MinMax sh[4];
    float lo = INFINITY, hi = -INFINITY;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float v = x[i];
        lo = fminf(lo, v);
        hi = fmaxf(hi, v);
    }
    const MinMax r = wg_minmax(lo, hi, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = r;
}
__global__ __launch_bounds__(64) void minmax_stage2_kernel(const MinMax* __restrict__ part, int nb, float* __restrict__ out) {
    float lo = INFINITY, hi = -INFINITY;
    for (int i = threadIdx.x; i < nb; i += 64) { lo = fminf(lo, part[i].lo); hi = fmaxf(hi, part[i].hi); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, o, 64));
        hi = fmaxf(hi, __shfl_xor(hi, o, 64));
    }
    if (threadIdx.x == 0) { out[0] = lo; out[1] = hi; }
}

// ---- PSNR: mean squared error of the (optionally min-max normalised) tensors -----------------------------------------
// stats = {pred min, pred max, target min, target max} on the device, or NULL for "compare as they are"
__global__ __launch_bounds__(256) void sqerr_stage1_kernel(const float* __restrict__ p, const float* __restrict__ t,
                                                           const float* __restrict__ stats, double* __restrict__ part, size_t n) {
    __shared__ double sh[4];
    float pl = 0.f, pr = 1.f, tl = 0.f, tr = 1.f;
    if (stats) { pl = stats[0]; pr = stats[1] - stats[0]; tl = stats[2]; tr = stats[3] - stats[2]; }
    double acc = 0.0;
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float a = stats ? (p[i] - pl) / pr : p[i], b = stats ? (t[i] - tl) / tr : t[i];
        const float d = a - b;
        acc += (double)(d * d);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}
// out[0] = mse, out[1] = 10 log10(data_range^2 / mse)   (torchmetrics 0.9.0 functional/image/psnr.py: _psnr_compute, base 10)
__global__ __launch_bounds__(64) void psnr_stage2_kernel(const double* __restrict__ part, int nb, double count, float data_range,
                                                         float* __restrict__ out) {
    if (threadIdx.x) return;
    double s = 0.0;
    for (int i = 0; i < nb; ++i) s += part[i];
    const float mse = (float)(s / count);
    out[0] = mse;
    out[1] = (2.f * logf(data_range) - logf(mse)) * (10.f / logf(10.f));
}

// ---- SSIM ------------------------------------------------------------------------------------------------------------
// torchmetrics 0.9.0 functional/image/ssim.py _ssim_compute on 5-D input: reflect-pad by R = (K-1)/2, depthwise K^3 Gaussian
// filter of {p, t, p^2, t^2, p t}, SSIM map, crop R from every face, mean.  After the crop only windows that lie fully inside
// the un-padded volume survive, so the padding never contributes: the kernel evaluates the "valid" windows only.
constexpr int ST = 8;           // outputs per tile edge
constexpr int SK = 11;          // largest filter
constexpr int SI = ST + SK - 1; // input tile edge (18)
struct SsimTaps { float w[SK]; int K; };

__global__ __launch_bounds__(256) void ssim_tile_kernel(const float* __restrict__ p, const float* __restrict__ t,
                                                        const float* __restrict__ stats, double* __restrict__ part, int D, int H,
                                                        int W, int tilesD, int tilesH, int tilesW, SsimTaps taps, float c1,
                                                        float c2) {
    extern __shared__ float lds[];
    const int K = taps.K, I = ST + K - 1;
    float* raw = lds;                                  // [2][I][I][I]
    float* f1 = raw + 2 * SI * SI * SI;                // [5][I][I][ST]   filtered along W
    float* f2 = f1 + 5 * SI * SI * ST;                 // [5][I][ST][ST]  filtered along H
    __shared__ double sh[4];
    const int Do = D - K + 1, Ho = H - K + 1, Wo = W - K + 1;
    unsigned b = blockIdx.x;
    const int tw = b % tilesW; b /= tilesW;
    const int th = b % tilesH; b /= tilesH;
    const int td = b % tilesD;
    const int vol = b / tilesD;
    const float* pv = p + (size_t)vol * D * H * W;
    const float* tv = t + (size_t)vol * D * H * W;
    float pl = 0.f, pr = 1.f, tl = 0.f, tr = 1.f;
    if (stats) { pl = stats[0]; pr = stats[1] - stats[0]; tl = stats[2]; tr = stats[3] - stats[2]; }
    const int d0 = td * ST, h0 = th * ST, w0 = tw * ST;
    for (int e = threadIdx.x; e < I * I * I; e += 256) {
        const int k = e % I, j = (e / I) % I, i = e / (I * I);
        const int di = min(d0 + i, D - 1), hj = min(h0 + j, H - 1), wk = min(w0 + k, W - 1);   // clamped reads feed masked outputs only
        const size_t off = ((size_t)di * H + hj) * W + wk;
        const float a = pv[off], c = tv[off];
        raw[(i * SI + j) * SI + k] = stats ? (a - pl) / pr : a;
        raw[SI * SI * SI + (i * SI + j) * SI + k] = stats ? (c - tl) / tr : c;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < I * I * ST; e += 256) {          // along W
        const int k = e % ST, j = (e / ST) % I, i = e / (ST * I);
        const float* rp = raw + (i * SI + j) * SI + k;
        const float* rt = rp + SI * SI * SI;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f;
        for (int q = 0; q < K; ++q) {
            const float w = taps.w[q], a = rp[q], c = rt[q];
            s0 = fmaf(w, a, s0); s1 = fmaf(w, c, s1); s2 = fmaf(w, a * a, s2); s3 = fmaf(w, c * c, s3); s4 = fmaf(w, a * c, s4);
        }
        float* o = f1 + (i * SI + j) * ST + k;
        o[0] = s0; o[SI * SI * ST] = s1; o[2 * SI * SI * ST] = s2; o[3 * SI * SI * ST] = s3; o[4 * SI * SI * ST] = s4;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 5 * I * ST * ST; e += 256) {     // along H
        const int k = e % ST, j = (e / ST) % ST, i = (e / (ST * ST)) % I, f = e / (ST * ST * I);
        const float* r = f1 + f * SI * SI * ST + (i * SI + j) * ST + k;
        float s = 0.f;
        for (int q = 0; q < K; ++q) s = fmaf(taps.w[q], r[q * ST], s);
        f2[f * SI * ST * ST + (i * ST + j) * ST + k] = s;
    }
    __syncthreads();
    double acc = 0.0;
    for (int e = threadIdx.x; e < ST * ST * ST; e += 256) {        // along D + the SSIM map
        const int k = e % ST, j = (e / ST) % ST, i = e / (ST * ST);
        if (d0 + i >= Do || h0 + j >= Ho || w0 + k >= Wo) continue;
        float m[5];
#pragma unroll
        for (int f = 0; f < 5; ++f) {
            const float* r = f2 + f * SI * ST * ST + (i * ST + j) * ST + k;
            float s = 0.f;
            for (int q = 0; q < K; ++q) s = fmaf(taps.w[q], r[q * ST * ST], s);
            m[f] = s;
        }
        const float mp2 = m[0] * m[0], mt2 = m[1] * m[1], mpt = m[0] * m[1];
        const float sp = m[2] - mp2, stt = m[3] - mt2, spt = m[4] - mpt;
        const float upper = 2.f * spt + c2, lower = sp + stt + c2;
        acc += (double)(((2.f * mpt + c1) * upper) / ((mp2 + mt2 + c1) * lower));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}
__global__ __launch_bounds__(64) void mean_stage2_kernel(const double* __restrict__ part, int nb, double count, float* __restrict__ out) {
    if (threadIdx.x) return;
    double s = 0.0;
    for (int i = 0; i < nb; ++i) s += part[i];
    out[0] = (float)(s / count);
}
}  // namespace diqt

using namespace diqt;

static unsigned crop_blocks(int P) { return grid_for((size_t)P * P * P, 256, 64); }

extern "C" size_t diqt_patch_pair_crop_workspace_bytes(int n_patches, int P) {
    if (n_patches <= 0 || P <= 0) return 0;
    return (size_t)2 * n_patches * crop_blocks(P) * sizeof(MinMax);
}
extern "C" int diqt_patch_pair_crop(const float* lr_vols, const float* hr_vols, const int* sel, float* lr_out, float* hr_out,
                                    void* workspace, size_t workspace_bytes, int n_patches, int V, int D, int H, int W, int P,
                                    int mode, float mean, float stdv, void* stream) {
    DIQT_REQUIRE(lr_vols && hr_vols && sel && lr_out && hr_out, DIQT_E_ALIGN, "patch_pair_crop: null pointer");
    DIQT_REQUIRE(V > 0 && D > 0 && H > 0 && W > 0 && P > 0 && P <= D && P <= H && P <= W, DIQT_E_SHAPE, "patch_pair_crop: bad shape");
    DIQT_REQUIRE(mode == 0 || mode == 1, DIQT_E_UNSUPPORTED, "patch_pair_crop: mode %d (0 = z-score, 1 = min-max)", mode);
    DIQT_REQUIRE(mode == 1 || stdv != 0.f, DIQT_E_SHAPE, "patch_pair_crop: std == 0");
    if (n_patches <= 0) return DIQT_OK;
    DIQT_REQUIRE(n_patches <= 65535, DIQT_E_SHAPE, "patch_pair_crop: at most 65535 patches per call");
    const unsigned nb = crop_blocks(P);
    MinMax* part = static_cast<MinMax*>(workspace);
    if (mode == 1) {
        DIQT_REQUIRE(workspace && workspace_bytes >= diqt_patch_pair_crop_workspace_bytes(n_patches, P), DIQT_E_SHAPE,
                     "patch_pair_crop: workspace too small");
        hipLaunchKernelGGL(crop_minmax_kernel, dim3(nb, n_patches, 2), dim3(256), 0, STREAM, lr_vols, hr_vols, sel, part, D, H, W, P);
        int rc = check_launch("patch_pair_crop/minmax");
        if (rc) return rc;
    }
    hipLaunchKernelGGL(crop_apply_kernel, dim3(nb, n_patches, 2), dim3(256), 0, STREAM, lr_vols, hr_vols, sel, part, lr_out, hr_out, D, H,
                       W, P, mode, mean, stdv);
    return check_launch("patch_pair_crop");
}

extern "C" int diqt_minmax(const float* x, size_t n, void* workspace_8k, float* out2, void* stream) {
    DIQT_REQUIRE(x && workspace_8k && out2 && n > 0, DIQT_E_ALIGN, "minmax: null pointer / empty input");
    const unsigned nb = grid_for(n, 256, 1024);
    hipLaunchKernelGGL(minmax_stage1_kernel, dim3(nb), dim3(256), 0, STREAM, x, static_cast<MinMax*>(workspace_8k), n);
    int rc = check_launch("minmax/stage1");
    if (rc) return rc;
    hipLaunchKernelGGL(minmax_stage2_kernel, dim3(1), dim3(64), 0, STREAM, static_cast<const MinMax*>(workspace_8k), (int)nb, out2);
    return check_launch("minmax/stage2");
}

extern "C" int diqt_psnr(const float* pred, const float* target, size_t n, const float* stats4, float data_range, void* workspace_8k,
                         float* out2, void* stream) {
    DIQT_REQUIRE(pred && target && workspace_8k && out2 && n > 0, DIQT_E_ALIGN, "psnr: null pointer / empty input");
    DIQT_REQUIRE(data_range > 0.f, DIQT_E_SHAPE, "psnr: data_range must be positive");
    const unsigned nb = grid_for(n, 256, 1024);
    hipLaunchKernelGGL(sqerr_stage1_kernel, dim3(nb), dim3(256), 0, STREAM, pred, target, stats4, static_cast<double*>(workspace_8k), n);
    int rc = check_launch("psnr/stage1");
    if (rc) return rc;
    hipLaunchKernelGGL(psnr_stage2_kernel, dim3(1), dim3(64), 0, STREAM, static_cast<const double*>(workspace_8k), (int)nb, (double)n,
                       data_range, out2);
    return check_launch("psnr/stage2");
}

static void ssim_tiles(int N, int D, int H, int W, int K, int& tD, int& tH, int& tW, size_t& blocks) {
    tD = (D - K + 1 + ST - 1) / ST; tH = (H - K + 1 + ST - 1) / ST; tW = (W - K + 1 + ST - 1) / ST;
    blocks = (size_t)N * tD * tH * tW;
}
extern "C" size_t diqt_ssim3d_workspace_bytes(int N, int D, int H, int W, int K) {
    if (N <= 0 || K < 1 || K > SK || D < K || H < K || W < K) return 0;
    int a, b, c;
    size_t blocks;
    ssim_tiles(N, D, H, W, K, a, b, c, blocks);
    return blocks * sizeof(double);
}
extern "C" int diqt_ssim3d(const float* pred, const float* target, int N, int D, int H, int W, const float* taps, int K,
                           const float* stats4, float data_range, float k1, float k2, void* workspace, size_t workspace_bytes,
                           float* out, void* stream) {
    DIQT_REQUIRE(pred && target && taps && out, DIQT_E_ALIGN, "ssim3d: null pointer");
    DIQT_REQUIRE(K >= 1 && K <= SK && (K & 1), DIQT_E_UNSUPPORTED, "ssim3d: filter size %d (odd, <= %d)", K, SK);
    DIQT_REQUIRE(N > 0 && D >= K && H >= K && W >= K, DIQT_E_SHAPE, "ssim3d: volume smaller than the filter");
    int tD, tH, tW;
    size_t blocks;
    ssim_tiles(N, D, H, W, K, tD, tH, tW, blocks);
    DIQT_REQUIRE(blocks <= 0x7fffffffu, DIQT_E_SHAPE, "ssim3d: too many tiles");
    DIQT_REQUIRE(workspace && workspace_bytes >= blocks * sizeof(double), DIQT_E_SHAPE, "ssim3d: workspace too small");
    SsimTaps tp;
    for (int i = 0; i < SK; ++i) tp.w[i] = i < K ? taps[i] : 0.f;      // taps: HOST pointer
    tp.K = K;
    const float c1 = (k1 * data_range) * (k1 * data_range), c2 = (k2 * data_range) * (k2 * data_range);
    const size_t lds = (size_t)(2 * SI * SI * SI + 5 * SI * SI * ST + 5 * SI * ST * ST) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(ssim_tile_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        DIQT_REQUIRE(e == hipSuccess, DIQT_E_LAUNCH, "ssim3d: LDS attribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    hipLaunchKernelGGL(ssim_tile_kernel, dim3((unsigned)blocks), dim3(256), lds, STREAM, pred, target, stats4,
                       static_cast<double*>(workspace), D, H, W, tD, tH, tW, tp, c1, c2);
    int rc = check_launch("ssim3d/tiles");
    if (rc) return rc;
    const double count = (double)N * (D - K + 1) * (H - K + 1) * (W - K + 1);
    hipLaunchKernelGGL(mean_stage2_kernel, dim3(1), dim3(64), 0, STREAM, static_cast<const double*>(workspace), (int)blocks, count, out);
    return check_launch("ssim3d/mean");
}
