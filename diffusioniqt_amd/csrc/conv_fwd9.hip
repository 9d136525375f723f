// Forward / backward-data convolution of the 3x3x3 and (1,3,3) filters of the U-Nets, version 9: the structure that took the weight
// gradient from 0.63 to 0.83 of the f32 MFMA peak (conv_wgrad.hip), applied to the forward implicit GEMM  M = voxels, N = co,
// K = taps x ci  on v_mfma_f32_32x32x2_f32 (exact fp32 products and sums; K is walked chunk-major -- 16 channels x all taps -- where
// conv_fwd_kernel / conv_fwd8_kernel walk 32-channel chunks, so results agree with theirs to rounding, not bit for bit).
//   * ONE wave per SIMD (256 threads, one workgroup per CU) with EIGHT accumulator tiles per wave: a workgroup owns 512 output voxels
//     x 64 co, a wave 128 voxels (4 blocks of 4 x 8) x 64 co.  Twice the voxels per weight byte of the 256-voxel tile: every KiB
//     landing in the CU costs ~100 cycles of matrix-pipe issue (profiles/r02_wgrad_ablation.md), and at 256 voxels the weight stream
//     alone is 6 % of the MFMA time.  Filter, tile extents and blocks per wave are template parameters (F9Cfg): 8x8x8 and 4x8x8
//     (256 voxels, 4 accumulator tiles: the 16^3 level) for 3x3x3, 1x16x32 / 2x16x16 / 4x8x8 for the per-frame (1,3,3) convs of the
//     pseudo-3D U-Net; the host picks the first variant whose tiles fill whole rounds of 256 workgroups.
//   * K is walked in 16-channel chunks so that TWO halo images (10^3 voxels x 64 B) fit the LDS beside a 3-slot ring of 2-tap weight
//     groups (152 KB): the next chunk's halo and the weight group two steps ahead arrive by LDS-DMA (`buffer_load_dwordx4 ... lds`)
//     while the current step computes -- no register staging, no tables, no full stop at a chunk boundary; a step ends with
//     `s_waitcnt vmcnt(0)` + ONE barrier per 128 MFMAs of a wave.  Zero padding / ragged tiles = out-of-range buffer offsets
//     (the DMA writes zeros), per lane from packed tile-independent coordinates (sign-bit test, no branches).
//   * the taps of a chunk are unrolled: every LDS offset of the 1728 MFMAs' operands is an immediate; fragments of tap t+1 are
//     read (ds_read_b128) between the MFMAs of tap t, also across step boundaries.
//   * persistent tile walk: the first chunk of the next tile is prefetched during the last chunk of the current one, the weight
//     ring runs on; a tile boundary is the epilogue (bias, residual, per-tile column sums for the consumer's GroupNorm / SE pool).
//   * small volumes (the 8^3 level: fewer tiles than CUs): gridDim.y splits the chunks into equal shares that write slabs, summed in
//     a fixed order (with bias / residual) by conv_fwd_reduce_kernel.
// Measured (MI355X, 64->64 3x3x3 @ 8x32^3): 411 us = 141 TFLOP/s = 0.90 of the f32 MFMA peak (conv_fwd8_kernel: 444 us); MFMA pipe
// busy 0.92 of the kernel's cycles, 6 % of wave cycles waiting (profiles/r02_pmc_sq_conv.json).
// Reference call sites: Block.project of every ResnetBlock (/root/reference/imagen_pytorch3D.py:535-566) and the per-frame Conv2d of
// the pseudo-3D blocks (/root/reference/imagen_video.py:352-381 Conv3d.spatial_conv, 671-697 Block).
#include "conv_fwd9_kernel.h"
#include <stdlib.h>

namespace diqt {


template <class C> static bool f9_try(F9Geom& g, size_t& lds, unsigned& grid, int mode, bool maySplit) {
    g.tilesD = (g.Do + C::TD - 1) / C::TD; g.tilesH = (g.Ho + C::TH - 1) / C::TH; g.tilesW = (g.Wo + C::TW - 1) / C::TW;
    const long long mt = (long long)g.B * g.tilesD * g.tilesH * g.tilesW;
    if (mt >= (1ll << 30)) return false;
    g.MT = (int)mt;
    // efficiency of the tile on ragged extents, and whole rounds of one workgroup per CU
    const double useful = (double)g.Do * g.Ho * g.Wo / ((double)g.tilesD * g.tilesH * g.tilesW * (double)(C::TD * C::TH * C::TW));
    long long nwg = mt * g.nNt;
    const int nC = g.Cin / F9_CH;
    g.ksplit = 1; g.chunksPerSplit = nC; g.slabElems = 0;
    if (nwg < 241 && maySplit && nC >= 2) {
        // too few tiles for one round of 256 workgroups: split K over the 16-channel chunks in equal shares, partial sums to slabs
        int ks = (int)(256 / nwg);
        if (ks > 16) ks = 16;
        while (ks > 1 && nC % ks != 0) --ks;
        if (ks > 1) { g.ksplit = ks; g.chunksPerSplit = nC / ks; g.slabElems = g.yBytes / 4u; nwg *= ks; }
    }
    if (mode != 2 && (useful < 0.9 || nwg < 241 || (double)nwg / (double)((nwg + 255) / 256 * 256) < 0.94)) return false;
    if (g.ksplit > 1 && (unsigned long long)g.yBytes * g.ksplit >= (1ull << 32)) return false;
    // persistent walk: a workgroup keeps its 64-channel block
    grid = (g.ksplit == 1 && nwg > 256 && 256 % g.nNt == 0) ? 256u : (unsigned)(nwg / g.ksplit);
    lds = C::LDS_BYTES;
    return true;
}

bool fwd9_plan(F9Geom& g, size_t& lds, unsigned& grid, int B, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw, int pd,
               int ph, int pw, int epd, int eph, int epw, size_t packedElems, bool maySplit) {
    static const int mode = [] { const char* e = getenv("DIQT_CONV_F9"); return e ? atoi(e) : 1; }();      // 0: never, 2: any tile count
    const bool k333 = kd == 3 && kh == 3 && kw == 3, k133 = kd == 1 && kh == 3 && kw == 3, k311 = kd == 3 && kh == 1 && kw == 1;
    if (!mode || !(k333 || k133 || k311) || Cin % F9_CH != 0 || Cin < F9_CH || Cout < 1) return false;
    if (D > 255 || H > 255 || W > 255) return false;                   // packed 8-bit halo coordinates
    g.B = B; g.D = D; g.H = H; g.W = W; g.Cin = Cin; g.Cout = Cout; g.pd = pd; g.ph = ph; g.pw = pw;
    g.Do = D + 2 * pd + epd - kd + 1; g.Ho = H + 2 * ph + eph - kh + 1; g.Wo = W + 2 * pw + epw - kw + 1;
    if (g.Do <= 0 || g.Ho <= 0 || g.Wo <= 0) return false;
    g.nNt = (Cout + 63) / 64; g.CoutPad = g.nNt * 64;
    const unsigned long long xb = (unsigned long long)B * D * H * W * Cin * 4ull, yb = (unsigned long long)B * g.Do * g.Ho * g.Wo * Cout * 4ull;
    const unsigned long long wb = (unsigned long long)packedElems * 4ull;
    if (xb >= (1ull << 30) || yb >= (1ull << 30) || wb >= (1ull << 30)) return false;
    g.xBytes = (unsigned)xb; g.yBytes = (unsigned)yb; g.wBytes = (unsigned)wb; g.stats = nullptr;
    g.gx = g.gmean = g.grstd = g.ggamma = g.gbeta = g.gscale = g.gshift = nullptr; g.gG = 1; g.gcs = 0; g.gact = 0; g.gnp = nullptr;
    g.gcoef = nullptr; g.gnaAct = 0;
    // un-split launches first (the statistics query of the consumer's GroupNorm plans with maySplit = false and must see the same tiles)
    if (k333) {
        if (f9_try<F9_333_512>(g, lds, grid, mode, false)) { g.variant = 0; return true; }
        if (f9_try<F9_333_256>(g, lds, grid, mode, false)) { g.variant = 1; return true; }
    } else if (k133) {
        if (f9_try<F9_133_A>(g, lds, grid, mode, false)) { g.variant = 2; return true; }
        if (f9_try<F9_133_B>(g, lds, grid, mode, false)) { g.variant = 3; return true; }
        if (f9_try<F9_133_C>(g, lds, grid, mode, false)) { g.variant = 4; return true; }
    } else {
        if (f9_try<F9_311_512>(g, lds, grid, mode, false)) { g.variant = 5; return true; }
        if (f9_try<F9_311_256>(g, lds, grid, mode, false)) { g.variant = 6; return true; }
    }
    if (!maySplit) return false;
    // split-K: the small tiles first (fewer slabs for the same number of workgroups)
    if (k333) {
        if (f9_try<F9_333_256>(g, lds, grid, mode, true)) { g.variant = 1; return true; }
        if (f9_try<F9_333_512>(g, lds, grid, mode, true)) { g.variant = 0; return true; }
    } else if (k133) {
        if (f9_try<F9_133_C>(g, lds, grid, mode, true)) { g.variant = 4; return true; }
        if (f9_try<F9_133_B>(g, lds, grid, mode, true)) { g.variant = 3; return true; }
        if (f9_try<F9_133_A>(g, lds, grid, mode, true)) { g.variant = 2; return true; }
    } else {
        if (f9_try<F9_311_256>(g, lds, grid, mode, true)) { g.variant = 6; return true; }
        if (f9_try<F9_311_512>(g, lds, grid, mode, true)) { g.variant = 5; return true; }
    }
    return false;
}

int fwd9_launch(const float* x, const float* packed, const float* bias, const float* residual, float* y, const F9Geom& g, size_t lds,
                unsigned grid, void* stream) {
    if (g.gx) return g.variant == 0 ? fwd9_launch_d(x, packed, bias, residual, y, g, lds, grid, stream)
                                    : fwd9_launch_e(x, packed, bias, residual, y, g, lds, grid, stream);
    if (g.gcoef) {
        switch (g.variant) {
            case 0: return fwd9_launch_f(x, packed, bias, residual, y, g, lds, grid, stream);
            case 1: case 4: return fwd9_launch_g(x, packed, bias, residual, y, g, lds, grid, stream);
            case 2: return fwd9_launch_h(x, packed, bias, residual, y, g, lds, grid, stream);
            case 3: return fwd9_launch_i(x, packed, bias, residual, y, g, lds, grid, stream);
        }
        set_error("conv3d_fwd(v9, GroupNorm-apply prologue): no variant %d", g.variant);
        return DIQT_E_UNSUPPORTED;
    }
    switch (g.variant) {
        case 0: return f9_launch<F9_333_512>(x, packed, bias, residual, y, g, lds, grid, stream);
        case 2: case 3: case 4: return fwd9_launch_b(x, packed, bias, residual, y, g, lds, grid, stream);
        case 1: case 5: case 6: return fwd9_launch_c(x, packed, bias, residual, y, g, lds, grid, stream);
    }
    set_error("conv3d_fwd(v9): no variant %d", g.variant);
    return DIQT_E_UNSUPPORTED;
}

}  // namespace diqt
