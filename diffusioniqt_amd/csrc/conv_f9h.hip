// Planner and launcher of conv_f9h_kernel (conv_f9h_kernel.h has the description) + its 3x3x3 instantiations.
#include "conv_f9h_kernel.h"
#include <stdlib.h>
#include <atomic>

// 0: conv_f9h_kernel never takes a launch, 1 (default; env DIQT_CONV_F9H): launches that fill the chip, 2: any tile count (tests run the
// kernel on small shapes this way).  mode >= 0 sets it, mode < 0 only queries; returns the previous value.
static std::atomic<int> g_f9h_mode{-1};
extern "C" int diqt_set_conv_f9h_mode(int mode) {
    int cur = g_f9h_mode.load();
    if (cur < 0) {
        const char* e = getenv("DIQT_CONV_F9H");
        const int v = e ? atoi(e) : 1;
        g_f9h_mode.compare_exchange_strong(cur, v < 0 ? 1 : v);
        cur = g_f9h_mode.load();
    }
    if (mode >= 0) g_f9h_mode.store(mode);
    return cur;
}

static unsigned long long* g_f9dbg = nullptr;   // diagnostic only (DIQT_F9H_DBG=1)
static unsigned g_f9dbg_n = 0;
// diagnostic only (not part of include/diqt.h): the cycle stamps of the last DIQT_F9H_DBG=1 launch, 32 per wave, 4 waves per workgroup
extern "C" int diqt_debug_f9h_stamps(unsigned long long* host_out, unsigned max_waves) {
    if (!g_f9dbg || !g_f9dbg_n) return 0;
    const unsigned n = g_f9dbg_n < max_waves ? g_f9dbg_n : max_waves;
    if (hipDeviceSynchronize() != hipSuccess) return 0;
    if (hipMemcpy(host_out, g_f9dbg, (size_t)n * 32 * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return 0;
    return (int)n;
}

namespace diqt {

// One candidate tiling: tile counts, grid, and an estimate of the launch's duration in (voxel x workgroup-round) units -- rounds of
// 256 OCC workgroup slots, each round as long as a tile (two workgroups sharing a CU run at half speed each).
template <class C> static bool f9h_try(H9Geom& g, size_t& lds, unsigned& grid, double& est) {
    g.tilesD = (g.Do + C::TD - 1) / C::TD; g.tilesH = (g.Ho + C::TH - 1) / C::TH; g.tilesW = (g.Wo + C::TW - 1) / C::TW;
    const long long mt = (long long)g.B * g.tilesD * g.tilesH * g.tilesW;
    if (mt >= (1ll << 30)) return false;
    g.MT = (int)mt;
    const long long nwg = mt * g.nNt;
    const long long slots = 256 * C::OCC;
    // persistent walk: a workgroup keeps its 64-channel block (its weight stream)
    unsigned gr = nwg > slots ? (unsigned)slots : (unsigned)nwg;
    if (nwg > slots) gr -= gr % (unsigned)g.nNt;
    if (gr == 0) return false;
    const long long rounds = (nwg + gr - 1) / gr;
    est = (double)rounds * (double)(C::TD * C::TH * C::TW) * (double)C::OCC * (nwg <= 256 ? 1.0 / C::OCC : 1.0);   // one workgroup per CU: full speed
    grid = gr;
    lds = C::LDS_BYTES;
    return true;
}

bool f9h_plan(H9Geom& g, size_t& lds, unsigned& grid, int B, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw, int pd, int ph,
              int pw, int epd, int eph, int epw, bool yHalf) {
    const int mode = diqt_set_conv_f9h_mode(-1);
    const bool k333 = kd == 3 && kh == 3 && kw == 3, k133 = kd == 1 && kh == 3 && kw == 3;
    if (!mode || !(k333 || k133) || Cin % h9::CK != 0 || Cout % 8 != 0 || Cout < 8) return false;
    if (B <= 0 || D <= 0 || H <= 0 || W <= 0 || D > 1000 || H > 1000 || W > 1000) return false;      // packed 10-bit halo coordinates
    if (pd < 0 || ph < 0 || pw < 0 || pd > 16 || ph > 16 || pw > 16) return false;
    g.B = B; g.D = D; g.H = H; g.W = W; g.Cin = Cin; g.Cout = Cout; g.pd = pd; g.ph = ph; g.pw = pw;
    g.Do = D + 2 * pd + epd - kd + 1; g.Ho = H + 2 * ph + eph - kh + 1; g.Wo = W + 2 * pw + epw - kw + 1;
    if (g.Do <= 0 || g.Ho <= 0 || g.Wo <= 0) return false;
    g.nNt = (Cout + 63) / 64; g.CoutPad = g.nNt * 64; g.nChunks = Cin / h9::CK;
    const unsigned long long xb = (unsigned long long)B * D * H * W * Cin * 2ull;
    const unsigned long long vox = (unsigned long long)B * g.Do * g.Ho * g.Wo;
    const unsigned long long yb = vox * Cout * (yHalf ? 2ull : 4ull), rb = vox * Cout * 4ull;
    const unsigned long long wb = (unsigned long long)g.nChunks * kd * kh * kw * g.CoutPad * h9::CK * 2ull;
    if (xb >= (1ull << 30) || rb >= (1ull << 31) || wb >= (1ull << 30)) return false;
    g.xBytes = (unsigned)xb; g.yBytes = (unsigned)yb; g.rBytes = (unsigned)rb; g.wBytes = (unsigned)wb; g.stats = nullptr; g.dbg = nullptr; g.dbgSkip = 0;
    // the candidate with the shortest estimate wins; ties go to the earlier one = the measured preference (MI355X, round 4): the 256-voxel
    // tiles at two workgroups per CU are 0-8 % faster than the 512-voxel ones on the 3x3x3 shapes of C2 (58.7 vs 63.4 us on 64 -> 64 @
    // 8 x 32^3 with an fp32 y) and on the 64-channel per-frame convs (208 vs 219 us @ 8 x 64^3), 4 % slower at 128 channels.  Small
    // grids are taken too: 128 -> 128 @ 8 x 8^3 (32 workgroups) 21.7 us against 47.6 us on conv_fwd_h_kernel.
    static const int force = [] { const char* e = getenv("DIQT_F9H_VARIANT"); return e ? atoi(e) : -1; }();      // experiments: this variant only
    H9Geom best = g; size_t bl = 0; unsigned bg = 0; double be = 1e300; int bv = -1;
#define F9H_TRY(CFG, V) if (force < 0 || force == V) { H9Geom t = g; size_t l_; unsigned g_; double e_; \
        if (f9h_try<h9::CFG>(t, l_, g_, e_) && e_ < be) { best = t; bl = l_; bg = g_; be = e_; bv = V; } }
    if (k333) {
        F9H_TRY(H9_333_256, 1)
        F9H_TRY(H9_333_512, 0)
    } else {
        if (Cin <= 64) { F9H_TRY(H9_133_D, 5) }
        F9H_TRY(H9_133_A, 2)
        F9H_TRY(H9_133_B, 3)
        F9H_TRY(H9_133_D, 5)
        F9H_TRY(H9_133_C, 4)
    }
#undef F9H_TRY
    if (bv < 0) return false;
    // tiles that are mostly padding (a volume much smaller than any tile): leave the launch to the other kernels
    const double usefulVox = (double)g.B * g.Do * g.Ho * g.Wo * g.nNt;
    if (mode != 2 && be * 256.0 > 6.0 * (usefulVox < 256.0 * 256.0 ? 256.0 * 256.0 : usefulVox)) return false;
    g = best; g.variant = bv; lds = bl; grid = bg;
    return true;
}

int f9h_stats_blocks(const H9Geom& g) { return g.tilesD * g.tilesH * g.tilesW * 2; }

int f9h_launch(const void* x, const unsigned short* packed_h, const float* bias, const float* residual, void* y, const H9Geom& g0, size_t lds,
               unsigned grid, int bf16, bool yHalf, void* stream) {
    H9Geom g = g0;
    static const int dbg_mode = [] { const char* e = getenv("DIQT_F9H_DBG"); return e ? atoi(e) : 0; }();
    const bool dbg_on = dbg_mode > 0;
    g.dbgSkip = dbg_mode == 2 ? 80 : 0;
    if (dbg_on) {
        if (!g_f9dbg) DIQT_REQUIRE(hipMalloc(&g_f9dbg, (size_t)1024 * 32 * sizeof(unsigned long long)) == hipSuccess, DIQT_E_LAUNCH, "conv3d_fwd_h(v9h): debug buffer");
        DIQT_REQUIRE(hipMemsetAsync(g_f9dbg, 0, (size_t)1024 * 32 * sizeof(unsigned long long), (hipStream_t)stream) == hipSuccess, DIQT_E_LAUNCH, "conv3d_fwd_h(v9h): debug buffer");
        g.dbg = g_f9dbg; g_f9dbg_n = grid * 4;
    }
    switch (g.variant) {
        case 0: return h9::launch_cfg<h9::H9_333_512>(x, packed_h, bias, residual, y, g, lds, grid, bf16, yHalf, stream);
        case 1: return h9::launch_b(x, packed_h, bias, residual, y, g, lds, grid, bf16, yHalf, stream);
        case 2: case 3: case 4: case 5: return h9::launch_c(x, packed_h, bias, residual, y, g, lds, grid, bf16, yHalf, stream);
    }
    set_error("conv3d_fwd_h(v9h): no variant %d", g.variant);
    return DIQT_E_UNSUPPORTED;
}

}  // namespace diqt
