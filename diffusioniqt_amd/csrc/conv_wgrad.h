// Internal interface of the version-3 weight-gradient kernel (conv_wgrad.hip), called from diqt_conv3d_bwd_weight (conv_mfma.hip).
#pragma once
#include <stddef.h>

namespace diqt {

struct W3Geom {
    int B, D, H, W, Cin, Cout, Do, Ho, Wo, pd, ph, pw;
    int tilesD, tilesH, tilesW, MT, tilesPerSplit;
    int nCoB, nCiB, CoutPad, nChunks32;
    unsigned xBytes, yBytes;
    unsigned long long* dbg;     // diagnostic cycle stamps (NULL in production): 8 words per wave
};

// Does conv_wgrad3_kernel take this shape?  Fills the geometry, the template variant, the split-K count (= slabs written, in the
// final [slice][Cout][Cin][taps] layout conv_reduce_dw3_kernel sums) and the dynamic LDS size.
bool wgrad3_plan(W3Geom& g, int& variant, int& ksplit, size_t& lds, int B, int D, int H, int W, int Cin, int Cout, int kd, int kh,
                 int kw, int pd, int ph, int pw, int epd, int eph, int epw);
int wgrad3_launch(const float* x, const float* dy, float* slabs, float* bias_part, const W3Geom& g, int variant, int ksplit, size_t lds,
                  void* stream);

extern unsigned long long* wgrad3_dbg_ptr;   // stamps of the last DIQT_CONV_DBG=1 launch (8 words per wave)
extern unsigned wgrad3_dbg_n;

}  // namespace diqt
