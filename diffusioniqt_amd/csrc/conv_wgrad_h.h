// Internal interface of the 16-bit-operand weight-gradient kernel (conv_wgrad_h.hip), called from diqt_conv3d_bwd_weight_h (conv_mfma.hip).
#pragma once
#include <stddef.h>

namespace diqt {

struct WHGeom {
    int B, D, H, W, Cin, Cout, Do, Ho, Wo, pd, ph, pw, kd, kh, kw;
    int tilesD, tilesH, tilesW, MT, tilesPerSplit;
    int nCoB, nCiB, CoutPad;
    unsigned xBytes, yBytes;
    int xHalf, dyHalf;           // x / dY hold 16-bit values of the operand type
};

// Does conv_wgrad_h_kernel take this shape (3x3x3, 1x3x3 or 3x1x1; Cin % 32 == 0, tensors < 1 GiB)?  Fills the geometry and the split-K count (= slabs
// written in the [slice][Cout][Cin][taps] layout that conv_reduce_dw3_kernel sums, + bias partials [slice][CoutPad]).
bool wgradh_plan(WHGeom& g, int& ksplit, int B, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw, int pd, int ph, int pw,
                 int epd, int eph, int epw, bool xHalf = false, bool dyHalf = false);
int wgradh_launch(const float* x, const float* dy, float* slabs, float* bias_part, const WHGeom& g, int ksplit, int bf16, void* stream);

}  // namespace diqt
