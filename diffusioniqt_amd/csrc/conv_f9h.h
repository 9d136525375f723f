// Internal interface of the version-9h 16-bit forward conv (conv_f9h.hip), called from convh_launch (conv_half.hip).
#pragma once
#include <stddef.h>

namespace diqt {

struct H9Geom {
    int B, D, H, W, Cin, Cout, Do, Ho, Wo, pd, ph, pw;
    int tilesD, tilesH, tilesW, MT, nNt, CoutPad, nChunks, variant;
    unsigned xBytes, yBytes, wBytes, rBytes;
    int dbgSkip;                 // diagnostic: first stamp slot recorded
    unsigned long long* dbg;     // diagnostic cycle stamps per wave (DIQT_F9H_DBG=1), NULL in production
    float* stats;                // optional column sums (sum, sum of squares) of the stored values: [B][tiles per batch * 2][2][Cout]
};

// Does conv_f9h_kernel take this launch?  16-bit x, Cin % 32 == 0, Cout % 8 == 0, a 3x3x3 or (1,3,3) filter, enough 512- / 256-voxel
// tiles to fill the chip, tensors < 1 GiB.  yHalf: y is stored in the operand type.
bool f9h_plan(H9Geom& g, size_t& lds, unsigned& grid, int B, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw, int pd, int ph,
              int pw, int epd, int eph, int epw, bool yHalf);
// rows of column sums per batch entry the kernel writes for this plan
int f9h_stats_blocks(const H9Geom& g);
int f9h_launch(const void* x, const unsigned short* packed_h, const float* bias, const float* residual, void* y, const H9Geom& g, size_t lds,
               unsigned grid, int bf16, bool yHalf, void* stream);

}  // namespace diqt
