// Internal interface of the version-9 forward conv (conv_fwd9.hip), called from conv3d_fwd_one (conv_mfma.hip).
#pragma once
#include <stddef.h>

namespace diqt {

struct F9GnParams {          // device-resident parameters of the GroupNorm-backward epilogue
    const float *mean, *rstd, *gamma, *beta, *scale, *shift;
    int G, cs, act, pad;
};

struct F9Geom {
    int B, D, H, W, Cin, Cout, Do, Ho, Wo, pd, ph, pw;
    int tilesD, tilesH, tilesW, MT, nNt, CoutPad, variant;
    int ksplit, chunksPerSplit;  // split-K launches: gridDim.y slabs of slabElems floats, chunksPerSplit 16-channel chunks each
    unsigned slabElems;
    unsigned xBytes, yBytes, wBytes;
    float* stats;                // optional per-tile column sums of the output: [B][tiles per batch][2][Cout]
    // GroupNorm-backward epilogue (this launch is the backward-data pass of the conv behind a GroupNorm + activation): with gx set,
    // the statistics rows hold sum(dz), sum(dz * xhat) of dz = output * act'(A gx + B), xhat = (gx - mean) rstd instead of the
    // output's sums -- the reduction pass of the GroupNorm backward, without its read of gx and of this output
    // (the kernel gets gx alone -- the flag -- and a device copy of the other fields through gnp, so that they occupy no scalar registers
    // in the main loop)
    const float *gx, *gmean, *grstd, *ggamma, *gbeta, *gscale, *gshift;
    int gG, gcs, gact;
    const struct F9GnParams* gnp;
    // GroupNorm-apply prologue (sampling path; the GNA instantiations): the input x is the RAW GroupNorm input and every halo piece is
    // replaced in the LDS, right after it landed, by act(A[b][c] x + Bc[b][c]) -- the pass of gn_act_fwd_kernel over the tensor (a read
    // and a write of the whole activation per conv) disappears.  gcoef = [2][B][Cin] floats: A, then Bc (diqt_gn_coef*).
    const float* gcoef;
    int gnaAct;                  // DIQT_ACT_MISH / DIQT_ACT_SILU, 0: none
};

// which (F9Geom::variant, activation) pairs have a GroupNorm-apply instantiation: Mish on the 3x3x3 tiles, SiLU on the (1,3,3) ones
inline bool fwd9_gna_available(int variant, int act) {
    return (act == 1 /* DIQT_ACT_MISH */ && (variant == 0 || variant == 1)) || (act == 2 /* DIQT_ACT_SILU */ && variant >= 2 && variant <= 4);
}

// maySplit: the caller has a workspace for split-K slabs (g.ksplit * output elements floats when g.ksplit > 1; the kernel then gets
// the slab base as y and no bias / residual / statistics).
// Does conv_fwd9_kernel take this launch (3x3x3 or (1,3,3), Cin % 16 == 0, whole rounds of 512- / 256-voxel tiles, tensors < 1 GiB)?
bool fwd9_plan(F9Geom& g, size_t& lds, unsigned& grid, int B, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw, int pd,
               int ph, int pw, int epd, int eph, int epw, size_t packedElems, bool maySplit);
int fwd9_launch(const float* x, const float* packed, const float* bias, const float* residual, float* y, const F9Geom& g, size_t lds,
                unsigned grid, void* stream);

}  // namespace diqt
