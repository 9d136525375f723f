// conv_fwd9_kernel with the GroupNorm-apply prologue (see conv_fwd9_kernel.h): the 2x16x16 (1,3,3) variant with SiLU (Family B, 16 x 16 frames)
#include "conv_fwd9_kernel.h"

namespace diqt {

int fwd9_launch_i(const float* x, const float* packed, const float* bias, const float* residual, float* y, const F9Geom& g, size_t lds,
                  unsigned grid, void* stream) {
    if (g.variant == 3 && g.gnaAct == DIQT_ACT_SILU) return f9_launch<F9_133_B, false, DIQT_ACT_SILU>(x, packed, bias, residual, y, g, lds, grid, stream);
    set_error("conv3d_fwd(v9, GroupNorm-apply prologue): no variant %d / activation %d in this unit", g.variant, g.gnaAct);
    return DIQT_E_UNSUPPORTED;
}

}  // namespace diqt
