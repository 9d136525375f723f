// conv_fwd9_kernel with the GroupNorm-backward epilogue: 256-voxel 3x3x3 and the (1,3,3) variants (see conv_fwd9.hip).  The temporal
// (3,1,1) convs never sit directly behind a GroupNorm.
#include "conv_fwd9_kernel.h"

namespace diqt {

int fwd9_launch_e(const float* x, const float* packed, const float* bias, const float* residual, float* y, const F9Geom& g, size_t lds,
                  unsigned grid, void* stream) {
    switch (g.variant) {
        case 1: return f9_launch<F9_333_256, true>(x, packed, bias, residual, y, g, lds, grid, stream);
        case 2: return f9_launch<F9_133_A, true>(x, packed, bias, residual, y, g, lds, grid, stream);
        case 3: return f9_launch<F9_133_B, true>(x, packed, bias, residual, y, g, lds, grid, stream);
        case 4: return f9_launch<F9_133_C, true>(x, packed, bias, residual, y, g, lds, grid, stream);
    }
    set_error("conv3d_fwd(v9, GroupNorm-backward epilogue): no variant %d in this unit", g.variant);
    return DIQT_E_UNSUPPORTED;
}

}  // namespace diqt
